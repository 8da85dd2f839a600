from .diffusion import Conditional_Model
