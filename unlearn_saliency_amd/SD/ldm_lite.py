"""`LatentDiffusionLite`: the slice of the reference's `LatentDiffusion`
(SD/ldm/models/diffusion/ddpm.py:424-430 q_sample, :913-973 get_input, :1093-1109 shared_step,
:1121 apply_model, :1286-1319 p_losses) that the SalUn scripts call — schedule buffers, `q_sample`,
`apply_model`, `shared_step`, `get_input` — around the U-Net of unet.py.

The frozen first stage (AutoencoderKL) and text encoder (CLIP) are outside the hot-path scope (SURVEY.md §2 S4;
their weights are not available offline), so batches carry *latents and context embeddings* directly:
``{"z": (B,4,64,64) fp32, "c": (B,77,768) fp32}``.  A caller that has the encoders can pass
`first_stage` / `cond_stage` callables and feed the reference's ``{"jpg": ..., "txt": ...}`` batches.
Parameter names keep the reference's prefix (`model.diffusion_model.*`), which is what the mask-key
rewrite in the scripts (`n.split("model.diffusion_model.")[-1]`, random_label.py:135) relies on.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import draws, ops
from .unet import UNetModel, V1_UNET_CONFIG


# AutoencoderKL (ddconfig of configs/stable-diffusion/v1-inference.yaml) + CLIP ViT-L/14 text model, counted from the
# reference's modules
SD_V1_FROZEN_PARAMS = 83_653_863 + 123_060_480


class _Wrapper(nn.Module):
    def __init__(self, unet):
        super().__init__()
        self.diffusion_model = unet


class LatentDiffusionLite(nn.Module):
    def __init__(self, unet_config=None, timesteps=1000, linear_start=0.00085, linear_end=0.0120,
                 first_stage=None, cond_stage=None, scale_factor=0.18215, bf16=False):
        super().__init__()
        self.model = _Wrapper(UNetModel(**(unet_config or V1_UNET_CONFIG)))
        self.num_timesteps = int(timesteps)
        self.first_stage_key, self.cond_stage_key = "jpg", "txt"
        self.first_stage, self.cond_stage, self.scale_factor = first_stage, cond_stage, scale_factor
        self.bf16 = bf16
        # Parameters of the frozen stages a full LatentDiffusion carries next to the U-Net: the reference's proximal step
        # ranks |theta - theta_0| over ALL of them (proximal_gradient.py:66-72,141-167), so their zeros take the lowest
        # ranks.  Attached stages that are nn.Modules are counted; with none attached a v1-shaped U-Net gets the SD-v1
        # constant (AutoencoderKL 83,653,863 + CLIP ViT-L/14 text model 123,060,480); anything else has no frozen
        # stages (0).  Only the proximal step's global ranking depends on it; `_unlearn` logs the value it used.
        cfg = unet_config or V1_UNET_CONFIG
        counted = sum(sum(p.numel() for p in st.parameters()) for st in (first_stage, cond_stage)
                      if isinstance(st, nn.Module))
        v1 = all(cfg.get(k) == V1_UNET_CONFIG.get(k) for k in ("model_channels", "channel_mult", "context_dim"))
        self.frozen_param_count = counted if counted else (SD_V1_FROZEN_PARAMS if v1 else 0)
        # "linear" schedule of the LDM code base: linspace(sqrt(start), sqrt(end))**2 in float64 (util.py:24-30)
        betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
        ac = np.cumprod(1.0 - betas, axis=0)
        self.register_buffer("sqrt_alphas_cumprod", torch.tensor(np.sqrt(ac), dtype=torch.float32))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32))

    @property
    def device(self):
        return self.sqrt_alphas_cumprod.device

    def use_mfma_convs(self) -> int:
        """Put the U-Net's convolutions on this package's matrix-core kernels: bf16 NHWC (conv_bf16.py, K11) in the
        bf16 configuration, fp32 NCHW (conv.py, K8) otherwise.  Returns the number of modules switched."""
        if self.bf16:
            from ..conv_bf16 import use_salun_convs_bf16, use_salun_linears_bf16
            # The transformer blocks' Linear layers run on K16 (csrc/salun_gemm.hip: direct-to-LDS bf16 GEMM, forward and
            # input gradient, bias / residual in the epilogue) and K11 (weight gradient, added in fp32 into the flat
            # gradient): cached bf16 weight images instead of a cast per forward / recompute / backward, no
            # AccumulateGrad launches.  Round 3 had them on the K11 1x1 kernels alone and lost 7 % of the step to the
            # library GEMM; K16 is at the library's speed on these shapes (profiles/r04_gemmbench_bf16.txt) and the step
            # gains 3 % (194 vs 200 ms).  `mfma_linears = False` keeps them on the library (A/B).
            if getattr(self, "mfma_linears", True):
                self.n_mfma_linears = use_salun_linears_bf16(self.model.diffusion_model)
            return use_salun_convs_bf16(self.model.diffusion_model)
        from ..conv import use_salun_convs
        return use_salun_convs(self.model.diffusion_model)

    def fill_zero_initialised(self, std: float = 0.02, seed: int = 11) -> int:
        """Benchmarks / full-size tests only (no checkpoint offline): the reference's `zero_module` layers (each
        ResBlock's last convolution, every transformer's `proj_out`, the output convolution) and all biases start at
        exactly zero, so a freshly constructed U-Net back-propagates zero gradients to everything in front of its last
        layer and multiplies by zeros throughout — not what a trained checkpoint does, and zero operands also let the
        chip clock higher (MI355X_MICROARCH.md, DVFS).  Every parameter element that is exactly 0 gets N(0, std)."""
        from ..flat import FlatArena
        arena = getattr(self, "_salun_unet_arena", None)
        if arena is None:
            arena = FlatArena(self.model.diffusion_model.named_parameters())
            object.__setattr__(self, "_salun_unet_arena", arena)
        z = arena.params == 0
        n = int(z.sum())
        if n:
            arena.params[z] = ops.fill_normal(arena.n, seed, 0.0, std)[z]
        return n

    # ---- the calls the scripts make
    def get_input(self, batch, k=None):
        """-> (z, c).  Accepts {"z","c"} (latents + context) or, with encoders attached, {"jpg","txt"}."""
        if "z" in batch:
            return batch["z"].to(self.device), batch["c"].to(self.device)
        if self.first_stage is None or self.cond_stage is None:
            raise RuntimeError("image/text batches need the frozen VAE and CLIP encoders (out of the hot-path scope); "
                               "feed {'z': latents, 'c': context} batches or attach first_stage/cond_stage callables")
        x = batch["jpg"].permute(0, 3, 1, 2).to(self.device).float()
        with torch.no_grad():
            return self.first_stage(x) * self.scale_factor, self.cond_stage(batch["txt"])

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return ops.qsample(x_start.contiguous().float(), noise.contiguous().float(), self.sqrt_alphas_cumprod,
                           self.sqrt_one_minus_alphas_cumprod, t.to(torch.int64).contiguous())

    def apply_model(self, x_noisy, t, cond):
        if self.bf16:
            # the 32 to_k / to_v projections each round the text context to bf16 before they read it: do it once (same values)
            if torch.is_tensor(cond) and cond.dtype == torch.float32 and not cond.requires_grad:
                cond = cond.to(torch.bfloat16)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return self.model.diffusion_model(x_noisy, t, context=cond).float()
        return self.model.diffusion_model(x_noisy, t, context=cond)

    def shared_step(self, batch):
        """(loss, {}) with loss = mean MSE(noise, eps(x_t, t, c)) — the reference's p_losses with
        logvar = 0, l_simple_weight = 1, original_elbo_weight = 0 (ddpm.py:82-90,1286-1319)."""
        x, c = self.get_input(batch, self.first_stage_key)
        # under data parallel with a sharded global batch (draws.scope) t / noise are drawn for the global batch and sliced
        t = draws.randint(self.num_timesteps, x.shape[0], self.device).long()
        noise = draws.randn_like(x)
        out = self.apply_model(self.q_sample(x, t, noise), t, c)
        return ops.mse_loss(noise, out), {}
