"""unlearn_saliency_amd.DDPM.runners — part of the MI355X-native SalUn hot path."""
