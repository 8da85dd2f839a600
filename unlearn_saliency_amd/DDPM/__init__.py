"""unlearn_saliency_amd.DDPM — part of the MI355X-native SalUn hot path."""
