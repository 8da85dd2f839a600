"""unlearn_saliency_amd.SD — part of the MI355X-native SalUn hot path."""
