"""Host-side mirror of the reference's Classification/ entry points for the SalUn hot path."""
