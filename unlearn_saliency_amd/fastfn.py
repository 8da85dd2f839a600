"""`torch.autograd.Function` with the per-call Python overhead that matters at ~10 k launches per step taken out.

`Function.apply` unwraps every argument for functorch (`_functorch.utils.unwrap_dead_wrappers`: a Python generator over
the arguments) and probes `setup_context` before it reaches the C++ `apply`; under the host profile of an SD step
(tools/hostprof_diffusion.py) that was 3,960 calls x 24 us.  No functorch transform is ever active on this path, so
`FastFunction.apply` goes straight to the C++ entry and falls back to the stock route only if a transform is active."""
import torch


_transforms_active = getattr(torch._C, "_are_functorch_transforms_active", None)


class FastFunction(torch.autograd.Function):
    if _transforms_active is not None:  # (a torch without the probe keeps the stock `apply`)
        @classmethod
        def apply(cls, *args):
            if _transforms_active():
                return super().apply(*args)
            return super(torch.autograd.Function, cls).apply(*args)
