"""unlearn_saliency_amd — MI355X-native SalUn hot path (gfx950).

Scope: saliency-mask generation and the masked random-label unlearning
step of OPTML-Group/Unlearn-Saliency, as hand-written HIP kernels behind a C-ABI
(include/salun.h -> libsalun.so), with a Python host side that mirrors the reference's
plugin surface (``Classification/unlearn``, ``DDPM/runners``) so it is a drop-in for
that path and nothing else.

    ops        tensor-level wrappers over the C-ABI (no CPU fallback)
    flat       flat, 16-byte-aligned parameter/gradient arena the kernels stream over
    dist       one-process-per-GPU helpers (RCCL all-reduce of the flat vectors)
    Classification / DDPM   host-side mirrors of the reference entry points
"""
__version__ = "0.1.0"

import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the
# other.  This package overlaps kernels on side streams (streams.py probes for a queue of their own); more queues make
# collisions with the communicator's streams rarer.  Read by the HIP runtime when it initialises, i.e. only effective
# when this import happens before the process first touches the device; never overrides the user's setting.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
