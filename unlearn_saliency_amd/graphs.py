"""Whole-step HIP graphs: the host off the critical path of the diffusion unlearning steps.

An SD `nsfw_removal` step issues ~15,000 kernel launches from Python (profiles/r03_sd_bf16_kernel_stats.csv: the host
needs 193 of the step's 208 ms to enqueue them), a DDPM step ~2,500.  `StepGraph` captures ONE step — forward passes,
backward (side streams included: they fork from and join the capture stream through events), the fused optimizer —
into a HIP graph (`torch.cuda.CUDAGraph` = hipGraph on ROCm) and replays it with one host call per step.

What makes a step capturable here:
  * nothing in it synchronises the host (the clip coefficient, the top-k status, the losses all stay on the device);
  * every kernel of libsalun.so takes the stream from `torch.cuda.current_stream()`, i.e. the capture stream;
  * per-step scalars that change between steps live in DEVICE memory: Adam's step count
    (`FusedMaskedAdam.use_device_step`, salun_adam_coefficients), the dropout key (`salun_dropout`'s seed word, advanced
    by `salun_u64_add`), torch's own Philox offsets (graph-safe generators);
  * inputs are copied into static buffers before each replay, outputs are read from static tensors after it;
  * host-side caches whose state a replay cannot see are made consistent: the bf16 weight images are re-packed by the
    captured step itself (the capture starts right after an optimizer step, when every image is stale).

Not captured (falls back to eager, loudly): data-parallel runs (collectives_on) — RCCL capture is left for a later
round — and anything that raises during capture.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import dist as sdist


class StepGraph:
    def __init__(self, step_fn: Callable[..., Sequence[torch.Tensor]], example_inputs: Sequence[torch.Tensor],
                 warmup: int = 2, on_replay: Optional[Callable[[], None]] = None):
        """`step_fn(*inputs)` runs one whole step and returns a tuple of device tensors (e.g. the loss).  It is run
        `warmup` times eagerly on a side stream (kernel selection, shape probes, workspaces, weight-image caches), then
        captured once.  `on_replay`: host bookkeeping per replay (e.g. the optimizer's host step counter)."""
        if sdist.collectives_on():
            raise RuntimeError("StepGraph: data-parallel steps are not captured (RCCL collectives stay eager)")
        self.step_fn = step_fn
        self.on_replay = on_replay
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                step_fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = step_fn(*self.static_in)
        self.static_out = tuple(out) if isinstance(out, (tuple, list)) else (out,)
        self.replays = 0

    def __call__(self, *inputs: torch.Tensor):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        if self.on_replay is not None:
            self.on_replay()
        return self.static_out
