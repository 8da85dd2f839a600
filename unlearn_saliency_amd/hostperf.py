"""Host-side hygiene for loops that issue ~10 k kernel launches per step.

CPython's cyclic collector runs a full (generation-2) pass every so many container allocations and walks EVERY tracked
object each time; with a U-Net's modules, parameters, hooks and caches alive that is tens of milliseconds, paid a few
times per step at whatever line happened to allocate (tools/hostprof_diffusion.py showed 30–45 ms per SD step charged to
a two-line tuple constructor).  `freeze_gc()` — called by the training loops once the model, optimizer and loaders exist
— collects once and moves everything alive into the permanent generation (`gc.freeze`), so later passes only look at the
step's own short-lived objects.  Idempotent and cheap to call again (e.g. after a model was rebuilt)."""
import gc


def freeze_gc() -> None:
    # unfreeze first: what an earlier call froze and has since become garbage (a previous run's model in the same process)
    # goes back to the collector instead of staying immortal
    gc.unfreeze()
    gc.collect()
    gc.freeze()
