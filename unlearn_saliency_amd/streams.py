"""Side streams that really run beside the main stream.

HIP maps every stream onto one of a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default); two streams on the same
queue execute strictly one after the other.  Single-process runs got lucky — the current stream and the first
`torch.cuda.Stream()` landed on different queues — but with a process group the RCCL communicator's own streams shift
the assignment, and the backward-weight side stream of resblock.py ended up on the MAIN stream's queue: the
data-parallel ResNet-18 step ran every kernel back to back, 10.03 ms instead of 8.53 (round 6,
`bench.py --force_collectives`; with GPU_MAX_HW_QUEUES=8 the same run takes 8.66 ms).

`concurrent_stream(device)` therefore PROBES: it keeps the first of a few candidate streams on which a small kernel
finishes while a few milliseconds of work are still queued on the current stream — and, when a process group over
RCCL exists, beside which a small all-reduce completes while the candidate is busy.  The collectives of that probe must
pair up across ranks, so they are issued only from `prepare()`, which `dist.init_from_env` calls on every rank right after
it created the process group (every rank probes the SAME number of candidates whatever each finds) and which stocks a
small pool; `concurrent_stream()` hands out the pool first and otherwise probes WITHOUT collectives — a rank that skips a
backward pass (empty shard) must not leave the others waiting in one.  ~10 - 40 ms per stream, once.
"""
from __future__ import annotations

import os

import torch

_PROBE = os.environ.get("SALUN_STREAM_PROBE", "1") != "0"
_keep: list = []   # rejected candidates stay alive: torch hands streams out of a pool, a freed one would come back
STATS = {"probes": 0, "rejected": 0}
_DP_CANDIDATES = int(os.environ.get("SALUN_STREAM_CANDIDATES", "4"))  # candidates every rank probes under a process group
_accepted: dict = {}  # device -> side streams handed out so far: a new one must run beside each of them too (the
                      # diffusion steps keep two: backward-weight and the no-grad target pass; sharing one queue cost the
                      # data-parallel DDPM step 19 % — 126 vs 106 ms — although the two are busy in different phases)


def _busy(stream, buf, n=24):
    with torch.cuda.stream(stream):
        for _ in range(n):          # ~3 ms of streaming passes over 128 MB
            buf.mul_(1.0)


def _beside_main(main, cand, buf, flag) -> bool:
    main.synchronize()
    cand.synchronize()
    e_c = torch.cuda.Event(enable_timing=True)
    e_m = torch.cuda.Event(enable_timing=True)
    _busy(main, buf)
    e_m.record(main)
    with torch.cuda.stream(cand):
        flag.add_(1.0)
        e_c.record(cand)
    main.synchronize()
    cand.synchronize()
    # same hardware queue: the candidate's kernel ran after everything queued on main -> e_c is not before e_m
    return e_c.elapsed_time(e_m) > 0.5


def _beside_collectives(main, cand, buf, probe) -> bool:
    """Data parallel: the communicator runs its collectives on a stream of its own.  If THAT stream shares the
    candidate's queue, every backward-weight kernel queued after a gradient slice's all-reduce waits for it — and the
    all-reduce waits for the main stream to reach the point it was issued at (seen once in round 6: 13.8 ms per step
    instead of 8.7).  So: with the candidate busy, a small all-reduce issued and awaited from the idle main stream must
    complete while the candidate is still working."""
    import torch.distributed as dist
    main.synchronize()
    cand.synchronize()
    e_end = torch.cuda.Event(enable_timing=True)
    e_coll = torch.cuda.Event(enable_timing=True)
    _busy(cand, buf)
    e_end.record(cand)
    with torch.cuda.stream(main):
        work = dist.all_reduce(probe, async_op=True)
        work.wait()
        e_coll.record(main)
    main.synchronize()
    cand.synchronize()
    return e_coll.elapsed_time(e_end) > 0.5


def _collective_beside_main(main, launcher, buf, probe) -> bool:
    """The pairing the candidate probes do not cover: the COMMUNICATOR's stream against the compute stream.  With the compute
    stream busy, a small all-reduce issued from the idle `launcher` stream must complete before that work ends; if it does
    not, the communicator's kernels sit in the compute stream's hardware queue, and in a training step a gradient slice's
    all-reduce (which waits for the backward-weight stream) would hold back every later kernel of the compute stream."""
    import torch.distributed as dist
    main.synchronize()
    launcher.synchronize()
    e_end = torch.cuda.Event(enable_timing=True)
    e_coll = torch.cuda.Event(enable_timing=True)
    _busy(main, buf)
    e_end.record(main)
    with torch.cuda.stream(launcher):
        work = dist.all_reduce(probe, async_op=True)
        work.wait()
        e_coll.record(launcher)
    main.synchronize()
    launcher.synchronize()
    return e_coll.elapsed_time(e_end) > 0.5


COLLECTIVES_BESIDE_MAIN: dict = {}   # device -> what the check above found (prepare())
_pool: dict = {}      # device -> streams probed by prepare() (incl. against the communicator) and not handed out yet


def prepare(device, count: int = 2) -> None:
    """Called by every rank at the same point (dist.init_from_env, after the process group exists): probe `count` side
    streams including the collective part and keep them for `concurrent_stream`."""
    dev = torch.device(device)
    if not _PROBE or not torch.cuda.is_available():
        return
    for _ in range(count):
        _pool.setdefault(dev, []).append(_probe(dev, 8, True))
    import torch.distributed as dist
    # diagnostic (set SALUN_STREAM_DEBUG on EVERY rank or on none: the check issues a collective)
    if (os.environ.get("SALUN_STREAM_DEBUG") and dist.is_available() and dist.is_initialized()
            and dist.get_backend() == "nccl" and _pool.get(dev)):
        with torch.cuda.device(dev):
            buf = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=dev).zero_()
            probe = torch.zeros(1024, dtype=torch.float32, device=dev)
            ok = _collective_beside_main(torch.cuda.current_stream(dev), _pool[dev][-1], buf, probe)
        COLLECTIVES_BESIDE_MAIN[dev] = ok
        import sys
        print(f"streams: the communicator's stream runs beside the compute stream: {ok}", file=sys.stderr)


def concurrent_stream(device, tries: int = 8) -> "torch.cuda.Stream":
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    if not _PROBE or torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device=dev)
    if _pool.get(dev):
        st = _pool[dev].pop(0)
        if os.environ.get("SALUN_STREAM_DEBUG"):
            import sys
            buf = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=dev).zero_()
            flag = torch.zeros(1, dtype=torch.float32, device=dev)
            print("streams: pooled stream beside the current stream at hand-out:",
                  _beside_main(torch.cuda.current_stream(dev), st, buf, flag), file=sys.stderr)
        return st
    return _probe(dev, tries, False)


def _probe(dev, tries: int, with_collectives: bool) -> "torch.cuda.Stream":
    import torch.distributed as dist
    dp = with_collectives and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
    main = torch.cuda.current_stream(dev)
    with torch.cuda.device(dev):
        buf = torch.empty(32 * 1024 * 1024, dtype=torch.float32, device=dev).zero_()
        flag = torch.zeros(1, dtype=torch.float32, device=dev)
        probe = None
        if dp:
            probe = torch.zeros(1024, dtype=torch.float32, device=dev)
            dist.all_reduce(probe)      # communicator set-up, if this is the process group's first collective
            main.synchronize()
        first = chosen = None
        verdicts = []   # per candidate: passed every check? (SALUN_STREAM_DEBUG)
        for _ in range(_DP_CANDIDATES if dp else tries):
            cand = torch.cuda.Stream(device=dev)
            first = first or cand
            STATS["probes"] += 1
            ok = _beside_main(main, cand, buf, flag)
            for prev in _accepted.get(dev, []):
                ok = _beside_main(prev, cand, buf, flag) and ok
            if dp:                       # every rank issues this collective for every candidate, accepted or not
                ok = _beside_collectives(main, cand, buf, probe) and ok
            verdicts.append(bool(ok))
            if ok and chosen is None:
                chosen = cand
                if not dp:
                    _accepted.setdefault(dev, []).append(cand)
                    return cand
            else:
                STATS["rejected"] += 0 if ok else 1
                _keep.append(cand)
        # nothing passed (a one-queue configuration): any stream is as good as another
        if os.environ.get("SALUN_STREAM_DEBUG"):
            import sys
            print(f"streams: probe verdicts {verdicts} (collectives={dp})", file=sys.stderr)
        if chosen is None:
            STATS["fallbacks"] = STATS.get("fallbacks", 0) + 1
            if os.environ.get("SALUN_STREAM_DEBUG"):
                import sys
                print(f"streams: NO candidate of {_DP_CANDIDATES if dp else tries} passed (collectives={dp}): taking the first", file=sys.stderr)
        _accepted.setdefault(dev, []).append(chosen or first)
        return chosen or first
