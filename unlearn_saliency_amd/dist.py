"""One process per GPU, `torch.distributed` over RCCL (backend "nccl" on ROCm) / xGMI.

The hot path has exactly two exchange steps: one all-reduce(SUM) of the
flat saliency accumulator at the end of mask generation, and one all-reduce(mean) of the
flat gradient per unlearning step.  Because parameters, gradients and the accumulator are
single flat vectors (flat.py), each is ONE collective over 44.7 MB (ResNet-18) /
154.5 MB (DDPM) / 3.44 GB (SD) — large, few messages, which is what the point-to-point
xGMI links want — instead of a bucket per tensor.  Every rank then runs the fused update
on identical inputs, so no parameter broadcast is ever needed.
CPU tests drive the same code with backend "gloo".
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def is_dist() -> bool:
    return dist.is_available() and dist.is_initialized()


def world_size() -> int:
    return dist.get_world_size() if is_dist() else 1


def rank() -> int:
    return dist.get_rank() if is_dist() else 0


def force_collectives() -> bool:
    """SALUN_FORCE_COLLECTIVES=1: create the process group and issue every collective of the data-parallel path even at
    world size 1 — how the RCCL branch (`device_id=`, ReduceOp.AVG, the async gradient buckets and their stream
    joins) is exercised on a single-GPU box (tests/test_rccl_ws1_gpu.py, `bench.py --force_collectives`)."""
    return os.environ.get("SALUN_FORCE_COLLECTIVES", "0") not in ("", "0")


def collectives_on() -> bool:
    """True when gradients / accumulators must go through the process group: more than one rank, or forced."""
    return is_dist() and (dist.get_world_size() > 1 or force_collectives())


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Initialise from torchrun's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; no-op when single-process.
    Returns (rank, local_rank, world_size) and binds this process to its GPU."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rk = int(os.environ.get("RANK", "0"))
    lrk = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(lrk % max(torch.cuda.device_count(), 1))
    if (ws > 1 or force_collectives()) and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # SALUN_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices): used by the
            # single-GPU smoke test of the data-parallel path, never for performance
            backend = os.environ.get("SALUN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
            if os.environ.get("SALUN_RCCL_HIGH_PRIORITY", "1") != "0":
                # the communicator's stream from the HIGH-priority pool: see the note behind init_process_group
                try:
                    opts = dist.ProcessGroupNCCL.Options()
                    opts.is_high_priority_stream = True
                    kw["pg_options"] = opts
                except Exception:  # a torch without the option object: default stream
                    pass
        dist.init_process_group(backend=backend, rank=rk, world_size=ws, **kw)
        # (High-priority communicator stream, round 6: the data-parallel ResNet-18 step at world size 1 read 12.5 ms instead
        # of 8.7 in 3 of 26 runs with the default stream and in 1 of 60 with this one — interleaved; later series put the
        # rate with it at ~3 %, so the evidence for the option is weak.  The slow state is not one the pairwise stream
        # probes of streams.py detect (they pass in it); it is decided at start-up, and rocprofv3 attached changes which
        # runs show it.)
        if backend == "nccl":
            # side streams (backward-weight, the diffusion steps' target pass, one spare) probed NOW, on every rank at the same
            # point, because their probe includes collectives (streams.py)
            from . import streams
            streams.prepare(kw["device_id"], 3)
    return rk, lrk, ws


def counted_ranks(device: Optional[torch.device] = None) -> int:
    """Number of ranks that took part in an actual all-reduce (SUM of ones) — what `rccl_ranks` in the bench line
    reports; 1 when single-process."""
    if not is_dist():
        return 1
    if device is None:
        device = (torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl"
                  else torch.device("cpu"))
    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(round(float(one.item())))


def launch_ranks(script: str, argv: list, nproc: int, require_devices: bool = True, extra_env: Optional[dict] = None
                 ) -> int:
    """`python script --gpus N` without torchrun: re-exec the script as N ranks of ONE node through
    `torch.distributed.run` (rendezvous on 127.0.0.1, free port), one rank per GPU.  Fails loudly when the node has
    fewer than N devices (`require_devices=False` is the CPU/gloo self-test of this launcher).  Returns the exit
    code of the launched job; the caller exits with it."""
    import socket
    import subprocess
    import sys
    if nproc < 2:
        raise ValueError("launch_ranks is for N > 1")
    if require_devices:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < nproc:
            raise SystemExit(f"{os.path.basename(script)} --gpus {nproc} needs {nproc} devices on this node, "
                             f"found {have}: refusing to run fewer ranks than asked")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // nproc)))
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    return subprocess.call(cmd, env=env)


def all_reduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    if collectives_on():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def all_reduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks.  RCCL reduces with AVG natively (no extra pass over the vector);
    gloo has no AVG, so SUM then scale."""
    ws = world_size()
    if collectives_on():
        if dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(ws)
    return flat


def barrier() -> None:
    if is_dist():
        dist.barrier()


def shard_bounds(n: int, rk: Optional[int] = None, ws: Optional[int] = None) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for this rank (ceil-sized shards, last may be short/empty)."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    per = (n + ws - 1) // ws
    return min(rk * per, n), min((rk + 1) * per, n)


def balanced_slice(b: int, rk: Optional[int] = None, ws: Optional[int] = None) -> tuple[int, int]:
    """Shard [lo, hi) of a global batch of b samples: sizes differ by at most one and no shard is empty while
    b >= world_size (ceil-sized shards leave the last ranks short or empty on a ragged tail batch)."""
    rk = rank() if rk is None else rk
    ws = world_size() if ws is None else ws
    return rk * b // ws, (rk + 1) * b // ws


def shard_loss_scale(loader) -> float:
    """Factor that turns this rank's shard-mean loss into its share of the GLOBAL batch mean under the AVG
    all-reduce of the gradients:  mean_global = (1/ws) * sum_r (n_r * ws / b) * mean_r.  1.0 when single-process or
    when the loader does not shard."""
    ws = world_size()
    sh = getattr(loader, "last_shard", None)
    if ws <= 1 or sh is None:
        return 1.0
    lo, hi, b = sh
    return (hi - lo) * ws / float(b)


def _wgrad_side_stream(device):
    """The backward-weight side stream of `device` if one was ever created (resblock.py), else None."""
    from . import resblock
    return resblock._side_streams.get(device)


class BucketedGradReducer:
    """Overlap the per-step gradient all-reduce with backward.

    The flat gradient is cut into `num_buckets` contiguous slices (equal element counts, snapped to parameter
    boundaries).  Backward fills the flat vector from its END (last layers first); a post-accumulate hook per
    parameter counts arrivals and, as soon as a slice is complete, issues `all_reduce(AVG)` on that slice with
    `async_op=True` — RCCL runs it on its own stream after the compute stream's work queued so far, while autograd
    keeps producing earlier layers' gradients.  `finish()` (called by the fused optimizer before its kernel)
    reduces whatever did not fire (unused parameters) and makes the compute stream wait for all slices.
    Few large messages (N/num_buckets x 4 B each: 11 MB for ResNet-18 with 4 buckets) are what the point-to-point
    xGMI links want; no per-tensor buckets, no parameter broadcast (replicas stay bit-identical)."""

    def __init__(self, arena, num_buckets: int = 4):
        self.arena = arena
        n = arena.n
        target = [n * (i + 1) // num_buckets for i in range(num_buckets)]
        ends, j = [], 0
        for off, k in zip(arena.offsets, arena.numels):
            if off + k >= target[j]:
                ends.append(off + k)
                while j < num_buckets - 1 and off + k >= target[j]:
                    j += 1
                if off + k >= n:
                    break
        if not ends or ends[-1] != n:
            ends.append(n)
        ends = sorted(set(ends))
        self.bounds = list(zip([0] + ends[:-1], ends))
        self.bucket_of = []
        b = 0
        for off in arena.offsets:
            while off >= self.bounds[b][1]:
                b += 1
            self.bucket_of.append(b)
        self.expected = [self.bucket_of.count(i) for i in range(len(self.bounds))]
        self.arrived = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)
        self.works = []
        self._order = []        # bucket indices in the order this step launched them (always descending)
        self._next = len(self.bounds) - 1  # the next slice allowed to go out
        self._handles = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                         for i, p in enumerate(arena._params)]

    def _make_hook(self, pidx):
        b = self.bucket_of[pidx]

        def hook(_param):
            self.arrived[b] += 1
            if self.arrived[b] == self.expected[b]:
                self._launch_ready()
        return hook

    def _launch_ready(self):
        """Slices go out in ONE order on every rank and every step — last slice first, the order backward completes
        them in — whatever order the hooks fire in: slice b is issued only once every slice above it has been.  (Ranks
        pair collectives by issue order; slices have different sizes, so two ranks issuing them in different orders
        would deadlock or mix gradients — a rank whose shard of a ragged tail batch is empty runs no backward at all and
        issues everything from `finish`.)"""
        while self._next >= 0 and self.arrived[self._next] == self.expected[self._next]:
            self._launch(self._next)
            self._next -= 1

    def _launch(self, b):
        lo, hi = self.bounds[b]
        sl = self.arena.grads[lo:hi]
        self.launched[b] = True
        self._order.append(b)
        if dist.get_backend() == "nccl":
            # The weight-gradient kernels of this slice may still be queued on the backward-weight side stream
            # (resblock.py / conv.py / conv_bf16.py overlap them with backward-data).  The collective must wait for them —
            # but the MAIN stream must not: a join per block (rounds 1 - 5) cost the data-parallel step 17 % at world size
            # 1 (10.0 vs 8.5 ms).  So the collective is issued FROM THE SIDE STREAM, after that stream has been told to
            # wait for the main stream's work so far (the normalisation / bias gradients of the slice; the side stream lags
            # the main one anyway and does the same wait before every backward-weight launch): RCCL orders its own
            # stream behind the stream it is called on, the side stream goes on with the next layers' kernels.
            # (Until late in round 6 a separate launch stream waited for both.  One more busy hardware queue: with it the
            # step was 8.65 ms, and 9.7 ms in EVERY run once the stream probes created 24 instead of 12 candidate streams;
            # from the side stream 8.60 ms in both.  A rare 12.4 ms state — ~3 % of processes, decided at start-up, cause
            # unknown — exists with either form: profiles/r06_dp_outliers.txt.)
            dev = sl.device
            side = _wgrad_side_stream(dev)
            if side is None:
                self.works.append((dist.all_reduce(sl, op=dist.ReduceOp.AVG, async_op=True), None))
            else:
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.works.append((dist.all_reduce(sl, op=dist.ReduceOp.AVG, async_op=True), None))
        else:
            if sl.is_cuda:  # gloo stages device tensors through the host behind the CURRENT stream only
                side = _wgrad_side_stream(sl.device)
                if side is not None:
                    torch.cuda.current_stream(sl.device).wait_stream(side)
            self.works.append((dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True), sl))

    def finish(self):
        # whatever backward did not complete (unused parameters; no backward at all on a rank with an empty shard) is
        # issued now, continuing the same descending order
        late = 0
        while self._next >= 0:
            self._launch(self._next)
            self._next -= 1
            late += 1
        if late > 1 and any(self.arrived) and not getattr(self, "_warned", False):
            # a backward ran, yet more than the head slice was left for finish(): a parameter that never receives a
            # gradient (or receives several: arrived != expected, deliberately not launched early — the slice may
            # still be written) holds back every slice below it.  Correct, but the overlap with backward is lost.
            self._warned = True
            import warnings
            stuck = [b for b in range(len(self.bounds)) if self.arrived[b] != self.expected[b]]
            warnings.warn(f"BucketedGradReducer: {late} of {len(self.bounds)} gradient slices were reduced after "
                          f"backward (slices with arrivals != parameters: {stuck}); the all-reduce no longer overlaps "
                          "backward")
        assert self._order == list(range(len(self.bounds) - 1, -1, -1)), self._order
        ws = world_size()
        for work, needs_div in self.works:
            work.wait()
            if needs_div is not None:
                needs_div.div_(ws)
        self.works.clear()
        self._order = []
        self._next = len(self.bounds) - 1
        self.arrived = [0] * len(self.bounds)
        self.launched = [False] * len(self.bounds)

    def remove(self):
        for h in self._handles:
            h.remove()
