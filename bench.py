#!/usr/bin/env python
"""bench.py — unlearn steps/sec (+ mask-gen seconds) for ResNet-18 / CIFAR-10, 10 %-random forget.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: ResNet-18 (11,173,962 params) on a CIFAR-shaped synthetic set
(45,000 x 32x32x3 uint8 from the counter-based generator, 4,500 forget samples drawn with
RandomState(1) like `--seed 2`), batch 256 per GPU, fp32, SalUn mask ratio 0.5,
SGD lr 0.013 / momentum 0.9 / wd 5e-4 (Classification/README.md:32-35).

A "step" is one unlearning step of the reference's RL loop (Classification/unlearn/RL.py:123-140):
device-side batch assembly (gather + RandomCrop + flip + /255) -> forward -> CE (random labels on forget
batches, true labels on retain batches, in the reference's 18:159 proportion) -> backward into the flat
gradient -> [N>1: RCCL all-reduce of the flat gradient] -> ONE fused masked SGD-momentum launch.
Nothing is skipped inside the timed region; inputs are resident in HBM before it starts.

`value` = (steps x ranks) / seconds: every rank processes its own 256-sample batch per step (weak scaling).
Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the hand-written kernel that owns the optimizer tail, salun_masked_sgd_step: algorithmic
                bytes (21 B x N) / mean launch duration (HIP events on the launch stream, inside the timed steps)
  cpu_baseline  the un-fused reference op sequence (oracle/torch_ref.py) timed on this host's cores, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N18 = 11_173_962
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_MATRIX_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 MFMA / vector peak
FWD_BWD_GFLOP_PER_IMG = 3.329  # SURVEY.md §6 (FlopCounterMode on the reference ResNet-18)
SGD_BYTES_PER_ELEM = 21      # r p,g,buf (12) + r mask (1) + w p,buf (8)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=177,
                    help="timed steps; the default is ONE epoch of the reference's RL loop (18 forget + 159 retain "
                         "batches), so both ragged tail batches (148 / 52 samples) are inside the window")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "ddpm"],
                    help="resnet18 = BASELINE configs[1] (the headline metric); ddpm = configs[3] at 1 GPU "
                         "(tools/bench_ddpm.py's line with the same contract fields)")
    ap.add_argument("--selftest_launcher", action="store_true",
                    help="host-only check of the --gpus N launcher: N ranks rendezvous (gloo on CPU, RCCL on GPUs), "
                         "all-reduce a one and rank 0 prints {n_gpus, rccl_ranks}; no workload is run")
    ap.add_argument("--batch_size", type=int, default=256, help="per-GPU batch (reference: 256)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_steps", type=int, default=5, help="reference-sequence steps timed on the host CPU")
    ap.add_argument("--no_mask_gen", action="store_true", help="skip timing Phase A (a random mask is used)")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="1 = keep cudnn.deterministic=True as the reference's setup_seed sets it (restricts MIOpen's "
                         "algorithm choice); 0 = let MIOpen pick its fastest fp32 kernels (same math, fp32)")
    ap.add_argument("--channels_last", type=int, default=0, help="1 = NHWC activations/weights")
    ap.add_argument("--fused_bn", type=int, default=1,
                    help="1 = BatchNorm(+residual)+ReLU as fused kernels (csrc/salun_norm.hip); 0 = PyTorch-ROCm ops")
    ap.add_argument("--salun_conv", type=int, default=1,
                    help="1 = convolutions on the hand-written fp32 MFMA kernels (csrc/salun_conv.hip); "
                         "0 = library (MIOpen) convolutions")
    return ap.parse_args()


class StepStream:
    """Endless stream of (image, target, loss weight of this rank's shard) batches in the reference's epoch pattern:
    all forget batches (random labels) then all retain batches (true labels), reshuffled every epoch."""

    def __init__(self, forget_loader, retain_loader, num_classes=10):
        self.fl, self.rl, self.nc = forget_loader, retain_loader, num_classes

    def __iter__(self):
        from unlearn_saliency_amd import dist as sdist
        while True:
            for x, y in self.fl:
                lo, hi, b = self.fl.last_shard  # labels drawn for the global batch, sliced to this rank's shard
                lab = torch.randint(0, self.nc, (b,))[lo:hi]
                yield x, lab.to(y.device, non_blocking=True), sdist.shard_loss_scale(self.fl)
            for x, y in self.rl:
                yield x, y, sdist.shard_loss_scale(self.rl)


def build_workload(device, rank, world, per_gpu_bs):
    from unlearn_saliency_amd.Classification.dataset import (ArrayDataset, BatchLoader, TRAIN_TRANSFORM,
                                                             replace_class, split_marked, synthetic_cifar10)
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.Classification import utils

    (xtr, ytr), _ = synthetic_cifar10()
    rs = np.random.RandomState(2)  # --seed 2: stratified 10 % validation split (dataset.py:576-593)
    valid = np.hstack([rs.choice(np.where(ytr == c)[0], 500, replace=False) for c in range(10)])
    keep = np.asarray(sorted(set(range(len(xtr))) - set(valid.tolist())))
    train = ArrayDataset(xtr[keep], ytr[keep].copy(), TRAIN_TRANSFORM)
    replace_class(train, -1, num_indexes_to_replace=4500, seed=1, only_mark=True)  # seed-1 (dataset.py:599-606)
    forget, retain = split_marked(train)
    assert len(forget) == 4500 and len(retain) == 40500
    utils.setup_seed(1)  # --train_seed 1: Kaiming init (utils.py:134-143)
    model = model_dict["resnet18"](num_classes=10).to(device)
    utils.setup_seed(2)
    gbs = per_gpu_bs * world  # each global batch is sharded contiguously over ranks
    mk = lambda ds: BatchLoader(ds, gbs, True, device_resident=True, device=device, rank=rank, world_size=world)
    return model, mk(forget), mk(retain)


def time_mask_gen(model, forget_loader, criterion):
    """Phase A wall time on this GPU: 4,500 forget samples fwd/bwd + flat accumulation + all 10 thresholds
    (u8 masks resident in HBM).  File writing (10 x 89 MB int64 .pt) is reported separately by generate_mask.py."""
    from unlearn_saliency_amd.Classification.generate_mask import (THRESHOLD_LIST, accumulate_saliency,
                                                                  masks_from_saliency)
    for _ in range(2):  # first pass warms MIOpen's find-db
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = accumulate_saliency(forget_loader, model, criterion)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        masks = masks_from_saliency(acc, THRESHOLD_LIST)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return masks[0.5], {"total_sec": t2 - t0, "saliency_sec": t1 - t0, "topk_10_thresholds_sec": t2 - t1}


def cpu_baseline(per_gpu_bs, steps):
    """The reference's op sequence for one RL step (fwd/bwd, per-tensor mask multiply, torch SGD, per-tensor
    restore; oracle/torch_ref.py) on the host cores, bounded to `steps` steps after one warm-up step."""
    from oracle import torch_ref
    from unlearn_saliency_amd.Classification.models import model_dict
    torch.manual_seed(1)
    model = model_dict["resnet18"](num_classes=10)
    model.train()
    crit = nn.CrossEntropyLoss()
    opt = torch.optim.SGD(model.parameters(), 0.013, momentum=0.9, weight_decay=5e-4)
    mask = {n: (torch.rand_like(p) < 0.5).to(torch.int64) for n, p in model.named_parameters()}
    theta0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    x = torch.rand(per_gpu_bs, 3, 32, 32)
    y = torch.randint(0, 10, (per_gpu_bs,))
    torch_ref.rl_step_cpu(model, crit, opt, x, y, mask, theta0)  # warm-up (buffers, thread pool)
    timers = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        torch_ref.rl_step_cpu(model, crit, opt, x, y, mask, theta0, timers)
    dt = time.perf_counter() - t0
    # the reference's Phase-A ranking (generate_mask.py:46-80: abs, cat, two argsorts of N,
    # per-tensor compare) for ONE ratio on a ResNet-18-sized gradient dict; the reference repeats it for 10 ratios
    g = torch.Generator().manual_seed(3)
    grads = {n: torch.randn(p.shape, generator=g) * 1e-3 for n, p in model.named_parameters()}
    t1 = time.perf_counter()
    torch_ref.masks_from_gradients_cpu(grads, [0.5])
    mask_one = time.perf_counter() - t1
    return {"value": steps / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "mask_topk_sec_one_ratio": mask_one, "mask_topk_sec_10_ratios_extrapolated": 10 * mask_one,
            "mask_topk_sample": "abs + cat + 2 x argsort(11,173,962) + 62 per-tensor compares for ratio 0.5 "
                                "(Classification/generate_mask.py:46-80); x10 for the reference's ten ratios",
            "sample": f"{steps} RL steps at batch {per_gpu_bs} (ResNet-18 fp32, reference op sequence: fwd+bwd, "
                      f"62x mask-mul, torch.optim.SGD, 62x restore) after 1 warm-up step",
            "ms_per_step": 1e3 * dt / steps, "host_cpu_count": os.cpu_count(),
            "breakdown_ms": {k: 1e3 * v / steps for k, v in timers.items()}}


def selftest_launcher(a):
    """`--gpus N --selftest_launcher`: prove that N ranks were created and can reduce (gloo on a CPU box)."""
    from unlearn_saliency_amd import dist as sdist
    rank, _, world = sdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher created WORLD_SIZE={world}")
    counted = sdist.counted_ranks()
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "rccl_ranks": counted,
                          "backend": torch.distributed.get_backend() if sdist.is_dist() else None}), flush=True)
    sdist.barrier()
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


def ensure_built():
    """A fresh clone has no libsalun.so: build it (hipcc cross-compiles) instead of dying with ImportError."""
    from unlearn_saliency_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        with __import__("contextlib").redirect_stdout(sys.stderr):
            __graft_entry__.build()
    return _lib.lib()  # raises loudly if the HIP extension still cannot be loaded


def main():
    a = parse()
    from unlearn_saliency_amd import dist as sdist
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without torchrun: become the launcher of N ranks (one per GPU, RCCL)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            ensure_built() if not a.selftest_launcher else None
        raise SystemExit(sdist.launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus,
                                            require_devices=not a.selftest_launcher))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} disagrees with the launcher's WORLD_SIZE={os.environ['WORLD_SIZE']}")
    if a.selftest_launcher:
        return selftest_launcher(a)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback for the measured path)")
    if torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", "1")):
        raise SystemExit(f"bench.py --gpus {a.gpus} needs {a.gpus} devices, found {torch.cuda.device_count()}")
    if int(os.environ.get("RANK", "0")) == 0:
        ensure_built()
    if a.workload == "ddpm":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_ddpm
        return bench_ddpm.main(["--steps", str(a.steps), "--warmup", str(a.warmup)]
                               + (["--no_cpu_baseline"] if a.no_cpu_baseline else []))
    rank, local_rank, world = sdist.init_from_env()
    from unlearn_saliency_amd import _lib
    _lib.lib()  # fail loudly if the HIP extension is missing
    rccl_ranks = sdist.counted_ranks()  # an actual all-reduce over the ranks RCCL sees
    assert rccl_ranks == world == a.gpus, (rccl_ranks, world, a.gpus)
    device = torch.device("cuda", torch.cuda.current_device())
    torch.backends.cudnn.benchmark = True

    from unlearn_saliency_amd.Classification.unlearn.impl import FusedMaskedSGD
    from unlearn_saliency_amd.flat import arena_of
    from unlearn_saliency_amd import ops

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # stdout carries exactly one JSON line
        model, forget_loader, retain_loader = build_workload(device, rank, world, a.batch_size)
    torch.backends.cudnn.deterministic = bool(a.deterministic)  # setup_seed() above turned it on
    torch.backends.cudnn.benchmark = True
    if a.channels_last:
        model = model.to(memory_format=torch.channels_last)
    n_salun_convs = 0
    if a.salun_conv and not a.channels_last:
        from unlearn_saliency_amd.conv import use_salun_convs
        n_salun_convs = use_salun_convs(model)
    n_fused_bn = 0
    if a.fused_bn and not a.channels_last:
        from unlearn_saliency_amd.norm import use_fused_bn
        n_fused_bn = use_fused_bn(model)
    criterion = nn.CrossEntropyLoss()
    arena = arena_of(model)
    assert arena.n == N18

    mask_gen = None
    if a.no_mask_gen:
        mask_u8 = ops.mask_topk(ops.fill_normal(N18, 5, 0.0, 1e-3), [int(N18 * 0.5)])[0]
    else:
        with contextlib.redirect_stdout(sys.stderr):
            mask_u8, mask_gen = time_mask_gen(model, forget_loader, criterion)
    assert ops.mask_popcount(mask_u8) == int(N18 * 0.5)

    opt = FusedMaskedSGD(arena, 0.013, momentum=0.9, weight_decay=5e-4)
    opt.set_mask(mask_u8)
    model.train()
    stream = iter(StepStream(forget_loader, retain_loader))

    def one_step(ev=None):
        x, y, w = next(stream)
        if a.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        loss = criterion(model(x), y)
        if w != 1.0:  # data parallel, ragged tail batch: count-weighted shard mean
            loss = loss * w
        opt.zero_grad()
        loss.backward()
        if ev is not None:
            ev[0].record()
        opt.step()
        if ev is not None:
            ev[1].record()

    # untimed preparation: both uint8 sets resident in HBM, and one step at each ragged tail-batch size of the
    # reference's epoch (4500 % 256 = 148 forget, 40500 % 256 = 52 retain) so shape-specialised library state
    # (BN / GEMM heuristics) exists before the clock starts
    for ld in (forget_loader, retain_loader):
        ld._resident()
    for tail in sorted({len(forget_loader.dataset) % (a.batch_size * world) // world,
                        len(retain_loader.dataset) % (a.batch_size * world) // world} - {0}):
        xs = torch.rand(tail, 3, 32, 32, device=device)
        ys = torch.randint(0, 10, (tail,), device=device)
        opt.zero_grad()
        criterion(model(xs), ys).backward()
        arena.zero_grad()  # discard: no parameter update from the shape warm-up
    for _ in range(a.warmup):
        one_step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    torch.cuda.synchronize()
    sdist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(events[i])
    torch.cuda.synchronize()
    sdist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # the optimizer tail on the launch stream: at N=1 exactly one kernel (salun_masked_sgd_step)
    tail_ms = sorted(s.elapsed_time(e) for s, e in events)
    tail_mean_s = 1e-3 * sum(tail_ms) / len(tail_ms)
    # per-step device time from consecutive event timestamps (shows clock/thermal drift over the run)
    step_ms = [events[i][0].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]

    # forward+backward device time per step: from the end of the previous step's update to the start of this one's
    fb_ms = [events[i][1].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]
    fb_mean_s = 1e-3 * sum(fb_ms) / max(len(fb_ms), 1)
    pmc_traffic, pmc_src = None, None
    try:  # HBM bytes per launch from the committed PMC passes (tools/pmc.sh; never collected inside this run)
        for rnd in ("r02", "r01"):
            pth = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
            if os.path.exists(pth):
                with open(pth) as f:
                    kern = json.load(f)["kernels"]
                pmc_traffic = (kern.get(f"k_masked_sgd_vec@{rnd}_n18") or kern["k_masked_sgd_vec@n18"])["traffic_bytes"]
                pmc_src = (f"profiles/{rnd}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate "
                           f"passes over tools/kbench.py, tools/pmc.sh; PMC cannot be collected inside this run)")
                break
    except Exception:
        pass

    from unlearn_saliency_amd import conv as sconv
    library_conv_calls = dict(sconv.LIBRARY_CONV_CALLS, total=sconv.library_conv_calls())
    if rank == 0:
        steps_per_s = a.steps * world / dt
        alg_bytes = SGD_BYTES_PER_ELEM * N18
        out = {
            "metric": "unlearn_steps_per_sec (ResNet-18/CIFAR-10 10%-forget, RL + SalUn mask, batch 256/GPU)",
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "backend": (torch.distributed.get_backend() if sdist.is_dist() else "single-process"),
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ResNet-18 (11,173,962 params) / CIFAR-10-shaped synthetic set, 10% random-data "
                                   "forget (4,500 of 45,000), RL unlearning step with SalUn mask ratio 0.5, "
                                   "SGD lr 0.013 momentum 0.9 wd 5e-4, RandomCrop+flip on device",
                       "per_gpu_batch": a.batch_size, "global_batch": a.batch_size * world,
                       "parallelism": f"dp{world}", "params": N18, "cudnn_deterministic": bool(a.deterministic),
                       "channels_last": bool(a.channels_last), "salun_mfma_convs": n_salun_convs,
                       "fused_bn_layers": n_fused_bn, "library_conv_calls": library_conv_calls},
            "samples_per_sec": steps_per_s * a.batch_size,
            "mask_gen_sec": None if mask_gen is None else mask_gen["total_sec"],
            "mask_gen": mask_gen,
            "step_ms_trend": {"first5": [round(v, 2) for v in step_ms[:5]], "last5": [round(v, 2) for v in step_ms[-5:]],
                              "min": round(min(step_ms), 2) if step_ms else None,
                              "max": round(max(step_ms), 2) if step_ms else None},
            "roofline": {"kernel": "salun_masked_sgd_step" + ("" if world == 1 else " (+ flat-gradient all-reduce)"),
                         "bound": "hbm", "achieved": alg_bytes / tail_mean_s / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_bytes / tail_mean_s / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic,
                         "traffic_source": pmc_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_us": tail_mean_s * 1e6,
                         "median_launch_us": 1e3 * tail_ms[len(tail_ms) // 2],
                         "timing": "HIP events on the launch stream around the launch, inside the timed steps"},
            "fwd_bwd": {"bound": "mfma", "gflop_per_step": FWD_BWD_GFLOP_PER_IMG * a.batch_size,
                        "achieved": FWD_BWD_GFLOP_PER_IMG * a.batch_size / (fb_mean_s * 1e3) if fb_mean_s else None,
                        "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                        "frac": (FWD_BWD_GFLOP_PER_IMG * a.batch_size / (fb_mean_s * 1e3) / FP32_MATRIX_PEAK_TF)
                        if fb_mean_s else None,
                        "mean_fwd_bwd_ms": fb_mean_s * 1e3,
                        "timing": "HIP events: end of step i's update -> start of step i+1's update",
                        "note": ("convolutions: hand-written fp32 MFMA implicit-GEMM kernels (salun_conv2d_*); "
                                 + ("BN(+add)+ReLU: fused kernels (salun_bn_*); pool/fc/CE: PyTorch-ROCm" if n_fused_bn
                                   else "BN/ReLU/pool/fc: PyTorch-ROCm")) if n_salun_convs else
                                "convolutions/GEMMs run in MIOpen/rocBLAS fp32 through PyTorch-ROCm"},
        }
        if world == 1 and not a.no_cpu_baseline:
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_baseline(a.batch_size, a.cpu_steps)
        print(json.dumps(out), flush=True)
    sdist.barrier()
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
