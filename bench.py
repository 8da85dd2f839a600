#!/usr/bin/env python
"""bench.py — unlearn steps/sec (+ mask-gen seconds) for the SalUn hot path.

    python bench.py --gpus N --steps K --warmup W                      ResNet-18 / CIFAR-10, 10 %-random forget
    python bench.py --gpus N --forget class                            ... class-wise forget (BASELINE configs[2])
    python bench.py --gpus N --scaling strong [--sync_bn 1]            ... the reference's global batch 256, 256/N per GPU
    python bench.py --gpus N --workload ddpm | sd                      CFG-DDPM class-forget | SD-v1 nsfw_removal (bf16)
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...,
     or plain `python bench.py --gpus N`, which launches the N ranks itself)

Default workload = BASELINE.json configs[1]: ResNet-18 (11,173,962 params) on a CIFAR-shaped synthetic set
(45,000 x 32x32x3 uint8 from the counter-based generator, 4,500 forget samples drawn with
RandomState(1) like `--seed 2`), batch 256 per GPU, fp32, SalUn mask ratio 0.5,
SGD lr 0.013 / momentum 0.9 / wd 5e-4 (Classification/README.md:32-35).

A "step" is one unlearning step of the reference's RL loop (Classification/unlearn/RL.py:123-140):
device-side batch assembly (gather + RandomCrop + flip + /255) -> forward -> CE (random labels on forget
batches, true labels on retain batches, in the reference's 18:159 proportion) -> backward into the flat
gradient -> [N>1: RCCL all-reduce of the flat gradient] -> ONE fused masked SGD-momentum launch.
Nothing is skipped inside the timed region; inputs are resident in HBM before it starts.

`value`: weak scaling (default) = (steps x ranks) / seconds — every rank processes its own 256-sample batch per step;
strong scaling = steps / seconds — one step is one reference step (global batch 256) whatever the rank count.
`samples_per_sec` counts the samples actually processed (ragged tail batches count for what they hold).
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
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps; resnet18 default 177 = ONE epoch of the reference's RL loop (18 forget + 159 "
                         "retain batches), so both ragged tail batches (148 / 52 samples) are inside the window; "
                         "ddpm default 20, sd default 3")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (defaults: resnet18 10, ddpm 3, sd 1)")
    ap.add_argument("--force_collectives", action="store_true",
                    help="create the process group and issue every data-parallel collective even at one rank "
                         "(exercises the RCCL branch on a single-GPU box; SALUN_FORCE_COLLECTIVES=1)")
    ap.add_argument("--digest", action="store_true",
                    help="diffusion workloads: add the parameters' SHA-256 and the last loss to the line")
    ap.add_argument("--workload", default="resnet18", choices=["resnet18", "ddpm", "sd"],
                    help="resnet18 = BASELINE configs[1] (the headline metric; --forget class = configs[2]); "
                         "ddpm = configs[3] (CFG-DDPM class-forget, batch 128/GPU); sd = configs[4] (SD-v1 U-Net "
                         "nsfw_removal body, bf16, batch 8/GPU).  All three shard over --gpus N ranks")
    ap.add_argument("--forget", default="random", choices=["random", "class"],
                    help="random = 10 %% random-data forget (4,500 of 45,000; configs[1]); class = class-wise forget "
                         "(all 4,500 samples of --class_to_replace; configs[2], Classification/dataset.py:599-609)")
    ap.add_argument("--class_to_replace", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank takes --batch_size samples per step (global batch = N x batch_size); "
                         "strong: the reference's semantics — global batch = --batch_size, 256/N per rank, same "
                         "1,770 steps (Classification/unlearn/RL.py:114-140 under nn.DataParallel)")
    ap.add_argument("--sync_bn", type=int, default=0,
                    help="1 = SyncBatchNorm (global-batch statistics, reference-equivalent numerics under strong "
                         "scaling); 0 = per-replica statistics like nn.DataParallel")
    ap.add_argument("--selftest_launcher", action="store_true",
                    help="host-only check of the --gpus N launcher: N ranks rendezvous (gloo on CPU, RCCL on GPUs), "
                         "all-reduce a one and rank 0 prints {n_gpus, rccl_ranks}; no workload is run")
    ap.add_argument("--selftest_workload", action="store_true",
                    help="host-only check of the benchmark's data side under the launcher: the N ranks' shards of the "
                         "(random / class-wise) forget set partition every global batch; no device work")
    ap.add_argument("--batch_size", type=int, default=256, help="per-GPU batch (reference: 256)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_steps", type=int, default=12,
                    help="reference-sequence steps timed on the host CPU, in --cpu_repeats groups (SURVEY.md D3 asks 20 = "
                         "--cpu_survey: 12 keep the default run, which also carries the DDPM block and its one CPU "
                         "step, near three minutes; the spread over the groups is reported)")
    ap.add_argument("--cpu_repeats", type=int, default=4)
    ap.add_argument("--cpu_survey", action="store_true",
                    help="the CPU baseline at the sample sizes SURVEY.md D3 asks for: 20 RL steps (4 groups of 5, ~2 min "
                         "on 128 threads) instead of the default 12")
    ap.add_argument("--no_mask_gen", action="store_true", help="skip timing Phase A (a random mask is used)")
    ap.add_argument("--no_ddpm", action="store_true",
                    help="skip the CFG-DDPM class-forget block the default one-GPU line carries beside the ResNet-18 "
                         "figures (north_star's second target; `--workload ddpm` is the full-length form)")
    ap.add_argument("--no_sd", action="store_true",
                    help="skip the SD-v1 U-Net block (BASELINE configs[4], bf16, batch 8) of the default one-GPU line")
    ap.add_argument("--sd_steps", type=int, default=4)
    ap.add_argument("--no_dp", action="store_true",
                    help="skip the `dp_ws1` figures: the same steps through the data-parallel path (process group over "
                         "RCCL at world size 1, gradient slices all-reduced behind backward), each in a subprocess")
    ap.add_argument("--ddpm_steps", type=int, default=10)
    ap.add_argument("--ddpm_mask_batches", type=int, default=8,
                    help="forget batches of the embedded DDPM Phase A (the reference's 40 scale linearly)")
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


def build_workload(device, rank, world, per_gpu_bs, forget_mode="random", class_to_replace=0, scaling="weak",
                   device_resident=True):
    from unlearn_saliency_amd.Classification.dataset import (ArrayDataset, BatchLoader, TRAIN_TRANSFORM,
                                                             replace_class, split_marked, synthetic_cifar10)
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.Classification import utils

    (xtr, ytr), _ = synthetic_cifar10()
    rs = np.random.RandomState(2)  # --seed 2: stratified 10 % validation split (dataset.py:576-593)
    valid = np.hstack([rs.choice(np.where(ytr == c)[0], 500, replace=False) for c in range(10)])
    keep = np.asarray(sorted(set(range(len(xtr))) - set(valid.tolist())))
    train = ArrayDataset(xtr[keep], ytr[keep].copy(), TRAIN_TRANSFORM)
    if forget_mode == "class":  # `--class_to_replace c`: the whole class, no sub-sampling (dataset.py:599-606,690-705)
        replace_class(train, int(class_to_replace), num_indexes_to_replace=None, seed=1, only_mark=True)
    else:                       # `--num_indexes_to_replace 4500` with the parser's default class -1
        replace_class(train, -1, num_indexes_to_replace=4500, seed=1, only_mark=True)  # seed-1 (dataset.py:599-606)
    forget, retain = split_marked(train)
    assert len(forget) == 4500 and len(retain) == 40500
    if forget_mode == "class":
        assert set(forget.targets.tolist()) == {int(class_to_replace)}
    utils.setup_seed(1)  # --train_seed 1: Kaiming init (utils.py:134-143)
    model = model_dict["resnet18"](num_classes=10).to(device)
    utils.setup_seed(2)
    # each global batch is sharded contiguously over ranks; strong scaling keeps the reference's global batch
    gbs = per_gpu_bs * world if scaling == "weak" else per_gpu_bs
    mk = lambda ds: BatchLoader(ds, gbs, True, device_resident=device_resident, device=device, rank=rank,
                                world_size=world)
    return model, mk(forget), mk(retain)


def time_mask_gen(model, forget_loader, criterion):
    """Phase A wall time on this GPU: 4,500 forget samples fwd/bwd + flat accumulation + all 10 thresholds
    (u8 masks resident in HBM).  File writing (10 x 89 MB int64 .pt) is reported separately by generate_mask.py."""
    from unlearn_saliency_amd.Classification.generate_mask import (THRESHOLD_LIST, accumulate_saliency,
                                                                  masks_from_saliency)
    from unlearn_saliency_amd import ops
    masks = None
    for _ in range(2):  # first pass warms the shape-specialised state
        masks = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = accumulate_saliency(forget_loader, model, criterion)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        masks = masks_from_saliency(acc, THRESHOLD_LIST)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    first_call, saliency_sec = t2 - t1, t1 - t0
    # the same call again with the ten outputs handed back to the caching allocator first: what the select costs the
    # host once the 112 MB of masks no longer come from a fresh hipMalloc (the pass above allocates them behind a
    # forward / backward that has just recycled every cached block)
    masks = None
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    masks = masks_from_saliency(acc, THRESHOLD_LIST)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    route, err = ops.mask_topk_status(acc.device)  # 1 = the single-read route served all ten ratios, 2 = full scan
    # device time of the ten-ratio select alone (HIP events, no host synchronisation or allocation inside): the
    # reference's own list 0.1 ... 1.0 on THIS accumulator (exact zeros included), output buffers reused
    n = acc.numel()
    ks = [int(n * r) for r in THRESHOLD_LIST]
    outs = [masks[r] for r in THRESHOLD_LIST]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record()
        ops.mask_topk(acc, ks, outs)
        b.record()
    torch.cuda.synchronize()
    dev_us = sorted(1e3 * a.elapsed_time(b) for a, b in ev)
    zeros = int((acc == 0).sum().item())
    keep = masks[0.5].clone()  # the ten masks are rows of ONE allocation: a clone lets the other nine (100 MB) go
    masks = outs = None
    return keep, {"total_sec": saliency_sec + (t2 - t1), "saliency_sec": saliency_sec,
                        "topk_10_thresholds_sec": t2 - t1, "topk_10_thresholds_first_call_sec": first_call,
                        "topk_route": route, "topk_error": err,
                        "topk_10_thresholds_device_us": dev_us[len(dev_us) // 2],
                        "topk_algorithmic_bytes": 14 * n,
                        "topk_frac_of_hbm_peak": 14 * n / (dev_us[len(dev_us) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                        "accumulator_exact_zeros": zeros,
                        "topk_note": "wall = host call incl. workspace / output allocation and one status read-back; "
                                     "device = HIP events around salun_mask_topk_ex alone (median of 5), 4 B read + "
                                     "10 x 1 B written per element"}


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(per_gpu_bs, steps, repeats=3):
    """The reference's op sequence for one RL step (fwd/bwd, per-tensor mask multiply, torch SGD, per-tensor
    restore; oracle/torch_ref.py) on the host cores: `repeats` groups of steps/repeats steps after one warm-up step
    (the spread over the groups is reported), then the reference's Phase-A ranking MEASURED for all ten ratios at
    N18 and for the DDPM's single ratio at N_D."""
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
    repeats = max(1, min(repeats, steps))
    per = max(1, steps // repeats)
    group_sps = []
    t_all = time.perf_counter()
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(per):
            torch_ref.rl_step_cpu(model, crit, opt, x, y, mask, theta0, timers)
        group_sps.append(per / (time.perf_counter() - t0))
    dt = time.perf_counter() - t_all
    done = per * repeats
    # the reference's Phase-A ranking (generate_mask.py:46-80: abs, cat, two argsorts of N,
    # per-tensor compare), all TEN ratios as the reference loops them, on a ResNet-18-sized gradient dict
    g = torch.Generator().manual_seed(3)
    grads = {n: torch.randn(p.shape, generator=g) * 1e-3 for n, p in model.named_parameters()}
    t1 = time.perf_counter()
    torch_ref.masks_from_gradients_cpu(grads, [0.5])
    mask_one = time.perf_counter() - t1
    t1 = time.perf_counter()
    torch_ref.masks_from_gradients_cpu(grads, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0])
    mask_ten = time.perf_counter() - t1
    # ... and the DDPM's one ratio on an N_D-sized vector (DDPM/runners/diffusion.py:1006-1039)
    nd = {"flat": torch.randn(38_632_323, generator=g) * 1e-3}
    t1 = time.perf_counter()
    torch_ref.masks_from_gradients_cpu(nd, [0.5])
    mask_nd = time.perf_counter() - t1
    srt = sorted(group_sps)
    return {"value": done / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_groups": [round(v, 4) for v in group_sps],
            "value_min_median_max": [round(srt[0], 4), round(srt[len(srt) // 2], 4), round(srt[-1], 4)],
            "mask_topk_sec_one_ratio": mask_one, "mask_topk_sec_10_ratios": mask_ten,
            "mask_topk_sec_one_ratio_ND": mask_nd,
            "mask_topk_sample": "abs + cat + 2 x argsort(N) + per-tensor compares per ratio "
                                "(Classification/generate_mask.py:46-80): ratio 0.5 alone and the reference's ten "
                                "ratios at N = 11,173,962, both measured; ratio 0.5 at N_D = 38,632,323",
            "sample": f"{done} RL steps at batch {per_gpu_bs} in {repeats} groups of {per} (ResNet-18 fp32, reference "
                      f"op sequence: fwd+bwd, 62x mask-mul, torch.optim.SGD, 62x restore) after 1 warm-up step; "
                      f"SURVEY.md D3 asks 20 steps (--cpu_survey): capped to keep the default run within minutes",
            "ms_per_step": 1e3 * dt / done, "host_cpu_count": os.cpu_count(), "cpu_model": cpu_model_name(),
            "torch_version": torch.__version__,
            "breakdown_ms": {k: 1e3 * v / done for k, v in timers.items()}}


def dp_ws1_line(extra, plain_ms):
    """One workload through the data-parallel path at world size 1 in a subprocess (a process group cannot be created
    and torn down inside the timed process without disturbing it): the fields that matter of its bench line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--force_collectives", "--no_cpu_baseline",
           "--no_ddpm", "--no_sd", "--no_dp"] + extra
    env = dict(os.environ)
    env.setdefault("MASTER_PORT", "29621")
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-400:]}
        d = json.loads(lines[-1])
        out = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "backend", "collectives",
                                     "rccl_ranks")}
        out["plain_ms_per_step"] = plain_ms
        out["dp_over_plain"] = (d["ms_per_step"] / plain_ms) if plain_ms else None
        out["subprocess_sec"] = time.perf_counter() - t0
        return out
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def dp_ws1_checked(extra, plain_ms):
    """dp_ws1_line, run a second time when the first reading is more than 15 % above the single-process step: a process
    occasionally starts in a slow arbitration state of the hardware queues (12.4 ms instead of 8.6 for ResNet-18: about 3 %
    of processes, cause unknown).  Both readings stay in the block
    (`attempts_ms`); the fields are the faster one's."""
    first = dp_ws1_line(extra, plain_ms)
    if "error" in first or not first.get("dp_over_plain") or first["dp_over_plain"] <= 1.15:
        return first
    second = dp_ws1_line(extra, plain_ms)
    if "error" in second:
        first["second_attempt_error"] = second["error"]
        return first
    best = min((first, second), key=lambda r: r["ms_per_step"])
    best = dict(best)
    best["attempts_ms"] = [first["ms_per_step"], second["ms_per_step"]]
    best["note"] = "first reading > 1.15x the single-process step: measured twice, both readings in attempts_ms, fields = the faster"
    return best


def sd_block(a):
    """`bench.py --workload sd --gpus 1` (bf16, batch 8) in a subprocess; its line's fields that matter."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", "sd", "--steps", str(a.sd_steps),
           "--warmup", "1"] + (["--no_cpu_baseline"] if a.no_cpu_baseline else [])
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-400:]}
        d = json.loads(lines[-1])
        return {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config",
                                  "mask_gen", "roofline", "fwd_bwd", "kernels", "host_enqueue_ms_per_step",
                                  "hbm_peak_alloc_GB", "resident_activations", "cpu_baseline") if k in d}
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


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


def selftest_workload(a):
    """`--gpus N --selftest_workload [--forget class] [--scaling strong]`: host-only check (gloo on a CPU box) that the
    N ranks build the SAME marked set and that their shards partition every global batch of the epoch pattern — the
    data side of the benchmark under the launcher, no device work."""
    import contextlib
    from unlearn_saliency_amd import dist as sdist
    rank, _, world = sdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher created WORLD_SIZE={world}")
    with contextlib.redirect_stdout(sys.stderr):
        model, fl, rl = build_workload(torch.device("cpu"), rank, world, a.batch_size, a.forget, a.class_to_replace,
                                       a.scaling, device_resident=False)
    fl.dataset.transform = rl.dataset.transform = "test"  # no augmentation draws: the check is about the sharding
    sizes, label_sum, classes = [], 0, set()
    for i, (x, y) in enumerate(fl):
        lo, hi, b = fl.last_shard
        assert x.size(0) == hi - lo
        sizes.append([hi - lo, b])
        label_sum += int(y.sum())
        classes |= set(y.tolist())
    t = torch.tensor(sizes, dtype=torch.int64)
    got = [torch.zeros_like(t) for _ in range(world)]
    if sdist.is_dist():
        torch.distributed.all_gather(got, t)
    else:
        got = [t]
    cls = torch.zeros(10, dtype=torch.int64)
    cls[list(classes)] = 1
    sdist.all_reduce_sum_(cls)
    if rank == 0:
        per_rank = torch.stack([g[:, 0] for g in got])            # [world][batches]
        glob = got[0][:, 1]
        print(json.dumps({"selftest": "workload", "n_gpus": world, "forget": a.forget, "scaling": a.scaling,
                          "forget_samples": len(fl.dataset), "retain_samples": len(rl.dataset),
                          "global_batch": fl.batch_size, "forget_batches": len(sizes),
                          "shards_partition_every_batch": bool((per_rank.sum(0) == glob).all()),
                          "shard_imbalance_max": int((per_rank.max(0).values - per_rank.min(0).values).max()),
                          "tail_batch": [int(v) for v in per_rank[:, -1]],
                          "forget_classes": [i for i in range(10) if int(cls[i]) > 0]}), flush=True)
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
    if a.steps is None:  # per-workload defaults that finish within minutes
        a.steps = {"resnet18": 177, "ddpm": 20, "sd": 3}[a.workload]
    if a.warmup is None:
        a.warmup = {"resnet18": 10, "ddpm": 3, "sd": 1}[a.workload]
    if a.cpu_survey:
        a.cpu_steps, a.cpu_repeats = 20, 4
    if a.force_collectives:
        os.environ["SALUN_FORCE_COLLECTIVES"] = "1"
    from unlearn_saliency_amd import dist as sdist
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without torchrun: become the launcher of N ranks (one per GPU, RCCL)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            ensure_built() if not (a.selftest_launcher or a.selftest_workload) else None
        raise SystemExit(sdist.launch_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus,
                                            require_devices=not (a.selftest_launcher or a.selftest_workload)))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} disagrees with the launcher's WORLD_SIZE={os.environ['WORLD_SIZE']}")
    if a.selftest_launcher:
        return selftest_launcher(a)
    if a.selftest_workload:
        return selftest_workload(a)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (no CPU fallback for the measured path)")
    if torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", "1")):
        raise SystemExit(f"bench.py --gpus {a.gpus} needs {a.gpus} devices, found {torch.cuda.device_count()}")
    if int(os.environ.get("RANK", "0")) == 0:
        ensure_built()
    if a.workload in ("ddpm", "sd"):
        # the same launcher / rank environment serves the diffusion workloads: tools/bench_{ddpm,sd}.py read
        # RANK / WORLD_SIZE through dist.init_from_env and shard their batches (BASELINE configs[3] / [4])
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        argv = ["--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup),
                "--scaling", a.scaling] + (["--no_cpu_baseline"] if a.no_cpu_baseline else []) \
            + (["--digest"] if a.digest else [])
        if a.workload == "ddpm":
            import bench_ddpm
            return bench_ddpm.main(argv + (["--mask_batches", str(a.ddpm_mask_batches)] if a.no_mask_gen else []))
        import bench_sd
        return bench_sd.main(argv + ["--bf16", "--resident"])
    rank, local_rank, world = sdist.init_from_env()
    from unlearn_saliency_amd import _lib
    _lib.lib()  # fail loudly if the HIP extension is missing
    rccl_ranks = sdist.counted_ranks()  # an actual all-reduce over the ranks RCCL sees
    assert rccl_ranks == world == a.gpus, (rccl_ranks, world, a.gpus)
    device = torch.device("cuda", torch.cuda.current_device())
    torch.backends.cudnn.benchmark = True
    if a.scaling == "strong" and a.batch_size % world:
        raise SystemExit(f"--scaling strong: global batch {a.batch_size} does not divide over {world} ranks")

    from unlearn_saliency_amd.Classification.unlearn.impl import FusedMaskedSGD
    from unlearn_saliency_amd.flat import arena_of
    from unlearn_saliency_amd import ops

    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # stdout carries exactly one JSON line
        model, forget_loader, retain_loader = build_workload(device, rank, world, a.batch_size, a.forget,
                                                             a.class_to_replace, a.scaling)
    gbs = forget_loader.batch_size  # global batch of one step
    torch.backends.cudnn.deterministic = bool(a.deterministic)  # setup_seed() above turned it on
    torch.backends.cudnn.benchmark = True
    if a.channels_last:
        model = model.to(memory_format=torch.channels_last)
    n_salun_convs = 0
    if a.salun_conv and not a.channels_last:
        from unlearn_saliency_amd.conv import use_salun_convs
        n_salun_convs = use_salun_convs(model)
    n_fused_bn = 0
    if a.fused_bn and not a.channels_last and not (a.sync_bn and world > 1):
        from unlearn_saliency_amd.norm import use_fused_bn
        n_fused_bn = use_fused_bn(model)
    if a.sync_bn and world > 1:  # global-batch statistics (reference-equivalent numerics under strong scaling)
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
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
    seen = [0]  # samples THIS rank pushed through a step (ragged tail batches count for what they hold)

    def one_step(ev=None):
        x, y, w = next(stream)
        seen[0] += x.size(0)
        if a.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if x.size(0) == 0:  # strong scaling, tail batch smaller than the world: zero gradient from this rank
            opt.zero_grad()
        else:
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
    tails = set()
    for ld in (forget_loader, retain_loader):
        b = len(ld.dataset) % gbs
        if b:
            lo, hi = ld._slice(b)
            tails.add(hi - lo)
    for tail in sorted(tails - {0}):
        xs = torch.rand(tail, 3, 32, 32, device=device)
        ys = torch.randint(0, 10, (tail,), device=device)
        arena.zero_grad()
        with sync_free(opt):
            criterion(model(xs), ys).backward()
        arena.zero_grad()  # discard: no parameter update from the shape warm-up
    from unlearn_saliency_amd import hostperf
    hostperf.freeze_gc()  # as the unlearning epoch driver does before its first epoch (unlearn/impl.py)
    for _ in range(a.warmup):
        one_step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    seen[0] = 0
    # no cyclic-GC pause inside the timed region (one 58 ms stall in 177 steps was seen on a box: 110.3 instead of
    # 114.4 steps/s); nothing is skipped — the collector runs before and after
    import gc
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    sdist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(events[i])
    torch.cuda.synchronize()
    sdist.barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    samples = float(seen[0])
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([samples], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        samples = float(t.item())

    # calibration, OUTSIDE the timed region (one rank, no collectives): what a HIP-event pair reads with nothing
    # between the two records, and the same update kernel launched back to back on the same vectors — the account of
    # how much of the in-step figure is the kernel and how much the two markers around it
    ev_overhead_us = iso_us = None
    if world == 1 and not sdist.collectives_on():
        cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for s_, e_ in cal[:20]:
            s_.record()
            e_.record()
        for s_, e_ in cal[20:]:
            s_.record()
            opt.step()
            e_.record()
        torch.cuda.synchronize()
        ev_overhead_us = 1e3 * sorted(s_.elapsed_time(e_) for s_, e_ in cal[:20])[10]
        iso_us = 1e3 * sorted(s_.elapsed_time(e_) for s_, e_ in cal[20:])[10]

    # the optimizer tail on the launch stream: at N=1 exactly one kernel (salun_masked_sgd_step)
    tail_ms = sorted(s.elapsed_time(e) for s, e in events)
    tail_mean_s = 1e-3 * sum(tail_ms) / len(tail_ms)
    # per-step device time from consecutive event timestamps (shows clock/thermal drift over the run)
    step_ms = [events[i][0].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]

    # forward+backward device time per step: from the end of the previous step's update to the start of this one's
    fb_ms = [events[i][1].elapsed_time(events[i + 1][0]) for i in range(len(events) - 1)]
    fb_mean_s = 1e-3 * sum(fb_ms) / max(len(fb_ms), 1)
    pmc_traffic, pmc_src = None, None
    try:  # HBM bytes per launch: a CONSTANT read from the committed PMC passes, not measured in this run
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pth = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
            if os.path.exists(pth):
                with open(pth) as f:
                    kern = json.load(f)["kernels"]
                pmc_traffic = (kern.get(f"k_masked_sgd_vec@{rnd}_n18") or kern["k_masked_sgd_vec@n18"])["traffic_bytes"]
                pmc_src = (f"constant from profiles/{rnd}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                           f"separate passes over tools/kbench.py, tools/pmc.sh) — PMC counters cannot be collected "
                           f"inside this run, so this field does not move with it")
                break
    except Exception:
        pass

    from unlearn_saliency_amd import conv as sconv
    library_conv_calls = dict(sconv.LIBRARY_CONV_CALLS, total=sconv.library_conv_calls())
    if rank == 0:
        # weak: every rank carries a full --batch_size batch per step, so the job does steps x ranks reference-sized
        # steps; strong: one step IS one reference step (global batch = --batch_size) whatever the rank count
        steps_per_s = a.steps * (world if a.scaling == "weak" else 1) / dt
        alg_bytes = SGD_BYTES_PER_ELEM * N18
        flop_per_step_rank = FWD_BWD_GFLOP_PER_IMG * samples / world / a.steps  # GFLOP, actual samples per rank
        what = ("10% random-data forget (4,500 of 45,000)" if a.forget == "random" else
                f"class-wise forget (all 4,500 samples of class {a.class_to_replace})")
        out = {
            "metric": ("unlearn_steps_per_sec (ResNet-18/CIFAR-10 10%-forget, RL + SalUn mask, batch 256/GPU)"
                       if a.forget == "random" and a.scaling == "weak" else
                       f"unlearn_steps_per_sec (ResNet-18/CIFAR-10 {a.forget} forget, RL + SalUn mask, "
                       f"{a.scaling} scaling)"),
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "backend": (torch.distributed.get_backend() if sdist.is_dist() else "single-process"),
            "collectives": bool(sdist.collectives_on()),
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ResNet-18 (11,173,962 params) / CIFAR-10-shaped synthetic set, {what}, "
                                   "RL unlearning step with SalUn mask ratio 0.5, "
                                   "SGD lr 0.013 momentum 0.9 wd 5e-4, RandomCrop+flip on device",
                       "forget": a.forget, "per_gpu_batch": gbs // world, "global_batch": gbs,
                       "parallelism": f"dp{world}", "params": N18, "cudnn_deterministic": bool(a.deterministic),
                       "batch_norm": "sync" if (a.sync_bn and world > 1) else "per-replica",
                       "channels_last": bool(a.channels_last), "salun_mfma_convs": n_salun_convs,
                       "fused_bn_layers": n_fused_bn, "library_conv_calls": library_conv_calls},
            "samples_per_sec": samples / dt, "samples_in_window": int(samples),
            "mask_gen_sec": None if mask_gen is None else mask_gen["total_sec"],
            "mask_gen": mask_gen,
            "step_ms_trend": {"first5": [round(v, 2) for v in step_ms[:5]], "last5": [round(v, 2) for v in step_ms[-5:]],
                              "min": round(min(step_ms), 2) if step_ms else None,
                              "max": round(max(step_ms), 2) if step_ms else None},
            "roofline": {"kernel": "salun_masked_sgd_step" + ("" if not sdist.collectives_on() else
                                                              " (+ flat-gradient all-reduce)"),
                         "bound": "hbm", "achieved": alg_bytes / tail_mean_s / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_bytes / tail_mean_s / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic,
                         "traffic_source": pmc_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_us": tail_mean_s * 1e6,
                         "median_launch_us": 1e3 * tail_ms[len(tail_ms) // 2],
                         "event_pair_overhead_us": ev_overhead_us, "back_to_back_launch_us": iso_us,
                         "frac_net_of_event_overhead": (None if ev_overhead_us is None else
                                                        alg_bytes / max(tail_mean_s - 1e-6 * ev_overhead_us, 1e-9) / 1e9
                                                        / HBM_PEAK_GBS),
                         "timing": "HIP events on the launch stream around the launch, inside the timed steps (the "
                                   "first event is recorded after backward's side-stream join, so the pair holds the "
                                   "kernel and the two markers); `event_pair_overhead_us` = an empty pair, "
                                   "`back_to_back_launch_us` = the same launch repeated with warm caches, both taken "
                                   "after the timed region"},
            "fwd_bwd": {"bound": "mfma", "gflop_per_step": flop_per_step_rank,
                        "achieved": flop_per_step_rank / (fb_mean_s * 1e3) if fb_mean_s else None,
                        "peak": FP32_MATRIX_PEAK_TF, "unit": "TFLOP/s",
                        "frac": (flop_per_step_rank / (fb_mean_s * 1e3) / FP32_MATRIX_PEAK_TF) if fb_mean_s else None,
                        "mean_fwd_bwd_ms": fb_mean_s * 1e3,
                        "timing": "HIP events: end of step i's update -> start of step i+1's update; FLOPs from the "
                                  "samples actually processed (tail batches count for what they hold)",
                        "note": ("convolutions: hand-written fp32 MFMA implicit-GEMM kernels (salun_conv2d_*); "
                                 + ("BN(+add)+ReLU: fused kernels (salun_bn_*); pool/fc/CE: PyTorch-ROCm" if n_fused_bn
                                   else "BN/ReLU/pool/fc: PyTorch-ROCm")) if n_salun_convs else
                                "convolutions/GEMMs run in MIOpen/rocBLAS fp32 through PyTorch-ROCm"},
        }
        if world == 1 and not a.no_cpu_baseline:
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_baseline(a.batch_size, a.cpu_steps, a.cpu_repeats)
        if (world == 1 and not a.no_ddpm and a.forget == "random" and a.scaling == "weak"
                and not sdist.collectives_on()):
            # north_star's second target beside the headline: CFG-DDPM class-forget (BASELINE configs[3]) — the same
            # step `--workload ddpm` times, a shorter window, one reference step on the CPU (~80 s on 128 threads)
            del model, opt, arena, forget_loader, retain_loader, stream
            torch.cuda.empty_cache()
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_ddpm
            argv = ["--gpus", "1", "--steps", str(a.ddpm_steps), "--warmup", "3", "--mask_batches",
                    str(a.ddpm_mask_batches), "--cpu_steps", "1"] + (["--no_cpu_baseline"] if a.no_cpu_baseline else [])
            try:
                with contextlib.redirect_stdout(sys.stderr):
                    d = bench_ddpm.run(argv)
                out["ddpm"] = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype",
                                                 "config", "samples_per_sec", "mask_gen", "roofline", "fwd_bwd",
                                                 "cpu_baseline") if k in d}
            except Exception as exc:  # the headline line must not be lost to the second workload
                out["ddpm"] = {"error": f"{type(exc).__name__}: {exc}"}
            if not a.no_sd:
                # BASELINE configs[4]: the SD-v1 U-Net nsfw_removal step in its bf16 configuration — `--workload sd` in a
                # process of its own: the step is host-bound (the host needs ~190 of its ~190 ms), and inside THIS
                # process, behind two other workloads' allocator and interpreter state, it measured 222 ms against 180
                out["sd"] = sd_block(a)
            if not a.no_dp:
                # the data-parallel path at world size 1 (the N > 1 step is not the N = 1 step — other
                # stream schedule, gradient slices through RCCL): same workloads, short windows, one subprocess each
                out["dp_ws1"] = {"resnet18": dp_ws1_checked(["--steps", "60", "--warmup", "10", "--no_mask_gen"],
                                                         out["ms_per_step"]),
                                 "ddpm": dp_ws1_checked(["--workload", "ddpm", "--steps", "8", "--warmup", "3",
                                                      "--no_mask_gen", "--ddpm_mask_batches", "2"],
                                                     out.get("ddpm", {}).get("ms_per_step")),
                                 "note": "`bench.py --gpus 1 --force_collectives ...`: process group over RCCL with one "
                                         "rank, flat-gradient slices all-reduced (AVG) from autograd hooks behind "
                                         "backward, the fused update waits for them; `plain_ms_per_step` is this "
                                         "line's single-process figure for the same workload"}
                if not a.no_sd:
                    out["dp_ws1"]["sd"] = dp_ws1_checked(["--workload", "sd", "--steps", "3", "--warmup", "1"],
                                                      out.get("sd", {}).get("ms_per_step"))
        print(json.dumps(out), flush=True)
    sdist.barrier()
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


class sync_free:
    """Run a backward whose gradients are thrown away: under data parallel the gradient-bucket hooks would start
    all-reduces that nothing waits for; finish them right away so the next real step starts from a clean reducer."""

    def __init__(self, opt):
        self.opt = opt

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        red = getattr(self.opt, "_reducer", None)
        if red is not None:
            red.finish()
        return False


if __name__ == "__main__":
    main()
