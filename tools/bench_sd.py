"""SD-v1 U-Net SalUn benchmark (BASELINE.json configs[4]: SD nudity; synthetic latents / contexts, randomly
initialised 859,520,964-parameter U-Net of the v1-inference shape, batch 8 per GPU at 64x64 latents).

    python bench.py --workload sd --gpus N --steps K --warmup W     (driver-reachable; launches N ranks, bf16)
    python tools/bench_sd.py [--steps K] [--warmup W] [--bf16] [--library_conv]             (one GPU)

Rank-aware (reads RANK / WORLD_SIZE through dist.init_from_env): every rank draws its own batch of 8 (the reference
script is single-GPU; under data parallel the global batch is 8 x N, weak scaling), Phase A sums the flat accumulator
over ranks once, Phase B averages the flat gradient per step in buckets overlapped with backward.

  Phase A  generate_nsfw_mask body: per batch 2 U-Net forwards + backward, flat accumulate, then the global
           top-k (ratio 0.5) over N_S = 859.5 M saliencies
  Phase B  nsfw_removal loop body: remain pass (fwd+bwd) + forget pass (fwd+bwd) + pseudo pass (fwd, no grad),
           masked Adam on the flat arena

Prints one JSON line.  python tools/bench_sd.py [--steps K] [--warmup W] [--bf16] [--library_conv]
"""
import argparse, contextlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

NS = 859_520_964


def _pmc_adam_traffic():
    """HBM bytes of one salun_masked_adam_step launch at N_S from the committed PMC passes: a CONSTANT (counters cannot be
    collected inside a timed run), labelled as such."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rnd in ("r06", "r05", "r04", "r02"):
        pth = os.path.join(root, "profiles", f"{rnd}_pmc_traffic.json")
        try:
            with open(pth) as f:
                t = json.load(f)["kernels"][f"k_masked_adam@{rnd}_ns"]["traffic_bytes"]
            return t, (f"constant from profiles/{rnd}_pmc_traffic.json (k_masked_adam at N_S; rocprofv3 --pmc FETCH_SIZE / "
                       f"WRITE_SIZE in separate passes, tools/pmc.sh) — not measured in this run")
        except Exception:
            continue
    return None, None

FWD_TFLOP_PER_SAMPLE = 0.803  # SURVEY.md §8 D2 (FlopCounterMode on the reference module)


def cpu_baseline(batch=1, steps=1):
    """The reference's nsfw_removal loop body (SD/train-scripts/nsfw_removal.py:60-150; oracle/torch_ref.py sd_unlearn:
    remain pass + forget pass + pseudo pass, backward, 686x mask multiply, torch.optim.Adam) in plain PyTorch fp32 — the
    precision the reference script runs in — on this host's cores: `steps` step(s) at batch `batch` on the 859.5 M
    parameter U-Net after nothing (the one step carries the thread pool's start-up; the sample is bounded: a batch-8
    step is ~8x the arithmetic).  No activation checkpointing on the CPU (the reference's config enables it; that
    is a memory device, one extra forward per backward)."""
    from oracle import torch_ref
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    torch.manual_seed(0)
    cfg = dict(V1_UNET_CONFIG)
    cfg["use_checkpoint"] = False
    t0 = time.perf_counter()
    unet = UNetModel(**cfg)
    with torch.no_grad():
        for p_ in unet.parameters():   # zero_module layers: give every tensor a gradient path, as the device run does
            if float(p_.abs().sum()) == 0.0:
                p_.normal_(0.0, 0.02)
    ldm = torch_ref.PlainLDM(unet)
    mask = {n: (torch.rand_like(p_) < 0.5).to(torch.int64) for n, p_ in unet.named_parameters()}
    t_init = time.perf_counter() - t0
    mk = lambda *sh: torch.randn(*sh)
    fb = [(mk(batch, 4, 64, 64), mk(batch, 77, 768), mk(batch, 77, 768)) for _ in range(steps)]
    rb = [(mk(batch, 4, 64, 64), mk(batch, 77, 768)) for _ in range(steps)]
    t0 = time.perf_counter()
    torch_ref.sd_unlearn(ldm, fb, rb, 0.1, 1e-5, mask)
    dt = time.perf_counter() - t0
    sps = steps / dt
    return {"value": sps * batch / 8.0, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "value_at_sampled_batch": sps, "sampled_batch": batch,
            "sample": f"{steps} nsfw_removal step(s) at batch {batch} (SD-v1 U-Net 859.5 M params fp32, reference op "
                      f"sequence: 3 forwards, 2 backwards, 686x mask-mul, torch.optim.Adam; no warm-up step, no "
                      f"checkpointing); `value` = steps/s at the metric's batch 8 assuming time linear in the batch "
                      f"(x {batch}/8), `value_at_sampled_batch` is what was timed",
            "ms_per_step_at_sampled_batch": 1e3 * dt / steps, "model_build_sec": t_init,
            "host_cpu_count": os.cpu_count(), "torch_version": torch.__version__}


def _aten_origins(step):
    """Which library kernels still run inside a step, and who asks for them (the own kernels are launched through ctypes
    and do not appear as operators)."""
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0 or e.device_type != torch.autograd.DeviceType.CPU:
            continue
        ours = [fr for fr in (e.stack or []) if "unlearn_saliency_amd" in fr or "tools/" in fr]
        shapes = str(getattr(e, "input_shapes", "")) if e.name.startswith("aten::") else ""
        key = (e.name, " <- ".join(f.split("unlearn_saliency_amd/")[-1].strip() for f in ours[:3]) + " " + shapes)
        r = rows.setdefault(key, [0, 0.0, set()])
        r[0] += 1
        r[1] += dt
        r[2].add(str(getattr(e, "input_shapes", "")))
    tot = sum(r[1] for r in rows.values())
    print(f"[aten_origins] {sum(r[0] for r in rows.values())} operator calls with device time, {tot / 1e3:.2f} ms in one step", file=sys.stderr)
    for (name, where), r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:90]:
        print(f"[aten_origins] {r[1] / 1e3:8.3f} ms {r[0]:5d}x  {name:32s} {where}", file=sys.stderr)


def main(argv=None):
    out = run(argv)
    from unlearn_saliency_amd import dist as sdist
    if out is not None:
        print(json.dumps(out), flush=True)
    sdist.barrier()
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


def run(argv=None):
    """-> the result line as a dict on rank 0 (None on the other ranks).  The process group is left to the caller."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--scaling", default="weak", choices=["weak"], help="batch 8 per GPU cannot be split further")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_batch", type=int, default=1, help="batch of the one fp32 reference step timed on the host")
    ap.add_argument("--digest", action="store_true",
                    help="add the SHA-256 of the parameters and the last step's loss to the line (equality tests)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mask_batches", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--library_conv", action="store_true")
    ap.add_argument("--own_linear", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--resident", action="store_true",
                    help="after the timed steps (the reference's configuration: activation checkpointing on) time the same "
                         "steps once more with every activation kept in HBM (unet.set_activation_checkpointing(False): no "
                         "recompute, identical results) and report it as `resident_activations`")
    ap.add_argument("--aten_origins", action="store_true",
                    help="diagnostic: one extra step under torch.profiler; the library (ATen) device kernels of the step "
                         "grouped by operator and the calling frames of this package go to stderr")
    ap.add_argument("--library_linear", action="store_true",
                    help="A/B: leave the transformer blocks' Linear layers on the library GEMM (hipBLASLt under autocast) "
                         "instead of K16 (forward / input gradient) + K11 (weight gradient)")
    a = ap.parse_args(argv)
    from unlearn_saliency_amd import dist as sdist
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.SD import train_scripts as TS
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    rank, local_rank, world = sdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} disagrees with WORLD_SIZE={world} (launch through bench.py --workload sd)")
    rccl_ranks = sdist.counted_ranks()
    assert rccl_ranks == world, (rccl_ranks, world)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(0)  # identical initial weights on every rank (replicas stay bit-identical: no broadcast)
    torch.backends.cudnn.benchmark = True
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sys.stderr):
        model = LatentDiffusionLite(bf16=a.bf16).to(dev)
    arena = TS._unet_arena(model)
    assert arena.n == NS, arena.n
    n_filled = model.fill_zero_initialised()  # zero_module layers / biases -> N(0, 0.02): gradients flow everywhere
    n_salun = n_linear = 0
    from unlearn_saliency_amd import conv as sconv
    if not a.library_conv and not a.bf16:
        n_salun = sconv.use_salun_convs(model)
    elif not a.library_conv:
        from unlearn_saliency_amd.conv_bf16 import use_salun_convs_bf16, use_salun_linears_bf16
        n_salun = use_salun_convs_bf16(model)
        if not a.library_linear:  # the transformer blocks' Linear layers on K16 / K11 (conv_bf16.SalunLinearBF16)
            n_linear = use_salun_linears_bf16(model)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    B = a.batch
    torch.manual_seed(1000 + rank)  # from here on every rank draws its OWN latents / noise / timesteps
    mk = lambda *s: torch.randn(*s, device=dev)
    # ---- Phase A
    batches = [(mk(B, 4, 64, 64), mk(B, 77, 768), mk(B, 77, 768)) for _ in range(a.mask_batches)]
    TS._saliency_mask(model, batches[:1], 7.5, None)  # warm-up (kernel selection, workspaces)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc_mask = TS._saliency_mask(model, batches, 7.5, None)
    torch.cuda.synchronize()
    t_mask = time.perf_counter() - t0
    # high-entropy magnitudes (a plain Irwin-Hall normal has ~4e5 distinct values: every threshold would split a
    # run of thousands of equal keys, which is the full-scan fallback, not the typical saliency vector)
    acc = ops.fill_normal(NS, 5, 0.0, 1e-3) * (1.0 + ops.fill_uniform(NS, 6, 0.0, 0.5))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    m = ops.mask_topk(acc, [int(NS * 0.5)], check=True)[0]
    ev[1].record()
    torch.cuda.synchronize()
    topk_ms = ev[0].elapsed_time(ev[1])
    assert ops.mask_popcount(m) == int(NS * 0.5)
    del acc
    # ---- Phase B
    mask_dict_path = None  # the resident u8 mask is installed directly (the 6.9 GB int64 file round trip is skipped)
    fdl = lambda k: [(mk(B, 4, 64, 64), mk(B, 77, 768), mk(B, 77, 768)) for _ in range(k)]
    rdl = lambda k: [(mk(B, 4, 64, 64), mk(B, 77, 768)) for _ in range(k)]

    def one_step(z_f, c_f, c_p, z_r, c_r):
        # same body as TS._unlearn with the saliency mask already packed
        opt = steps_.opt
        opt.zero_grad()
        remain_loss = model.shared_step({"z": z_r, "c": c_r})[0]
        t = torch.randint(0, model.num_timesteps, (B,), device=dev).long()
        noise = torch.randn_like(z_f)
        z_noisy = model.q_sample(x_start=z_f, t=t, noise=noise)
        forget_out, pseudo_out = TS.forget_and_target(model, z_noisy, t, c_f, c_p)
        loss = ops.mse_loss(pseudo_out, forget_out) + 0.1 * remain_loss
        loss.backward()
        steps_.last_loss = loss.detach()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step()
        e1.record()
        steps_.tail.append((e0, e1))
        return (loss.detach(),)

    def steps_(k):
        model.train()
        steps_.tail = []
        for (z_f, c_f, c_p), (z_r, c_r) in zip(fdl(k), rdl(k)):
            one_step(z_f, c_f, c_p, z_r, c_r)
        return steps_.tail

    from unlearn_saliency_amd.optim import FusedMaskedAdam
    steps_.opt = FusedMaskedAdam(arena, lr=1e-5)
    steps_.opt.set_mask(m)
    from unlearn_saliency_amd import hostperf
    hostperf.freeze_gc()  # what the training loops of train_scripts.py do before their first step
    steps_(a.warmup)
    torch.cuda.synchronize()
    if a.aten_origins:
        _aten_origins(lambda: steps_(1))
    sdist.barrier()
    if os.environ.get("SALUN_SYNC_DEBUG"):  # diagnostics: warn on every host-synchronising call inside the timed steps
        torch.cuda.set_sync_debug_mode(1)
    import gc
    gc.collect()
    gc.disable()  # no collector pause inside the timed steps (re-enabled below)
    t0 = time.perf_counter()
    tail = steps_(a.steps)
    host_enqueue_s = time.perf_counter() - t0  # the host has issued every step; the device may still be running
    torch.cuda.synchronize()
    sdist.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    tail_ms = sum(e0.elapsed_time(e1) for e0, e1 in tail) / len(tail)
    # step = 3 forwards + 2 backwards (2x forward each) + 2 recomputed forwards (activation checkpointing)
    tflop = B * FWD_TFLOP_PER_SAMPLE * (3 + 4 + 2)
    out = {
        "metric": "sd_unlearn_steps_per_sec (SD-v1 U-Net nsfw_removal body, batch 8, 64x64 latents)",
        "value": world / dt, "unit": "steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
        "backend": (torch.distributed.get_backend() if sdist.is_dist() else "single-process"),
        "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": 1e3 * host_enqueue_s / a.steps, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "config": {"workload": "Stable Diffusion v1 LDM U-Net (859,520,964 params), nsfw_removal loop body "
                               "(SD/train-scripts/nsfw_removal.py:33-175): remain pass + forget pass + pseudo pass, "
                               "SalUn mask ratio 0.5, masked Adam 1e-5, 64x64 latents, batch 8 per GPU "
                               "(BASELINE.json configs[4])",
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}"},
        "dtype": "bf16 autocast (fp32 master weights / Adam)" if a.bf16 else "f32", "data": "synthetic",
        "params": NS, "zero_initialised_elements_filled": n_filled, "salun_mfma_convs": n_salun, "salun_mfma_linears": n_linear, "library_conv_calls": sconv.library_conv_calls(), "init_sec": t_init,
        "mask_gen": {"batches": a.mask_batches, "saliency_sec": t_mask, "topk_ms_at_NS": topk_ms,
                     "topk_GBps_algorithmic": 5.0 * NS / (topk_ms * 1e-3) / 1e9},
        "roofline": {"kernel": "salun_masked_adam_step @ N_S" + ("" if not sdist.collectives_on() else
                                                                  " (+ gradient-bucket join)"),
                     "traffic": _pmc_adam_traffic()[0], "traffic_source": _pmc_adam_traffic()[1],
                     "bound": "hbm", "algorithmic_bytes": 29 * NS,
                     "mean_launch_ms": tail_ms, "achieved": 29.0 * NS / (tail_ms * 1e-3) / 1e9, "peak": 8000.0,
                     "unit": "GB/s", "frac": 29.0 * NS / (tail_ms * 1e-3) / 1e9 / 8000.0},
        "fwd_bwd": {"tflop_per_step": tflop, "achieved_TFLOPs": tflop / dt,
                    "peak_TFLOPs": 2500.0 if a.bf16 else 157.3, "frac": tflop / dt / (2500.0 if a.bf16 else 157.3),
                    "note": "3 fwd + 2 bwd + 2 checkpoint recomputes of 0.803 TFLOP/sample; peak = dense bf16 / fp32 MFMA"},
        "kernels": {"conv": "K11 bf16 NHWC MFMA" if (a.bf16 and n_salun) else ("K8 fp32 MFMA" if n_salun else "library"),
                    "group_norm": "K12 bf16 NHWC" if a.bf16 else "fused fp32",
                    "attention": "K13 fused bf16" if a.bf16 else "library scaled_dot_product_attention",
                    "layer_norm_geglu": "K14 bf16 tokens" if a.bf16 else "library"},
        "hbm_peak_alloc_GB": torch.cuda.max_memory_allocated() / 1e9,
    }
    if a.resident:
        # the same loop with the activations resident: 3 forwards + 2 backwards, no recomputed forwards
        from unlearn_saliency_amd.SD.unet import set_activation_checkpointing
        n_blocks = set_activation_checkpointing(model.model.diffusion_model, False)
        torch.cuda.reset_peak_memory_stats()
        steps_(1)
        torch.cuda.synchronize()
        sdist.barrier()
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        steps_(a.steps)
        host_r = time.perf_counter() - t0
        torch.cuda.synchronize()
        sdist.barrier()
        dt_r = (time.perf_counter() - t0) / a.steps
        gc.enable()
        if world > 1:
            t = torch.tensor([dt_r], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_r = float(t.item())
        tflop_r = B * FWD_TFLOP_PER_SAMPLE * (3 + 4)
        out["resident_activations"] = {
            "value": world / dt_r, "unit": "steps/s", "ms_per_step": dt_r * 1e3,
            "host_enqueue_ms_per_step": 1e3 * host_r / a.steps, "steps": a.steps, "blocks_switched": n_blocks,
            "hbm_peak_alloc_GB": torch.cuda.max_memory_allocated() / 1e9,
            "fwd_bwd": {"tflop_per_step": tflop_r, "achieved_TFLOPs": tflop_r / dt_r,
                        "frac": tflop_r / dt_r / (2500.0 if a.bf16 else 157.3)},
            "note": "same steps, same results (tests/test_sd_gpu.py: bit-identical parameters), activation checkpointing of "
                    "the reference's v1-inference.yaml switched off: with 288 GB of HBM the ~50 GB of activations of a "
                    "batch-8 step stay resident instead of being recomputed inside backward (3 fwd + 2 bwd = 7 "
                    "forward-equivalents instead of 9).  `value` of this line stays the checkpointed configuration."}
        set_activation_checkpointing(model.model.diffusion_model, True)
    if a.digest:
        import hashlib
        out["params_sha256"] = hashlib.sha256(arena.params.cpu().numpy().tobytes()).hexdigest()
        out["last_loss"] = float(steps_.last_loss)
    out["collectives"] = bool(sdist.collectives_on())
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        del model, arena, m
        steps_.opt = None
        torch.cuda.empty_cache()
        with contextlib.redirect_stdout(sys.stderr):
            out["cpu_baseline"] = cpu_baseline(a.cpu_batch, 1)
    return out if rank == 0 else None


if __name__ == "__main__":
    main()
