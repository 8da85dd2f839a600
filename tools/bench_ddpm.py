"""DDPM class-forget benchmark (BASELINE.json configs[3]): CFG-DDPM U-Net (38,632,323 params), CIFAR-shaped
synthetic data, class 0 forget, batch 128 per GPU — times Phase A (40 forget batches, CFG loss, per-batch clip) and the
masked unlearning step (method rl: remain pass + forget pass + pseudo pass, clip -> mask -> fused Adam).

    python bench.py --workload ddpm --gpus N --steps K --warmup W        (the driver-reachable form; launches N ranks)
    python tools/bench_ddpm.py [--steps K] [--warmup W] [--mask_batches M] [--library_conv]      (one GPU)

Rank-aware: reads RANK / WORLD_SIZE through dist.init_from_env; every global batch (128 x N under weak scaling, 128
under strong) is sharded contiguously over the ranks by the loaders, noise / timesteps are drawn for the global batch
and sliced (runners.diffusion.ShardDraws), Phase A all-reduces the flat gradient per batch (the clip needs the
global norm), Phase B all-reduces it per step in buckets overlapped with backward.  Prints one JSON line on rank 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib
import torch
from types import SimpleNamespace

ND = 38_632_323


class Take:
    """The first `k` batches of a loader, still exposing `.last_shard` (what ShardDraws reads under data parallel)."""

    def __init__(self, loader, k):
        self.loader, self.k = loader, k

    @property
    def last_shard(self):
        return self.loader.last_shard

    def __iter__(self):
        it = iter(cycle_(self.loader))
        for _ in range(self.k):
            yield next(it)


def cycle_(dl):
    while True:
        for b in dl:
            yield b


def cpu_baseline(cfg, steps=3):
    """The reference's saliency_unlearn loop body (DDPM/runners/diffusion.py:520-593, method rl) as plain PyTorch-CPU
    ops on this host's cores: remain eps-MSE + forget/pseudo MSE, backward, clip_grad_norm_, per-tensor mask multiply,
    torch.optim.Adam — `steps` steps at batch 128 after one warm-up step at batch 16 (bounded sample)."""
    from oracle import torch_ref
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    from unlearn_saliency_amd.DDPM.runners.diffusion import get_beta_schedule
    torch.manual_seed(0)
    model = Conditional_Model(cfg)
    model.train()
    d = cfg.diffusion
    betas = torch.from_numpy(get_beta_schedule(d.beta_schedule, beta_start=d.beta_start, beta_end=d.beta_end,
                                               num_diffusion_timesteps=d.num_diffusion_timesteps)).float()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.optim.lr)
    mask = {n: (torch.rand_like(p) < 0.5).to(torch.int64) for n, p in model.named_parameters()}

    def step(bs):
        x, c = torch.rand(bs, 3, 32, 32) * 2 - 1, torch.randint(1, 10, (bs,))
        xf, cf = torch.rand(bs, 3, 32, 32) * 2 - 1, torch.zeros(bs, dtype=torch.int64)
        T = betas.numel()
        t = torch.randint(0, T, (bs,))
        e = torch.randn_like(x)
        out = model(torch_ref.qsample_cpu(x, e, betas, t), t.float(), c, cond_drop_prob=0.1, mode="train")
        remain = torch_ref.eps_mse_cpu(e, out)
        e = torch.randn_like(xf)
        xt = torch_ref.qsample_cpu(xf, e, betas, t)
        o = model(xt, t.float(), cf, mode="train")
        with torch.no_grad():
            pseudo = model(xt, t.float(), (cf + 1) % 10, mode="train")
        loss = torch.nn.MSELoss()(pseudo, o) + 1e-3 * remain
        opt.zero_grad()
        loss.backward()
        torch_ref.masked_adam_step_cpu(model, opt, mask, 1.0)

    step(16)
    t0 = time.perf_counter()
    for _ in range(steps):
        step(128)
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} rl step(s) at batch 128 (CFG-DDPM 38.6 M params fp32: remain pass + forget pass + pseudo "
                      f"pass, clip, 334x mask-mul, torch.optim.Adam) after one warm-up step at batch 16",
            "ms_per_step": 1e3 * dt / steps, "host_cpu_count": os.cpu_count()}


def main(argv=None):
    out = run(argv)
    if out is not None:
        print(json.dumps(out), flush=True)
    from unlearn_saliency_amd import dist as sdist
    sdist.barrier()
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


def pmc_tail_traffic():
    """HBM bytes of the optimizer tail (salun_grad_sqnorm + salun_masked_adam_step at N_D) from the committed PMC passes:
    a CONSTANT (counters cannot be collected inside a timed run), labelled as such."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        pth = os.path.join(root, "profiles", f"{rnd}_pmc_traffic.json")
        if not os.path.exists(pth):
            continue
        try:
            with open(pth) as f:
                kern = json.load(f)["kernels"]
            t = kern[f"k_sqnorm_partial@{rnd}_nd"]["traffic_bytes"] + kern[f"k_masked_adam@{rnd}_nd"]["traffic_bytes"]
            return t, (f"constant from profiles/{rnd}_pmc_traffic.json (k_sqnorm_partial + k_masked_adam at N_D; rocprofv3 "
                       f"--pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc.sh) — not measured in this run")
        except Exception:
            continue
    return None, None


def run(argv=None):
    """One DDPM measurement; returns the result dict on rank 0 (None elsewhere).  bench.py's default line embeds it."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mask_batches", type=int, default=40)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--library_conv", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--digest", action="store_true",
                    help="add the SHA-256 of the parameters and the last step's loss to the line (equality tests)")
    ap.add_argument("--cpu_steps", type=int, default=3,
                    help="rl steps at batch 128 timed on the host cores: SURVEY.md D3 (iii) asks 3 (one step takes ~77 s "
                         "on the GPU box's 128 threads, i.e. ~4 min for the baseline leg; --no_cpu_baseline skips it)")
    a = ap.parse_args(argv)
    from unlearn_saliency_amd import dist as sdist
    from unlearn_saliency_amd.DDPM.functions import load_config, get_optimizer, cycle
    from unlearn_saliency_amd.DDPM.runners.diffusion import Diffusion
    from unlearn_saliency_amd.flat import arena_of
    from unlearn_saliency_amd import ops
    rank, local_rank, world = sdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} disagrees with WORLD_SIZE={world} (launch through bench.py --workload ddpm)")
    rccl_ranks = sdist.counted_ranks()
    assert rccl_ranks == world, (rccl_ranks, world)
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = load_config(os.path.join(here, "..", "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    per_gpu = cfg.training.batch_size  # 128 (configs/cifar10_saliency_unlearn.yml)
    if a.scaling == "weak":
        cfg.training.batch_size = per_gpu * world  # the loaders shard each GLOBAL batch over the ranks
    elif per_gpu % world:
        raise SystemExit(f"--scaling strong: batch {per_gpu} does not divide over {world} ranks")
    args = SimpleNamespace(ckpt_folder=None, label_to_forget=0, cond_scale=2.0, mask_path=None, method="rl",
                           alpha=1e-3, synthetic=True, library_conv=a.library_conv)
    torch.manual_seed(1234)  # identical on every rank: global-batch draws are sliced per rank (ShardDraws)
    np_seed = 1234
    import numpy as np
    np.random.seed(np_seed)  # the flip frozen at materialisation time must agree across ranks
    torch.backends.cudnn.benchmark = True
    with contextlib.redirect_stdout(sys.stderr):
        runner = Diffusion(args, cfg)
        remain_loader, forget_loader = runner._loaders()
        model = runner._load_model()
    device = runner.device
    arena = arena_of(model)
    assert arena.n == ND
    # Phase A
    runner.accumulate_saliency(model, Take(forget_loader, 2), arena)  # warm-up
    torch.cuda.synchronize()
    sdist.barrier()
    t0 = time.perf_counter()
    acc = runner.accumulate_saliency(model, Take(forget_loader, a.mask_batches), arena)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mask = ops.mask_topk(acc, [int(ND * 0.5)], check=True)[0]
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # Phase B
    opt = get_optimizer(cfg, arena=arena)
    opt.set_mask(mask)
    model.train()
    ri, fi = cycle(remain_loader), cycle(forget_loader)
    from unlearn_saliency_amd import hostperf
    hostperf.freeze_gc()  # as Diffusion.saliency_unlearn does before its first step
    for _ in range(a.warmup):
        runner.unlearn_step(model, opt, next(ri), next(fi))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    real_step = opt.step
    samples = 0
    torch.cuda.synchronize()
    sdist.barrier()
    if os.environ.get("SALUN_SYNC_DEBUG"):  # diagnostics: warn on every host-synchronising call inside the timed steps
        torch.cuda.set_sync_debug_mode(1)
    import gc
    gc.collect()
    gc.disable()  # no collector pause inside the timed steps (re-enabled below)
    t3 = time.perf_counter()
    for i in range(a.steps):
        # events bracket the optimizer tail ([all-reduce join] + sq-norm + masked Adam) by patching step()
        def timed_step(real=real_step, e=ev[i]):
            e[0].record(); r = real(); e[1].record(); return r
        opt.step = timed_step
        rb, fb_ = next(ri), next(fi)
        samples += rb[0].size(0)
        last_loss = runner.unlearn_step(model, opt, rb, fb_)
    opt.step = real_step
    host_enqueue_s = time.perf_counter() - t3  # the host has issued every step; the device may still be running
    torch.cuda.synchronize()
    sdist.barrier()
    dt = time.perf_counter() - t3
    gc.enable()
    samples = float(samples)
    if world > 1:
        t = torch.tensor([dt, 0.0], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0].item())
        t = torch.tensor([samples], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
        samples = float(t.item())
    tail_s = 1e-3 * sum(s.elapsed_time(e) for s, e in ev) / a.steps
    alg = 33 * ND  # grad sq-norm (4 B) + masked Adam (29 B) per element, SURVEY.md §8 D2
    # per sample of the remain batch one step does: remain fwd+bwd, forget fwd+bwd, pseudo fwd
    flops_step_rank = (samples / world / a.steps) * (2 * 37.34 + 12.45) * 1e9
    from unlearn_saliency_amd import conv as sconv
    if rank == 0:
        out = {"metric": "ddpm_unlearn_steps_per_sec (CFG-DDPM/CIFAR-10 class-forget, rl, batch 128/GPU)",
               "value": a.steps * (world if a.scaling == "weak" else 1) / dt, "unit": "steps/s", "n_gpus": world,
               "rccl_ranks": rccl_ranks,
               "backend": (torch.distributed.get_backend() if sdist.is_dist() else "single-process"),
               "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "params": ND,
               "config": {"workload": "CFG-DDPM U-Net (38,632,323 params) / CIFAR-10-shaped synthetic set, class-0 forget, "
                                      "saliency_unlearn method rl, alpha 1e-3, batch 128 per GPU, Adam 1e-4, clip 1.0, "
                                      "SalUn mask ratio 0.5 (BASELINE.json configs[3])",
                          "per_gpu_batch": cfg.training.batch_size // world, "global_batch": cfg.training.batch_size,
                          "parallelism": f"dp{world}",
                          "library_conv_calls": dict(sconv.LIBRARY_CONV_CALLS, total=sconv.library_conv_calls())},
               "samples_per_sec": samples / dt,
               "host_enqueue_ms_per_step": 1e3 * host_enqueue_s / a.steps,
               "mask_gen": {"batches": a.mask_batches, "saliency_sec": t1 - t0, "topk_sec": t2 - t1},
               "roofline": {"kernel": "salun_grad_sqnorm + salun_masked_adam_step"
                                      + ("" if not sdist.collectives_on() else " (+ gradient-bucket join)"),
                            "bound": "hbm", "achieved": alg / tail_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                            "frac": alg / tail_s / 1e9 / 8000.0, "mean_tail_us": tail_s * 1e6, "algorithmic_bytes": alg,
                            "traffic": pmc_tail_traffic()[0], "traffic_source": pmc_tail_traffic()[1]},
               "fwd_bwd": {"bound": "mfma", "tflop_per_step": flops_step_rank / 1e12,
                           "achieved_whole_step": flops_step_rank / (dt / a.steps) / 1e12, "peak": 157.3,
                           "frac_whole_step": flops_step_rank / (dt / a.steps) / 1e12 / 157.3, "unit": "TFLOP/s"},
               "mfma_convs": not a.library_conv}
        if a.digest:
            import hashlib
            out["params_sha256"] = hashlib.sha256(arena.params.cpu().numpy().tobytes()).hexdigest()
            out["last_loss"] = float(last_loss.detach())
            out["collectives"] = bool(sdist.collectives_on())
        if world == 1 and not a.no_cpu_baseline:
            with contextlib.redirect_stdout(sys.stderr):
                out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_steps)
        return out
    return None


if __name__ == "__main__":
    main()
