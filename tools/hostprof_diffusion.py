"""Where the HOST's time goes in one diffusion unlearning step (cProfile over 3 steps after warm-up; the device runs
behind).  `python tools/hostprof_diffusion.py ddpm|sd`"""
import cProfile, io, os, pstats, sys, time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ddpm():
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.DDPM.functions import cycle, get_optimizer, load_config
    from unlearn_saliency_amd.DDPM.runners.diffusion import Diffusion
    from unlearn_saliency_amd.flat import arena_of
    cfg = load_config(os.path.join(ROOT, "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    args = SimpleNamespace(ckpt_folder=None, label_to_forget=0, cond_scale=2.0, mask_path=None, method="rl",
                           alpha=1e-3, synthetic=True, library_conv=False)
    runner = Diffusion(args, cfg)
    rl, fl = runner._loaders()
    model = runner._load_model()
    arena = arena_of(model)
    opt = get_optimizer(cfg, arena=arena)
    sal = ops.fill_normal(arena.n, 5, 0.0, 1e-3) * (1.0 + ops.fill_uniform(arena.n, 6, 0.0, 0.5))
    opt.set_mask(ops.mask_topk(sal, [arena.n // 2])[0])
    model.train()
    ri, fi = cycle(rl), cycle(fl)
    return lambda: runner.unlearn_step(model, opt, next(ri), next(fi))


def sd():
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.optim import FusedMaskedAdam
    from unlearn_saliency_amd.SD import train_scripts as TS
    from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
    dev = torch.device("cuda")
    model = LatentDiffusionLite(bf16=True).to(dev)
    arena = TS._unet_arena(model)
    model.use_mfma_convs()
    model.fill_zero_initialised()
    opt = FusedMaskedAdam(arena, lr=1e-5)
    sal = ops.fill_normal(arena.n, 5, 0.0, 1e-3) * (1.0 + ops.fill_uniform(arena.n, 6, 0.0, 0.5))
    opt.set_mask(ops.mask_topk(sal, [arena.n // 2])[0])
    del sal
    model.train()
    B = 8
    mk = lambda *s: torch.randn(*s, device=dev)
    z_f, c_f, c_p, z_r, c_r = mk(B, 4, 64, 64), mk(B, 77, 768), mk(B, 77, 768), mk(B, 4, 64, 64), mk(B, 77, 768)

    def step():
        opt.zero_grad()
        remain_loss = model.shared_step({"z": z_r, "c": c_r})[0]
        t = torch.randint(0, model.num_timesteps, (B,), device=dev).long()
        z_noisy = model.q_sample(x_start=z_f, t=t, noise=torch.randn_like(z_f))
        fo, po = TS.forget_and_target(model, z_noisy, t, c_f, c_p)
        loss = ops.mse_loss(po, fo) + 0.1 * remain_loss
        loss.backward()
        opt.step()
    return step


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "ddpm"
    step = {"ddpm": ddpm, "sd": sd}[which]()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    # the engine runs device backward nodes on a thread of its own, which cProfile does not see: keep them on this thread
    with torch.autograd.set_multithreading_enabled(False):
        pr.enable()
        for _ in range(n):
            step()
        pr.disable()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{which}: host {1e3 * t_host / n:.1f} ms/step under cProfile (device done after {1e3 * t_all / n:.1f} ms/step)")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(70)
    txt = s.getvalue()
    print("\n".join(l[:170] for l in txt.splitlines()[:100]))


if __name__ == "__main__":
    main()
