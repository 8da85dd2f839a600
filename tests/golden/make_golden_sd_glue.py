"""Reference-run goldens for the Stable-Diffusion SalUn scripts.

    python tests/golden/make_golden_sd_glue.py          ->  tests/golden/sd_glue.npz

EXECUTES the reference's own functions, imported from /root/reference/SD/train-scripts (build container only):

    generate_mask.generate_mask / generate_nsfw_mask      (generate_mask.py:8-108, :111-211)
    nsfw_removal.nsfw_removal                              (nsfw_removal.py:33-175)
    random_label.certain_label                             (random_label.py:13-156)
    proximal_gradient.proximal_gradient                    (proximal_gradient.py:18-200)

What is stubbed is only what the image lacks and what the functions use as *plumbing*: `dataset` (its `setup_model`
/ `setup_*_data` return the objects built here), `diffusers.LMSDiscreteScheduler` (constructed, never used),
`convertModels.savemodelDiffusers`, `ldm.models.diffusion.ddim.DDIMSampler`, `torchvision.utils.make_grid`, `sleep`.
`LatentDiffusion` itself needs pytorch_lightning, so the model handed to the scripts is a duck-typed module around
the REFERENCE's `UNetModel` (openaimodel.py:428-847) that implements the five members the scripts touch
(`get_input`, `q_sample`, `apply_model`, `shared_step`, `num_timesteps` / `device` / `first_stage_key`) from
ddpm.py:424-430, :1093-1109, :1121, :1286-1319 with the reference's own `make_beta_schedule` /
`extract_into_tensor`.  "Images" are latents and "prompts" index a table of fixed context embeddings (the frozen VAE /
CLIP are outside the hot-path scope); `get_input` is deterministic, so `pseudo_input == forget_input` as in this
build's formulation.  For `proximal_gradient` the module also carries frozen stand-ins for the first stage and the
text encoder, because the reference ranks |theta - theta_0| over `model.parameters()` of the WHOLE model.

Every `torch.randint` / `torch.randn_like` result is recorded in call order (the tests replay them), `torch.abs_` is
observed to capture the accumulated gradients inside the mask functions, `torch.optim.Adam` to keep the optimizer.
Stored: data only — draws, accumulators, masks (packed bits), per-step losses, Adam moments and final weights
(strided samples + float64 checksums), no reference source.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
from fixtures import fill_params, sd_glue_batches, sd_glue_config, sd_glue_contexts  # noqa: E402
from make_golden import _stub  # noqa: E402
from unlearn_saliency_amd import rng  # noqa: E402

REF_SD = "/root/reference/SD"
sys.dont_write_bytecode = True
torch.set_num_threads(8)
STRIDE = 7  # strided samples of N-sized vectors kept in the fixture


# ------------------------------------------------------------------ the duck-typed LatentDiffusion
def build_ref_ldm(frozen: bool = False):
    from ldm.modules.diffusionmodules.openaimodel import UNetModel as RefUNet
    from ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule

    class DiffusionWrapper(nn.Module):  # ddpm.py: DiffusionWrapper with conditioning_key "crossattn"
        def __init__(self, unet):
            super().__init__()
            self.diffusion_model = unet

    class DuckLDM(nn.Module):
        def __init__(self):
            super().__init__()
            self.model = DiffusionWrapper(fill_params(RefUNet(**sd_glue_config()), 9100))
            if frozen:  # stand-ins for first_stage_model / cond_stage_model: parameters that never receive a gradient
                # (140,300 next to the U-Net's 250,372 — about the proportion of SD v1: 207 M next to 860 M)
                self.first_stage_model = nn.Linear(300, 300)
                self.cond_stage_model = nn.Linear(200, 250, bias=False)
                for q in list(self.first_stage_model.parameters()) + list(self.cond_stage_model.parameters()):
                    q.requires_grad_(False)
                fill_params(self.first_stage_model, 9200)
                fill_params(self.cond_stage_model, 9300)
            self.first_stage_key, self.num_timesteps = "jpg", 1000
            betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120, cosine_s=8e-3)
            ac = np.cumprod(1.0 - betas, axis=0)
            self.register_buffer("sqrt_alphas_cumprod", torch.tensor(np.sqrt(ac), dtype=torch.float32))
            self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.tensor(np.sqrt(1.0 - ac), dtype=torch.float32))
            self.contexts = {k: torch.from_numpy(v) for k, v in sd_glue_contexts().items()}

        @property
        def device(self):
            return self.sqrt_alphas_cumprod.device

        def get_input(self, batch, k):
            x = batch[k].permute(0, 3, 1, 2).contiguous().float()       # ddpm.py get_input: b h w c -> b c h w
            c = torch.stack([self.contexts[p] for p in batch["txt"]])
            return x, c

        def q_sample(self, x_start, t, noise=None):                      # ddpm.py:424-430
            return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
                    + extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)

        def apply_model(self, x_noisy, t, cond):                         # ddpm.py:1121 -> DiffusionWrapper "crossattn"
            return self.model.diffusion_model(x_noisy, t, context=cond)

        def shared_step(self, batch):                                    # ddpm.py:1093-1109, p_losses :1286-1319
            x, c = self.get_input(batch, self.first_stage_key)
            t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
            noise = torch.randn_like(x)
            model_output = self.apply_model(self.q_sample(x_start=x, t=t, noise=noise), t, c)
            loss_simple = torch.nn.functional.mse_loss(noise, model_output, reduction="none").mean([1, 2, 3])
            logvar_t = torch.zeros_like(loss_simple)                     # logvar_init = 0, not learned
            loss = (loss_simple / torch.exp(logvar_t) + logvar_t).mean() * 1.0   # l_simple_weight = 1
            return loss, {}                                              # original_elbo_weight = 0: no vlb term

    return DuckLDM()


# ------------------------------------------------------------------ importing the scripts
def import_scripts():
    _stub("omegaconf")
    _stub("omegaconf.listconfig")
    sys.modules["omegaconf.listconfig"].ListConfig = type("ListConfig", (), {})
    for m in ("diffusers", "convertModels", "dataset", "torchvision", "torchvision.utils", "ldm.models.diffusion.ddim"):
        sys.modules.pop(m, None)
    sys.path.insert(0, REF_SD)
    import ldm.models.diffusion  # noqa: F401  (the real package; only its `ddim` module is replaced)
    dd = types.ModuleType("ldm.models.diffusion.ddim")
    dd.DDIMSampler = lambda model: None
    sys.modules["ldm.models.diffusion.ddim"] = dd
    df = types.ModuleType("diffusers")
    df.LMSDiscreteScheduler = lambda **k: None
    sys.modules["diffusers"] = df
    cm = types.ModuleType("convertModels")
    cm.savemodelDiffusers = lambda *a, **k: None
    sys.modules["convertModels"] = cm
    tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tvu
    ds = types.ModuleType("dataset")
    for name in ("setup_model", "setup_forget_data", "setup_forget_nsfw_data", "setup_remain_data"):
        setattr(ds, name, None)
    sys.modules["dataset"] = ds
    mods = {}
    for name in ("generate_mask", "nsfw_removal", "random_label", "proximal_gradient"):
        sp = importlib.util.spec_from_file_location("ref_sd_" + name, f"{REF_SD}/train-scripts/{name}.py")
        m = importlib.util.module_from_spec(sp)
        sp.loader.exec_module(m)
        if hasattr(m, "sleep"):
            m.sleep = lambda s: None
        mods[name] = m
    return mods, ds


class Recorder:
    """Record torch.randint / torch.randn_like outputs in call order; observe torch.abs_ and torch.optim.Adam."""

    def __init__(self, seed):
        self.seed = seed
        self.randint, self.randn, self.abs_args, self.optimizers = [], [], [], []

    def __enter__(self):
        self.real = (torch.randint, torch.randn_like, torch.abs_, torch.optim.Adam)
        real, rec = self.real, self
        torch.manual_seed(self.seed)

        def randint(*a, **k):
            out = real[0](*a, **k)
            rec.randint.append(out.clone())
            return out

        def randn_like(x, **k):
            out = real[1](x, **k)
            rec.randn.append(out.clone())
            return out

        def abs_(x):
            out = real[2](x)
            rec.abs_args.append(out)
            return out

        class Adam(real[3]):
            def __init__(self, *a, **k):
                super().__init__(*a, **k)
                rec.optimizers.append(self)

        torch.randint, torch.randn_like, torch.abs_, torch.optim.Adam = randint, randn_like, abs_, Adam
        return self

    def __exit__(self, *a):
        torch.randint, torch.randn_like, torch.abs_, torch.optim.Adam = self.real


def flat(params):
    return torch.cat([p.detach().reshape(-1) for p in params]).numpy()


def draws(rec, out, tag):
    out[f"{tag}__randint"] = np.stack([t.numpy() for t in rec.randint]).astype(np.int64)
    out[f"{tag}__randn"] = np.stack([t.numpy() for t in rec.randn]).astype(np.float32)


def adam_state(opt, unet, out, tag):
    n = sum(p.numel() for p in unet.parameters())
    m1, m2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
    off = 0
    for p in unet.parameters():
        st = opt.state.get(p)
        if st:
            m1[off:off + p.numel()] = st["exp_avg"].reshape(-1).numpy()
            m2[off:off + p.numel()] = st["exp_avg_sq"].reshape(-1).numpy()
        off += p.numel()
    out[f"{tag}__exp_avg_s"], out[f"{tag}__exp_avg_sq_s"] = m1[::STRIDE].copy(), m2[::STRIDE].copy()
    out[f"{tag}__exp_avg_sum"] = np.float64(m1.astype(np.float64).sum())
    out[f"{tag}__exp_avg_sq_sum"] = np.float64(m2.astype(np.float64).sum())


def main():
    mods, ds = import_scripts()
    out = {}
    B = 4
    forget3, remain2, forget_nsfw, remain_nsfw = sd_glue_batches()
    descriptions = [f"class {i}" for i in range(10)]
    cwd = os.getcwd()
    work = tempfile.mkdtemp()
    os.chdir(work)
    os.makedirs("mask")  # generate_nsfw_mask saves to mask/nude_0.5.pt without creating the directory (:211)
    try:
        # ---------------------------------------------------------------- A13: generate_nsfw_mask
        model = build_ref_ldm()
        names = [n for n, _ in model.model.diffusion_model.named_parameters()]
        out["param_names"] = np.array(names)
        out["init_sum"] = np.float64(flat(model.model.diffusion_model.parameters()).astype(np.float64).sum())
        ds.setup_model = lambda *a, **k: model
        ds.setup_forget_nsfw_data = lambda bs, size: (forget_nsfw, remain_nsfw)
        G = mods["generate_mask"]
        G.setup_model, G.setup_forget_nsfw_data = ds.setup_model, ds.setup_forget_nsfw_data
        with Recorder(11) as rec:
            G.generate_nsfw_mask(7.5, B, 1, 1e-5, None, None, None, "cpu", image_size=8, num_timesteps=1000)
        hard = torch.load(os.path.join("mask", "nude_0.5.pt"), weights_only=False)
        assert list(hard.keys()) == names
        acc = np.concatenate([t.reshape(-1).numpy() for t in rec.abs_args]).astype(np.float32)
        m = torch.cat([v.reshape(-1) for v in hard.values()]).numpy().astype(np.uint8)
        assert acc.size == m.size and int(m.sum()) == int(acc.size * 0.5)
        draws(rec, out, "nsfw_mask")          # randint: [unused t, t] per batch (generate_mask.py:141-143,160-162)
        out["nsfw_mask__abs_acc"] = acc
        out["nsfw_mask__mask_bits"] = np.packbits(m)
        srt = np.sort(acc)[::-1]
        k = int(acc.size * 0.5)
        out["nsfw_mask__tau_unique"] = np.bool_(srt[k - 1] != srt[k] and srt[k - 1] != srt[k - 2])
        print("generate_nsfw_mask:", acc.size, "params,", len(rec.randint), "randint draws, tau unique:",
              bool(out["nsfw_mask__tau_unique"]))

        # ---------------------------------------------------------------- A13: generate_mask (class version)
        model = build_ref_ldm()
        ds.setup_model = lambda *a, **k: model
        cls_dl = [(z, torch.tensor([3] * z.shape[0])) for z in forget_nsfw]
        G.setup_model = ds.setup_model
        G.setup_forget_data = lambda c, bs, size: (cls_dl, descriptions + [""])
        model.contexts.update({d: model.contexts["a photo of a nude person"] + 0.01 * i
                               for i, d in enumerate(descriptions)})
        with Recorder(12) as rec:
            G.generate_mask(3, 7.5, B, 1, 1e-5, None, None, None, "cpu", image_size=8, num_timesteps=1000)
        hard = torch.load(os.path.join("mask", "3", "with_0.5.pt"), weights_only=False)
        acc = np.concatenate([t.reshape(-1).numpy() for t in rec.abs_args]).astype(np.float32)
        m = torch.cat([v.reshape(-1) for v in hard.values()]).numpy().astype(np.uint8)
        draws(rec, out, "class_mask")
        out["class_mask__abs_acc_s"] = acc[::STRIDE].copy()
        out["class_mask__abs_acc_sum"] = np.float64(acc.astype(np.float64).sum())
        out["class_mask__mask_bits"] = np.packbits(m)
        print("generate_mask: popcount", int(m.sum()))

        # ---------------------------------------------------------------- A14: nsfw_removal (mask, full / xattn)
        mask_bits = np.unpackbits(out["nsfw_mask__mask_bits"])[:out["nsfw_mask__abs_acc"].size]
        for method in ("full", "xattn"):
            model = build_ref_ldm()
            unet = model.model.diffusion_model
            off, mask = 0, {}
            for n, p in unet.named_parameters():
                mask[n] = torch.from_numpy(mask_bits[off:off + p.numel()].astype(np.int64)).view_as(p)
                off += p.numel()
            torch.save(mask, "nude_mask.pt")
            R = mods["nsfw_removal"]
            R.setup_model = lambda *a, **k: model
            R.setup_forget_nsfw_data = lambda bs, size: (forget_nsfw, remain_nsfw)
            with Recorder(21) as rec:
                # the reference's `losses` list is local; its entries are loss.item() / batch_size — observed via .item()
                seen = []
                real_item = torch.Tensor.item

                def item(self):
                    v = real_item(self)
                    if self.requires_grad:
                        seen.append(v)
                    return v
                torch.Tensor.item = item
                try:
                    R.nsfw_removal(method, 0.5, B, 1, 1e-4, None, None, "nude_mask.pt", None, "cpu", image_size=8)
                finally:
                    torch.Tensor.item = real_item
            tag = f"nsfw_removal_{method}"
            draws(rec, out, tag)
            out[f"{tag}__losses"] = np.array(seen[0::2], np.float64)   # .item() is called twice per step (:149,:160)
            w = flat(unet.parameters())
            out[f"{tag}__weights_s"] = w[::STRIDE].copy()
            out[f"{tag}__weights_sum"] = np.float64(w.astype(np.float64).sum())
            adam_state(rec.optimizers[-1], unet, out, tag)
            saved = torch.load(os.path.join("models", f"compvis-nsfw-mask-method_{method}-lr_0.0001",
                                            f"compvis-nsfw-mask-method_{method}-lr_0.0001.pt"), weights_only=False)
            assert all(k.startswith("model.diffusion_model.") or "alphas_cumprod" in k for k in saved)
            print(tag, "losses", out[f"{tag}__losses"])

        # ---------------------------------------------------------------- A14: certain_label (random_label.py), no mask
        model = build_ref_ldm()
        unet = model.model.diffusion_model
        model.contexts.update({d: model.contexts["a photo of a nude person"] + 0.01 * i
                               for i, d in enumerate(descriptions)})
        C = mods["random_label"]
        C.setup_model = lambda *a, **k: model
        fdl = [(z, torch.tensor([3] * z.shape[0])) for z in forget_nsfw]
        rdl = [(z, torch.tensor([(5 + i) % 10 if (5 + i) % 10 != 3 else 4 for i in range(z.shape[0])])) for z in remain_nsfw]
        C.setup_remain_data = lambda c, bs, size: (rdl, descriptions)
        C.setup_forget_data = lambda c, bs, size: (fdl, descriptions)
        seen = []
        real_item = torch.Tensor.item

        def item2(self):
            v = real_item(self)
            if self.requires_grad:
                seen.append(v)
            return v
        with Recorder(31) as rec:
            torch.Tensor.item = item2
            try:
                C.certain_label(3, "full", 0.5, B, 2, 1e-4, None, None, None, None, "cpu", image_size=8)
            finally:
                torch.Tensor.item = real_item
        draws(rec, out, "certain_label")
        out["certain_label__losses"] = np.array(seen[0::2], np.float64)
        out["certain_label__remain_labels"] = np.stack([l.numpy() for _, l in rdl])
        w = flat(unet.parameters())
        out["certain_label__weights_s"] = w[::STRIDE].copy()
        out["certain_label__weights_sum"] = np.float64(w.astype(np.float64).sum())
        adam_state(rec.optimizers[-1], unet, out, "certain_label")
        print("certain_label losses", out["certain_label__losses"])

        # ---------------------------------------------------------------- F2: proximal_gradient (frozen stages count)
        model = build_ref_ldm(frozen=True)
        unet = model.model.diffusion_model
        model.contexts.update({d: model.contexts["a photo of a nude person"] + 0.01 * i
                               for i, d in enumerate(descriptions)})
        n_unet = sum(p.numel() for p in unet.parameters())
        n_all = sum(p.numel() for p in model.parameters())
        init = flat(unet.parameters()).copy()
        P = mods["proximal_gradient"]
        P.setup_model = lambda *a, **k: model
        P.setup_remain_data = lambda c, bs, size: (rdl, descriptions)
        P.setup_forget_data = lambda c, bs, size: (fdl, descriptions)
        seen = []
        ended = ""
        with Recorder(41) as rec:
            torch.Tensor.item = item2
            try:
                P.proximal_gradient(3, "full", 0.5, B, 2, 1e-4, None, None, 0.5, None, "cpu", image_size=8,
                                    second_device="cpu")
            except NameError as e:   # proximal_gradient.py:200 `save_history(losses, name, classes)`: undefined name,
                ended = repr(e)      # raised after training and after the model was saved
            finally:
                torch.Tensor.item = real_item
        draws(rec, out, "proximal")
        out["proximal__losses"] = np.array(seen[0::2], np.float64)
        w = flat(unet.parameters())
        out["proximal__weights_s"] = w[::STRIDE].copy()
        out["proximal__weights_sum"] = np.float64(w.astype(np.float64).sum())
        out["proximal__n_unet"], out["proximal__n_all"] = np.int64(n_unet), np.int64(n_all)
        out["proximal__reset_count"] = np.int64((w == init).sum())
        out["proximal__ended_with"] = np.array(ended)
        out["proximal__mask_ratio"] = np.float64(0.5)
        print("proximal_gradient losses", out["proximal__losses"], "n_unet", n_unet, "n_all", n_all, "reset",
              int(out["proximal__reset_count"]), "ended:", ended)
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "sd_glue.npz"), **out)
    print("wrote sd_glue.npz", os.path.getsize(os.path.join(HERE, "sd_glue.npz")), "bytes")


if __name__ == "__main__":
    main()
