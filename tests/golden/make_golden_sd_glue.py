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
The model handed to the scripts is the REFERENCE's own `LatentDiffusion` (ldm/models/diffusion/ddpm.py:605-) around
its `UNetModel` (openaimodel.py:428-847): `get_input`, `q_sample` (:424-430), `apply_model` (:1121), `shared_step` /
`forward` / `p_losses` (:1093-1109, :1286-1319) are executed, not restated.  `pytorch_lightning` is absent from the
image; `LightningModule` is stood in by `nn.Module` + a `device` property (round 4; rounds 2-3 used a duck-typed module
that re-wrote those five members — the fixture did not change by a bit when the real class replaced it).  "Images" are
latents and "prompts" index a table of fixed context embeddings: the frozen VAE / CLIP are outside the hot-path scope,
so `first_stage_config` / `cond_stage_config` point at an identity encoder and a prompt table; `get_input` is
deterministic, so `pseudo_input == forget_input` as in this build's formulation.  For `proximal_gradient` the two
stages carry parameters that never move, because the reference ranks |theta - theta_0| over `model.parameters()` of
the WHOLE model.

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


# ------------------------------------------------------------------ the reference's LatentDiffusion
def _stub_lightning_and_stages():
    """What `ldm.models.diffusion.ddpm` imports but the image lacks: pytorch_lightning (its LightningModule is used as an
    nn.Module with a `device` property; nothing of the Trainer is touched by the SalUn scripts) and
    `ldm.models.autoencoder` (taming / lightning; only class NAMES are imported from it).  The frozen stages themselves
    (AutoencoderKL, CLIP) are outside the hot-path scope: `first_stage_config` / `cond_stage_config` point at the stand-ins
    below — an identity encoder ("images" are latents) and a prompt -> fixed-embedding table."""
    if "pytorch_lightning" not in sys.modules or not hasattr(sys.modules["pytorch_lightning"], "LightningModule"):
        pl = types.ModuleType("pytorch_lightning")

        class LightningModule(nn.Module):
            @property
            def device(self):
                for t in list(self.parameters()) + list(self.buffers()):
                    return t.device
                return torch.device("cpu")

        pl.LightningModule = LightningModule
        util = types.ModuleType("pytorch_lightning.utilities")
        dist = types.ModuleType("pytorch_lightning.utilities.distributed")
        dist.rank_zero_only = lambda fn: fn
        util.distributed = dist
        pl.utilities = util
        sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": util,
                            "pytorch_lightning.utilities.distributed": dist})
    ae = types.ModuleType("ldm.models.autoencoder")
    for name in ("AutoencoderKL", "IdentityFirstStage", "VQModelInterface"):
        setattr(ae, name, type(name, (nn.Module,), {}))
    sys.modules["ldm.models.autoencoder"] = ae
    st = types.ModuleType("salun_golden_stages")

    class LatentsAsImages(nn.Module):
        """first stage: `encode` is the identity (the data handed to the scripts are latents)."""

        def encode(self, x):
            return x

        def decode(self, z):
            return z

    class FrozenLatentsAsImages(nn.Linear):
        """the same with 90,300 parameters that never receive a gradient (proximal_gradient ranks over them too)."""

        def __init__(self):
            super().__init__(300, 300)

        def encode(self, x):
            return x

    class PromptTable(nn.Module):
        """cond stage: prompt -> fixed (7, 24) context embedding."""

        def __init__(self):
            super().__init__()
            self.contexts = {k: torch.from_numpy(v) for k, v in sd_glue_contexts().items()}

        def encode(self, prompts):
            return torch.stack([self.contexts[p] for p in prompts])

    class FrozenPromptTable(PromptTable):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(200, 250, bias=False)  # 50,000 parameters that never receive a gradient

    st.LatentsAsImages, st.FrozenLatentsAsImages = LatentsAsImages, FrozenLatentsAsImages
    st.PromptTable, st.FrozenPromptTable = PromptTable, FrozenPromptTable
    sys.modules["salun_golden_stages"] = st


def build_ref_ldm(frozen: bool = False):
    """The REFERENCE's `LatentDiffusion` (ldm/models/diffusion/ddpm.py:605-), constructed as `v1-inference.yaml`
    constructs it (crossattn conditioning, eps-parameterisation, linear schedule 0.00085 .. 0.0120, no EMA) around the
    reference's `UNetModel` at the reduced size: its own `get_input`, `q_sample`, `apply_model`, `shared_step` /
    `forward` / `p_losses` run — nothing of them is restated here."""
    _stub_lightning_and_stages()
    from ldm.models.diffusion.ddpm import LatentDiffusion
    model = LatentDiffusion(
        first_stage_config={"target": "salun_golden_stages." + ("FrozenLatentsAsImages" if frozen else "LatentsAsImages")},
        cond_stage_config={"target": "salun_golden_stages." + ("FrozenPromptTable" if frozen else "PromptTable")},
        unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.UNetModel", "params": sd_glue_config()},
        conditioning_key="crossattn", cond_stage_key="txt", first_stage_key="jpg", cond_stage_trainable=False,
        scale_factor=1.0, timesteps=1000, linear_start=0.00085, linear_end=0.0120, image_size=8, channels=4,
        use_ema=False, monitor=None)
    fill_params(model.model.diffusion_model, 9100)
    if frozen:
        fill_params(model.first_stage_model, 9200)
        fill_params(model.cond_stage_model, 9300)
        assert not any(q.requires_grad for q in list(model.first_stage_model.parameters())
                       + list(model.cond_stage_model.parameters()))
    model.contexts = model.cond_stage_model.contexts  # the script below adds class prompts to the table
    return model


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
            assert sum(k.startswith("model.diffusion_model.") for k in saved) == len(names)
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
