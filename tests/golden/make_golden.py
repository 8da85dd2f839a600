"""Golden-vector generator: runs the REFERENCE's own functions (imported from /root/reference,
build container only) on inputs from the counter-based generator and stores inputs' seeds +
expected outputs as small fixtures in this directory.

    python tests/golden/make_golden.py [classification] [ddpm] [sd] [--big]

The reference cannot be imported as shipped: torchvision
and lmdb are absent and `trainer/__init__` imports a symbol that does not exist, so the
import goes through permissive stub modules that are never called on the paths exercised
here.  Nothing in this file is copied from the reference; it only *calls* it.  The
fixtures are data (inputs + expected outputs), never source.
"""
from __future__ import annotations

import hashlib
import importlib.machinery as M
import importlib.util
import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from unlearn_saliency_amd import rng  # noqa: E402  (counter-based inputs, no reference code)

REF = "/root/reference"
sys.dont_write_bytecode = True
torch.set_num_threads(8)


# ------------------------------------------------------------------ import shims
class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()


def _stub(name):
    m = types.ModuleType(name)
    m.__file__ = "/dev/null/" + name
    m.__spec__ = M.ModuleSpec(name, None)
    m.__path__ = []

    def _getattr(n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any

    m.__getattr__ = _getattr
    sys.modules[name] = m
    if "." in name:  # make `import a.b.c as x` resolve through the parent stub
        parent, leaf = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, m)
    return m


def _load(mod, path):
    sp = importlib.util.spec_from_file_location(mod, path)
    m = importlib.util.module_from_spec(sp)
    sys.modules[mod] = m
    sp.loader.exec_module(m)
    return m


def import_reference_classification():
    _stub("torchvision")
    for s in ("transforms", "transforms.functional", "datasets", "models", "utils"):
        _stub("torchvision." + s)
    for n in ("CIFAR10", "CIFAR100", "SVHN", "STL10", "ImageFolder"):
        setattr(sys.modules["torchvision.datasets"], n, type(n, (), {}))
    _stub("lmdb")
    C = REF + "/Classification"
    sys.path.insert(0, C)
    pkg = types.ModuleType("trainer")
    pkg.__path__ = [C + "/trainer"]
    pkg.__spec__ = M.ModuleSpec("trainer", None, is_package=True)
    sys.modules["trainer"] = pkg
    tt = _load("trainer.train", C + "/trainer/train.py")
    tt.train_with_rewind = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    tv = _load("trainer.val", C + "/trainer/val.py")
    pkg.train, pkg.validate = tt.train, tv.validate
    pkg.get_optimizer_and_scheduler, pkg.train_with_rewind = tt.get_optimizer_and_scheduler, tt.train_with_rewind
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import generate_mask as ref_generate_mask
    import unlearn as ref_unlearn
    return ref_generate_mask, ref_unlearn


# ------------------------------------------------------------------- helpers
def packbits_sha(mask_u8: np.ndarray) -> str:
    return hashlib.sha256(np.packbits(mask_u8.astype(np.uint8)).tobytes()).hexdigest()


class _ListLoader(list):
    """list of (x, y) batches with the attributes the reference touches (.dataset, len())."""

    def __init__(self, batches):
        super().__init__(batches)
        self.dataset = SimpleNamespace(targets=np.concatenate([np.asarray(b[1]).reshape(-1) for b in batches])
                                       if batches else np.zeros(0))


class LinearProbe(nn.Module):
    """loss = -criterion(model(w), .) = -<theta, w>  =>  grad of every parameter is exactly -w's
    slice: lets an arbitrary saliency vector be pushed through the reference's own
    save_gradient_ratio (abs -> cat -> argsort -> argsort -> compare -> torch.save)."""

    def __init__(self, shapes):
        super().__init__()
        self.ps = nn.ParameterList([nn.Parameter(torch.zeros(s)) for s in shapes])

    def forward(self, w):
        off, tot = 0, 0.0
        for p in self.ps:
            k = p.numel()
            tot = tot + (p.reshape(-1) * w[off:off + k]).sum()
            off += k
        return tot


def reference_masks(ref_gm, shapes, saliency: np.ndarray, batches=None):
    """Run the reference save_gradient_ratio on a LinearProbe; returns {ratio: flat u8 mask}."""
    model = LinearProbe(shapes)
    batches = batches or [saliency]
    loader = _ListLoader([(torch.from_numpy(np.ascontiguousarray(b)), torch.zeros(1)) for b in batches])
    with tempfile.TemporaryDirectory() as d:
        args = SimpleNamespace(unlearn_lr=0.01, momentum=0.9, weight_decay=5e-4, save_dir=d)
        ref_gm.save_gradient_ratio({"forget": loader}, model, lambda out, tgt: out, args)
        out = {}
        for r in [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]:
            md = torch.load(os.path.join(d, f"with_{r}.pt"), weights_only=False)
            assert all(v.dtype == torch.int64 for v in md.values())
            assert [tuple(v.shape) for v in md.values()] == [tuple(s) for s in shapes]
            out[r] = np.concatenate([v.reshape(-1).numpy() for v in md.values()]).astype(np.uint8)
    return out


RESNET18_SHAPES = None


def resnet18_shapes():
    global RESNET18_SHAPES
    if RESNET18_SHAPES is None:
        from unlearn_saliency_amd.Classification.models import model_dict
        RESNET18_SHAPES = [tuple(p.shape) for p in model_dict["resnet18"](num_classes=10).parameters()]
    return RESNET18_SHAPES


from fixtures import TinyCNN, saliency_vector, saliency_vector_wide, tau_is_unique, tiny_batches, tiny_state  # noqa: E402  (tests/fixtures.py)


# ------------------------------------------------------------ classification
def make_classification(big: bool):
    ref_gm, ref_unlearn = import_reference_classification()
    fx = {}

    # (1a) toy 3-tensor mask KAT, unique values
    shapes = [(3, 4), (5,), (2, 3, 3)]
    n = sum(int(np.prod(s)) for s in shapes)
    sal = rng.normal(n, 101, 0.0, 1e-3)
    assert len(np.unique(np.abs(sal))) == n
    masks = reference_masks(ref_gm, shapes, sal)
    np.savez(os.path.join(HERE, "mask_toy.npz"), saliency=sal, shapes=np.array([str(s) for s in shapes]),
             **{f"mask_{r}": m for r, m in masks.items()})

    # (1b) mid-size unique-threshold KAT (N = 200,003 over 7 tensors), inputs regenerate from the seed
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (128,), (10, 512), (10,), (119275,)]
    n = sum(int(np.prod(s)) for s in shapes)
    sal = saliency_vector(n, 202, 1e-3)
    masks = reference_masks(ref_gm, shapes, sal)
    srt = np.sort(np.abs(sal))[::-1]
    fx["mask_mid"] = dict(n=n, seed=202, std=1e-3, shapes=[list(s) for s in shapes],
                          sha256={str(r): packbits_sha(m) for r, m in masks.items()},
                          popcount={str(r): int(m.sum()) for r, m in masks.items()},
                          tau_unique={str(r): bool(tau_is_unique(srt, int(n * r))) for r in masks})

    # (1c) tie cases: zeros block + quantised block.  The reference's argsort is unstable, so only
    # popcounts and the elements whose |value| differs from the threshold value are pinned.
    shapes = [(1000,), (50, 40), (3000,)]
    n = 6000
    sal = rng.normal(n, 303, 0.0, 1.0)
    sal[500:2500] = 0.0
    sal[3000:5000] = np.round(sal[3000:5000] * 4) / 4
    masks = reference_masks(ref_gm, shapes, sal)
    np.savez(os.path.join(HERE, "mask_ties.npz"), saliency=sal, shapes=np.array([str(s) for s in shapes]),
             **{f"mask_{r}": m for r, m in masks.items()})

    # (1d) accumulation over several batches through the probe (sum of 3 "gradients", sign flips)
    shapes = [(7, 9), (11,)]
    n = 74
    bs = [rng.normal(n, 404 + i, 0.0, 1.0) for i in range(3)]
    masks = reference_masks(ref_gm, shapes, None, batches=bs)
    np.savez(os.path.join(HERE, "mask_accum.npz"), batches=np.stack(bs), shapes=np.array([str(s) for s in shapes]),
             **{f"mask_{r}": m for r, m in masks.items()})

    # (1e) ResNet-18-shaped vector, 62 tensors, N = 11,173,962: hashes + per-tensor popcounts
    if big:
        shapes = resnet18_shapes()
        n = sum(int(np.prod(s)) for s in shapes)
        assert n == 11_173_962
        sal = saliency_vector(n, 2024, 1e-3)
        masks = reference_masks(ref_gm, shapes, sal)
        offs = np.cumsum([0] + [int(np.prod(s)) for s in shapes])
        srt = np.sort(np.abs(sal))[::-1]
        fx["mask_resnet18"] = dict(
            n=n, seed=2024, std=1e-3, tau_unique={str(r): bool(tau_is_unique(srt, int(n * r))) for r in masks},
            sha256={str(r): packbits_sha(m) for r, m in masks.items()},
            popcount={str(r): int(m.sum()) for r, m in masks.items()},
            per_tensor_popcount={str(r): [int(m[offs[i]:offs[i + 1]].sum()) for i in range(len(shapes))]
                                 for r, m in masks.items()})
        sal = saliency_vector_wide(n, 4048)
        masks = reference_masks(ref_gm, shapes, sal)
        srt = np.sort(np.abs(sal))[::-1]
        fx["mask_resnet18_wide"] = dict(
            n=n, seed=4048, tau_unique={str(r): bool(tau_is_unique(srt, int(n * r))) for r in masks},
            sha256={str(r): packbits_sha(m) for r, m in masks.items()},
            popcount={str(r): int(m.sum()) for r, m in masks.items()},
            per_tensor_popcount={str(r): [int(m[offs[i]:offs[i + 1]].sum()) for i in range(len(shapes))]
                                 for r, m in masks.items()})
    elif os.path.exists(os.path.join(HERE, "classification.json")):
        old = json.load(open(os.path.join(HERE, "classification.json")))
        for key in ("mask_resnet18", "mask_resnet18_wide"):
            if key in old:
                fx[key] = old[key]

    # k = int(N * r) table
    fx["k_table"] = {str(N): {str(r): int(N * r) for r in [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]}
                     for N in (11_173_962, 38_632_323, 859_520_964, 6000, 200_003)}

    # (2) A1: saliency accumulation of a real (tiny) BN network, captured at the reference's abs_ call
    model = TinyCNN()
    model.load_state_dict(tiny_state(11))
    batches = tiny_batches(3, 16, 500)
    batches[-1] = (batches[-1][0][:9], batches[-1][1][:9])  # ragged last batch
    loader = _ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in batches])
    captured = []
    real_abs_ = torch.abs_
    torch.abs_ = lambda t: (captured.append(t.clone()), real_abs_(t))[1]
    try:
        with tempfile.TemporaryDirectory() as d:
            args = SimpleNamespace(unlearn_lr=0.01, momentum=0.9, weight_decay=5e-4, save_dir=d)
            ref_gm.save_gradient_ratio({"forget": loader}, model, nn.CrossEntropyLoss(), args)
            m05 = torch.load(os.path.join(d, "with_0.5.pt"), weights_only=False)
    finally:
        torch.abs_ = real_abs_
    acc = np.concatenate([t.reshape(-1).numpy() for t in captured])
    np.savez(os.path.join(HERE, "saliency_tinycnn.npz"), acc=acc,
             mask_05=np.concatenate([v.reshape(-1).numpy() for v in m05.values()]).astype(np.uint8),
             names=np.array(list(m05.keys())))

    # (3) A4+A5: _apply_mask_to_grads -> SGD.step -> _restore_masked_params, N = 4096, 3 steps
    ref_RL = sys.modules["unlearn.RL"]  # the submodule (the package attribute `RL` is the plugin function)
    N = 4096
    p0 = rng.normal(N, 600, 0.0, 0.05)
    mask = (rng.u8(N, 601) & 1).astype(np.int64)
    holder = nn.Module()
    holder.w = nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD(holder.parameters(), 0.013, momentum=0.9, weight_decay=5e-4)
    maskd = {"w": torch.from_numpy(mask)}
    theta0 = {"w": holder.w.detach().clone()}
    ps, bufs = [], []
    for step in range(3):
        holder.w.grad = torch.from_numpy(rng.normal(N, 610 + step, 0.0, 1e-2))
        ref_RL._apply_mask_to_grads(holder, maskd)
        opt.step()
        ref_RL._restore_masked_params(holder, maskd, theta0, opt)
        ps.append(holder.w.detach().numpy().copy())
        bufs.append(opt.state[holder.w]["momentum_buffer"].numpy().copy())
    np.savez(os.path.join(HERE, "sgd_step.npz"), p0=p0, mask=mask.astype(np.uint8), p=np.stack(ps),
             buf=np.stack(bufs), lr=0.013, momentum=0.9, weight_decay=5e-4, grad_seeds=np.array([610, 611, 612]),
             grad_std=1e-2)

    # (4) RL epoch: tiny model, 2 forget + 3 retain batches, random labels captured
    for tag, use_mask in (("masked", True), ("unmasked", False)):
        model = TinyCNN()
        model.load_state_dict(tiny_state(21))
        fb = tiny_batches(2, 16, 700)
        rb = tiny_batches(3, 16, 800)
        forget = _ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in fb])
        retain = _ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in rb])
        names = [n for n, _ in model.named_parameters()]
        sizes = [p.numel() for p in model.parameters()]
        mflat = (rng.u8(sum(sizes), 900) & 1).astype(np.int64)
        off = np.cumsum([0] + sizes)
        maskd = {n: torch.from_numpy(mflat[off[i]:off[i + 1]]).view_as(p)
                 for i, (n, p) in enumerate(model.named_parameters())} if use_mask else None
        drawn = []
        real_randint = torch.randint

        def rec_randint(*a, **k):
            t = real_randint(*a, **k)
            drawn.append(t.clone())
            return t

        torch.manual_seed(5)
        torch.randint = rec_randint
        try:
            args = SimpleNamespace(unlearn_lr=0.013, momentum=0.9, weight_decay=5e-4, decreasing_lr="91,136",
                                   rewind_epoch=0, imagenet_arch=False, unlearn="RL", unlearn_epochs=2,
                                   dataset="cifar10", num_classes=10, warmup=0, print_freq=50, batch_size=16)
            ref_unlearn.RL({"forget": forget, "retain": retain}, model, nn.CrossEntropyLoss(), args, maskd)
        finally:
            torch.randint = real_randint
        sd = model.state_dict()
        np.savez(os.path.join(HERE, f"rl_epoch_{tag}.npz"),
                 random_labels=np.stack([d.numpy() for d in drawn]),
                 mask=mflat.astype(np.uint8) if use_mask else np.zeros(0, np.uint8),
                 **{"sd_" + k: v.numpy() for k, v in sd.items()})

    with open(os.path.join(HERE, "classification.json"), "w") as f:
        json.dump(fx, f, indent=1, sort_keys=True)
    print("classification fixtures written")


if __name__ == "__main__":
    what = [a for a in sys.argv[1:] if not a.startswith("--")] or ["classification"]
    big = "--big" in sys.argv
    if "classification" in what:
        make_classification(big)
    if "ddpm" in what:
        from make_golden_ddpm import make_ddpm
        make_ddpm()
    if "sd" in what:
        from make_golden_sd import make_sd
        make_sd()
