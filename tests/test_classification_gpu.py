"""GPU end-to-end parity of the Classification plugin surface: the reference-shaped API
(`unlearn.get_unlearn_method("RL")`, `generate_mask.save_gradient_ratio`) running the HIP path,
against (i) golden vectors captured from the reference and (ii) the CPU oracle.
Tolerance for trained weights / losses: 1e-5 relative (north_star); masks bit-exact."""
import json
import os
import tempfile
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from fixtures import TinyCNN, saliency_vector, saliency_vector_wide, tiny_batches, tiny_state

pytestmark = pytest.mark.gpu
RATIOS = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]


class ListLoader(list):
    """Any iterable of (x, y) batches with a .dataset works as a loader for the plugins."""

    def __init__(self, batches):
        super().__init__(batches)
        self.dataset = SimpleNamespace(targets=np.zeros(0))


def _loader(batches, device="cuda"):
    return ListLoader([(torch.from_numpy(x).to(device), torch.from_numpy(np.asarray(y)).to(device))
                       for x, y in batches])


def _args(**kw):
    base = dict(unlearn_lr=0.013, momentum=0.9, weight_decay=5e-4, decreasing_lr="91,136", rewind_epoch=0,
                imagenet_arch=False, unlearn="RL", unlearn_epochs=2, dataset="cifar10", num_classes=10, warmup=0,
                print_freq=50, batch_size=16, alpha=0.2, no_l1_epochs=0)
    base.update(kw)
    return SimpleNamespace(**base)


@pytest.fixture(autouse=True)
def _deterministic():
    torch.backends.cudnn.deterministic = True
    yield


def test_flat_arena_views_and_mask_roundtrip():
    from unlearn_saliency_amd.flat import FlatArena
    model = TinyCNN().cuda()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    arena = FlatArena.from_module(model)
    assert arena.n == sum(p.numel() for p in model.parameters())
    assert arena.params.data_ptr() % 256 == 0
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k])
    # parameters are views: writing the flat vector changes the module
    arena.params.mul_(2.0)
    assert torch.equal(model.conv1.weight, before["conv1.weight"] * 2)
    # grads accumulate into the flat vector
    x = torch.rand(4, 3, 8, 8, device="cuda")
    arena.zero_grad()
    model(x).sum().backward()
    g1 = arena.grads.clone()
    assert g1.abs().sum() > 0 and model.fc.bias.grad.data_ptr() == arena.grads[-10:].data_ptr()
    model(x).sum().backward()
    assert torch.allclose(arena.grads, 2 * g1, rtol=1e-5, atol=1e-6)
    # mask dict <-> flat u8
    flat = (torch.rand(arena.n, device="cuda") < 0.5).to(torch.uint8)
    d = arena.unpack_mask(flat)
    assert list(d) == [n for n, _ in model.named_parameters()]
    assert all(v.dtype == torch.int64 and v.shape == p.shape for v, p in zip(d.values(), model.parameters()))
    assert torch.equal(arena.pack_mask(d), flat)
    assert torch.equal(arena.pack_mask({k: v.cpu() for k, v in d.items()}), flat)  # CPU masks accepted
    with pytest.raises(KeyError):
        arena.pack_mask({})


def test_save_gradient_ratio_matches_reference_golden(golden_dir):
    """Phase A through the reference-named function: Σ∇(−CE) over 3 batches (ragged last) -> 10 mask files."""
    from unlearn_saliency_amd.Classification import generate_mask as gm
    g = np.load(os.path.join(golden_dir, "saliency_tinycnn.npz"))
    model = TinyCNN()
    model.load_state_dict(tiny_state(11))
    model.cuda()
    batches = tiny_batches(3, 16, 500)
    batches[-1] = (batches[-1][0][:9], batches[-1][1][:9])
    loaders = {"forget": _loader(batches)}
    acc = gm.accumulate_saliency(loaders["forget"], model, nn.CrossEntropyLoss())
    # 1e-5 relative to the vector's scale: individual components are sums with cancellation
    assert np.allclose(acc.cpu().numpy(), g["acc"], rtol=1e-5, atol=1e-5 * float(np.abs(g["acc"]).max()))
    with tempfile.TemporaryDirectory() as d:
        gm.save_gradient_ratio(loaders, model, nn.CrossEntropyLoss(), SimpleNamespace(save_dir=d, thresholds=None))
        files = sorted(os.listdir(d))
        assert files == sorted(f"with_{r}.pt" for r in RATIOS)
        m = torch.load(os.path.join(d, "with_0.5.pt"), weights_only=False)
        assert list(m.keys()) == list(g["names"])
        assert all(v.dtype == torch.int64 for v in m.values())
        flat = np.concatenate([v.reshape(-1).cpu().numpy() for v in m.values()]).astype(np.uint8)
        # the golden mask was computed from the reference's own accumulator; ours may differ from it only where
        # two saliencies are within float rounding of each other around the threshold
        assert int(flat.sum()) == int(g["mask_05"].sum())
        assert (flat != g["mask_05"]).sum() <= 2
        one = torch.load(os.path.join(d, "with_1.0.pt"), weights_only=False)
        assert all(bool(v.all()) for v in one.values())


def test_masks_from_saliency_resnet18_hashes(golden_dir):
    """HIP top-k on the N18 vector == masks the REFERENCE produced for the same vector (hash pinned)."""
    import hashlib
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.Classification.generate_mask import masks_from_saliency
    fx = json.load(open(os.path.join(golden_dir, "classification.json")))
    for key in ("mask_mid", "mask_resnet18", "mask_resnet18_wide"):
        if key not in fx:
            continue
        f = fx[key]
        n = f["n"]
        wide = key.endswith("wide")
        z = ops.fill_normal(n, f["seed"], 0.0, 1.0 if wide else f["std"])
        u = ops.fill_uniform(n, f["seed"] + 7, 0.0, 0.5)
        sal = z * (1.0 + u)
        if wide:
            j = torch.floor(ops.fill_uniform(n, f["seed"] + 13, 0.0, 40.0)).to(torch.int32) - 20
            # exact 2^j from its bit pattern (torch.ldexp goes through pow(), which is not exact on the GPU)
            sal = sal * ((j + 127) << 23).view(torch.float32)
        # the device-built vector is the very vector the reference saw (bitwise), checked on a prefix
        head = 200_000
        want = (saliency_vector_wide(head, f["seed"]) if wide else saliency_vector(head, f["seed"], f["std"]))
        if not wide or True:
            full_head = sal[:head].cpu().numpy()
            # generators are index-addressed, so a prefix of the big vector equals the small vector
            assert np.array_equal(full_head.view(np.uint32), want.view(np.uint32))
        masks = masks_from_saliency(sal, RATIOS)
        for r in RATIOS:
            m = masks[r].cpu().numpy()
            assert int(m.sum()) == f["popcount"][str(r)]
            if f["tau_unique"][str(r)]:
                assert hashlib.sha256(np.packbits(m).tobytes()).hexdigest() == f["sha256"][str(r)], (key, r)


@pytest.mark.parametrize("tag,use_mask", [("masked", True), ("unmasked", False)])
def test_rl_plugin_matches_reference_epoch(golden_dir, monkeypatch, tag, use_mask):
    """unlearn.get_unlearn_method('RL') on the HIP path vs the reference's RL on the same batches and the same
    random labels (2 epochs x (2 forget + 3 retain) steps, BN in train mode)."""
    from unlearn_saliency_amd.Classification import unlearn
    g = np.load(os.path.join(golden_dir, f"rl_epoch_{tag}.npz"))
    model = TinyCNN()
    init = tiny_state(21)
    model.load_state_dict(init)
    model.cuda()
    names = [n for n, _ in model.named_parameters()]
    sizes = [p.numel() for p in model.parameters()]
    mask = None
    if use_mask:
        off = np.cumsum([0] + sizes)
        mask = {n: torch.from_numpy(g["mask"][off[i]:off[i + 1]].astype(np.int64)).view_as(p)
                for i, (n, p) in enumerate(model.named_parameters())}  # CPU int64, the reference's artefact format
    labels = [torch.from_numpy(l) for l in g["random_labels"]]
    monkeypatch.setattr(torch, "randint", lambda *a, **k: labels.pop(0))
    loaders = {"forget": _loader(tiny_batches(2, 16, 700)), "retain": _loader(tiny_batches(3, 16, 800))}
    unlearn.get_unlearn_method("RL")(loaders, model, nn.CrossEntropyLoss(), _args(), mask)
    assert not labels
    sd = model.state_dict()
    for k, v in sd.items():
        ref = g["sd_" + k]
        assert np.allclose(v.cpu().numpy(), ref, rtol=1e-5, atol=1e-6), k
    if use_mask:
        now = np.concatenate([sd[n].reshape(-1).cpu().numpy() for n in names])
        was = np.concatenate([init[n].reshape(-1).numpy() for n in names])
        frozen = g["mask"] == 0
        assert np.array_equal(now[frozen].view(np.uint32), was[frozen].view(np.uint32))  # bit-identical to theta0


def test_ga_and_ft_plugins_run_and_respect_mask():
    from unlearn_saliency_amd.Classification import unlearn
    for name in ("GA", "FT", "FT_l1", "GA_l1"):
        model = TinyCNN()
        init = tiny_state(31)
        model.load_state_dict(init)
        model.cuda()
        n = sum(p.numel() for p in model.parameters())
        flat = (np.arange(n) % 3 == 0).astype(np.int64)
        off = np.cumsum([0] + [p.numel() for p in model.parameters()])
        mask = {k: torch.from_numpy(flat[off[i]:off[i + 1]]).view_as(p).cuda()
                for i, (k, p) in enumerate(model.named_parameters())}
        loaders = {"forget": _loader(tiny_batches(2, 16, 700)), "retain": _loader(tiny_batches(2, 16, 800))}
        unlearn.get_unlearn_method(name)(loaders, model, nn.CrossEntropyLoss(), _args(unlearn=name, unlearn_epochs=1),
                                         mask)
        now = np.concatenate([p.detach().reshape(-1).cpu().numpy() for p in model.parameters()])
        was = np.concatenate([init[k].reshape(-1).numpy() for k, _ in model.named_parameters()])
        assert np.array_equal(now[flat == 0], was[flat == 0]), name
        assert (now[flat == 1] != was[flat == 1]).mean() > 0.5, name


def test_registry_surface():
    from unlearn_saliency_amd.Classification import unlearn
    for name in ("raw", "RL", "GA", "FT", "FT_l1", "fisher", "retrain", "fisher_new", "wfisher", "FT_prune",
                 "FT_prune_bi", "GA_prune", "GA_prune_bi", "GA_l1", "boundary_expanding", "boundary_shrink",
                 "RL_proximal"):
        assert callable(unlearn.get_unlearn_method(name))
    with pytest.raises(NotImplementedError):
        unlearn.get_unlearn_method("nope")
    assert unlearn.get_unlearn_method("raw")({}, None, None, None) is None


def test_device_loader_matches_host_loader():
    from unlearn_saliency_amd.Classification.dataset import ArrayDataset, BatchLoader, synthetic_cifar10
    (xtr, ytr), _ = synthetic_cifar10(n_train=700, n_test=10)
    ds = ArrayDataset(xtr, ytr, transform="test")
    host = BatchLoader(ds, 256, shuffle=False)
    dev = BatchLoader(ds, 256, shuffle=False, device_resident=True, device=torch.device("cuda"))
    for (xh, yh), (xd, yd) in zip(host, dev):
        assert torch.equal(xh, xd.cpu()) and torch.equal(yh, yd.cpu())
    assert len(host) == len(dev) == 3


def test_validate_accuracy():
    from unlearn_saliency_amd.Classification.trainer import validate
    model = TinyCNN()
    model.load_state_dict(tiny_state(11))
    model.cuda()
    batches = tiny_batches(3, 16, 500)
    acc = validate(_loader(batches), model, nn.CrossEntropyLoss(), SimpleNamespace(print_freq=50))
    model.eval()
    with torch.no_grad():
        hits = sum(int((model(torch.from_numpy(x).cuda()).argmax(1).cpu() == torch.from_numpy(y)).sum())
                   for x, y in batches)
    assert abs(acc - 100.0 * hits / 48) < 1e-9


def test_projection_branch_beside_the_main_chain_changes_nothing():
    """resblock._BasicBlockFn.forward issues the 1x1 stride-2 projection + its BatchNorm of a strided block on the side
    stream (round 6): outputs, running statistics and every gradient are bit-identical to the one-stream order."""
    import torch
    from unlearn_saliency_amd import resblock
    from unlearn_saliency_amd.Classification.models import resnet_cifar
    from unlearn_saliency_amd.conv import use_salun_convs

    def run(flag):
        torch.manual_seed(7)
        net = resnet_cifar.resnet18(num_classes=10).cuda()
        use_salun_convs(net)
        net.train()
        x = torch.randn(64, 3, 32, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
        old = resblock.FWD_SHORTCUT_BESIDE
        resblock.FWD_SHORTCUT_BESIDE = flag
        try:
            y = net(x)
            y.square().mean().backward()
            torch.cuda.synchronize()
        finally:
            resblock.FWD_SHORTCUT_BESIDE = old
        bufs = [b.clone() for b in net.buffers()]
        return [y.detach().clone()] + [p.grad.clone() for p in net.parameters()] + bufs

    a, b = run(True), run(False)
    assert len(a) == len(b) and all(torch.equal(u, v) for u, v in zip(a, b))
