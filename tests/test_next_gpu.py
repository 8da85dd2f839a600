""""Next" rows on the HIP path: the proximal / EWC kernels bit-exact vs the oracle, and the
boundary_shrink / boundary_expanding / RL_proximal plugins vs outputs of the reference's own functions."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from fixtures import TinyCNN, ewc_inputs, next_rows_datasets, tiny_batches, tiny_state
from unlearn_saliency_amd import rng

pytestmark = pytest.mark.gpu


class ListLoader(list):
    def __init__(self, batches):
        super().__init__(batches)
        self.dataset = SimpleNamespace(targets=np.concatenate([np.asarray(b[1].cpu()).reshape(-1) for b in batches]))


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


def _check_state(model, g, rtol=1e-5, atol=2e-7):  # 3x the worst use measured on the MI355X (tools/measure_tolerances.py)
    for k, v in model.state_dict().items():
        assert np.allclose(v.cpu().numpy(), g["sd_" + k], rtol=rtol, atol=atol), k


# ------------------------------------------------------------------------------------ kernels
def test_proximal_step_matches_reference_expressions(golden_dir):
    from unlearn_saliency_amd import ops
    g = np.load(os.path.join(golden_dir, "proximal_step.npz"))
    n = int(g["n"])
    p0 = rng.normal(n, int(g["seeds"][0]), 0.0, 0.05)
    p_init = (p0 + rng.normal(n, int(g["seeds"][1]), 0.0, 0.01)).astype(np.float32)
    for ratio, out, tau in zip(g["ratios"], g["out"], g["tau"]):
        p = torch.from_numpy(p_init.copy()).cuda()
        t = ops.proximal_step(p, torch.from_numpy(p0).cuda(), int(ratio))
        assert np.float32(t.item()) == np.float32(tau)
        assert np.array_equal(p.cpu().numpy().view(np.uint32), out.view(np.uint32))
    with pytest.raises(IndexError):
        ops.proximal_step(torch.from_numpy(p_init.copy()).cuda(), torch.from_numpy(p0).cuda(), 0)


@pytest.mark.parametrize("n,offset", [(1, 0), (1023, 0), (4099, 1), (1_000_003, 0), (11_173_962, 0)])
def test_proximal_step_vs_oracle(oracle_mod, n, offset):
    """bit-exact against the numpy restatement, incl. unaligned views and ragged tails; at N18 also the
    size-independent property: at least `ratio` weights land exactly on theta0, the rest moved by exactly tau."""
    from unlearn_saliency_amd import ops
    p0 = rng.normal(n, 5, 0.0, 0.05)
    p = (p0 + rng.normal(n, 6, 0.0, 0.01)).astype(np.float32)
    ratio = max(1, n // 3)
    pd = torch.zeros(n + offset, device="cuda")[offset:]
    p0d = torch.zeros(n + offset, device="cuda")[offset:]
    pd.copy_(torch.from_numpy(p)); p0d.copy_(torch.from_numpy(p0))
    t = ops.proximal_step(pd, p0d, ratio)
    ref = p.copy()
    thr = oracle_mod.soft_threshold_step(ref, p0, ratio)
    assert np.float32(t.item()) == np.float32(thr)
    got = pd.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert int((got == p0).sum()) >= ratio
    moved = got != p0
    assert np.allclose(np.abs(got[moved] - p[moved]), thr, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("n,offset", [(5, 0), (4099, 1), (1_000_003, 0)])
def test_ewc_penalty_grad_vs_oracle(oracle_mod, n, offset):
    from unlearn_saliency_amd import ops
    star = rng.normal(n, 11, 0.0, 0.05)
    p = (star + rng.normal(n, 12, 0.0, 0.01)).astype(np.float32)
    F = np.abs(rng.normal(n, 13, 0.0, 1.0)).astype(np.float32)
    g0 = rng.normal(n, 14, 0.0, 1e-2)
    dev = lambda a: (lambda t: (t.copy_(torch.from_numpy(a)), t)[1])(torch.zeros(n + offset, device="cuda")[offset:])
    gd = dev(g0)
    out = ops.ewc_penalty_grad(dev(p), dev(star), dev(F), gd, 10.0)
    ref = g0.copy()
    loss, s = oracle_mod.ewc_penalty_grad(p, star, F, ref, 10.0)
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    o = out.cpu().numpy()
    assert abs(o[0] - loss) <= 2e-6 * abs(loss) and abs(o[1] - s) <= 2e-6 * abs(s)


def test_ewc_term_matches_reference_autograd(golden_dir):
    from unlearn_saliency_amd import ops
    g = np.load(os.path.join(golden_dir, "ewc_term.npz"))
    p, star, F = (torch.from_numpy(a).cuda() for a in ewc_inputs())
    grad = torch.zeros_like(p)
    out = ops.ewc_penalty_grad(p, star, F, grad, float(g["lam"]))
    assert np.array_equal(grad.cpu().numpy().view(np.uint32), g["grad"].view(np.uint32))
    assert abs(out[0].item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))


# ------------------------------------------------------------------------------------ plugins
@pytest.mark.parametrize("tag,use_mask", [("masked", True), ("unmasked", False)])
def test_boundary_shrink_plugin_matches_reference(golden_dir, tag, use_mask):
    from unlearn_saliency_amd.Classification import unlearn
    g = np.load(os.path.join(golden_dir, f"boundary_shrink_{tag}.npz"))
    model = TinyCNN()
    init = tiny_state(21)
    model.load_state_dict(init)
    model.cuda()
    mask = None
    if use_mask:
        sizes = [p.numel() for p in model.parameters()]
        off = np.cumsum([0] + sizes)
        mask = {n: torch.from_numpy(g["mask"][off[i]:off[i + 1]].astype(np.int64)).view_as(p)
                for i, (n, p) in enumerate(model.named_parameters())}
    unlearn.get_unlearn_method("boundary_shrink")({"forget": _loader(tiny_batches(2, 16, 700))}, model,
                                                  nn.CrossEntropyLoss(), _args(unlearn="boundary_shrink"), mask)
    _check_state(model, g)
    if use_mask:
        names = [n for n, _ in model.named_parameters()]
        sd = model.state_dict()
        now = np.concatenate([sd[n].reshape(-1).cpu().numpy() for n in names])
        was = np.concatenate([init[n].reshape(-1).numpy() for n in names])
        frozen = g["mask"] == 0
        assert np.array_equal(now[frozen].view(np.uint32), was[frozen].view(np.uint32))


def test_boundary_expanding_plugin_matches_reference(golden_dir):
    from unlearn_saliency_amd.Classification import unlearn
    g = np.load(os.path.join(golden_dir, "boundary_expanding.npz"))
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.cuda()
    torch.manual_seed(int(g["init_seed"]))
    unlearn.get_unlearn_method("boundary_expanding")({"forget": _loader(tiny_batches(2, 16, 700))}, model,
                                                     nn.CrossEntropyLoss(), _args(unlearn="boundary_expanding"), None)
    assert model.fc.out_features == 11 and model.fc.weight.is_cuda
    _check_state(model, g)
    with pytest.raises(RuntimeError):  # a pre-expansion mask cannot apply (the reference fails on the shape too)
        m2 = TinyCNN().cuda()
        unlearn.get_unlearn_method("boundary_expanding")(
            {"forget": _loader(tiny_batches(1, 16, 700))}, m2, nn.CrossEntropyLoss(), _args(),
            {n: torch.ones_like(p, dtype=torch.int64) for n, p in m2.named_parameters()})


def test_rl_proximal_plugin_matches_reference(golden_dir):
    from unlearn_saliency_amd.Classification import unlearn
    from unlearn_saliency_amd.Classification.dataset import BatchLoader
    g = np.load(os.path.join(golden_dir, "rl_proximal.npz"))
    fds, rds = next_rows_datasets()
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.cuda()
    loaders = {"forget": BatchLoader(fds, 16, False), "retain": BatchLoader(rds, 16, False)}
    np.random.seed(int(g["seed"]))
    torch.manual_seed(int(g["seed"]))
    unlearn.get_unlearn_method("RL_proximal")(loaders, model, nn.CrossEntropyLoss(),
                                              _args(unlearn="RL_proximal", mask_ratio=float(g["mask_ratio"])),
                                              {"ignored": None})
    _check_state(model, g, rtol=1e-5, atol=2e-7)
    with pytest.raises(AttributeError):
        unlearn.get_unlearn_method("RL_proximal")(loaders, model, nn.CrossEntropyLoss(), _args(), None)


# ------------------------------------------------------------------------------------ GA / GA_l1 / FT / FT_l1 plugins
@pytest.mark.parametrize("name,tag", [("GA", "masked"), ("GA", "unmasked"), ("FT", "masked"), ("FT", "unmasked"),
                                      ("FT_l1", "masked"), ("FT_l1", "unmasked"), ("GA_l1", "unmasked")])
def test_ga_ft_plugins_match_reference(golden_dir, name, tag):
    """Registry plugins on the device (fused masked-SGD step, l1 term through autograd on the main stream) against the
    state_dict produced by the reference's own GA / FT / FT_l1 / GA_l1 functions (tests/golden/make_golden_next.py)."""
    from unlearn_saliency_amd.Classification import unlearn
    g = np.load(os.path.join(golden_dir, f"{name.lower()}_{tag}.npz"))
    model = TinyCNN()
    init = tiny_state(21)
    model.load_state_dict(init)
    model.cuda()
    mask = None
    if g["mask"].size:
        sizes = [p.numel() for p in model.parameters()]
        off = np.cumsum([0] + sizes)
        mask = {n: torch.from_numpy(g["mask"][off[i]:off[i + 1]].astype(np.int64)).view_as(p)
                for i, (n, p) in enumerate(model.named_parameters())}
    key, batches = ("forget", tiny_batches(2, 16, 700)) if name.startswith("GA") else ("retain", tiny_batches(3, 16, 800))
    args = _args(unlearn=name, alpha=float(g["alpha"]), no_l1_epochs=int(g["no_l1_epochs"]) if "no_l1_epochs" in g else 0)
    unlearn.get_unlearn_method(name)({key: _loader(batches)}, model, nn.CrossEntropyLoss(), args, mask)
    _check_state(model, g, rtol=1e-5, atol=2e-7)
    if mask is not None:
        names = [n for n, _ in model.named_parameters()]
        sd = model.state_dict()
        now = np.concatenate([sd[n].reshape(-1).cpu().numpy() for n in names])
        was = np.concatenate([init[n].reshape(-1).numpy() for n in names])
        frozen = g["mask"] == 0
        assert np.array_equal(now[frozen].view(np.uint32), was[frozen].view(np.uint32))


def test_validate_and_mia_on_the_device_match_the_reference(golden_dir):
    """A7 on the GPU: `trainer.validate` (top-1, sample-weighted) and `SVC_MIA` on device-resident loaders against the
    numbers the reference's own functions returned for the same model and data (eval_tinycnn.npz)."""
    import importlib
    cpu_test = importlib.import_module("test_eval_vs_golden")
    assert hasattr(cpu_test, "run_eval"), "tests/test_eval_vs_golden.py exposes run_eval(device)"
    cpu_test.run_eval(golden_dir, "cuda")


def test_a_failed_select_poisons_the_weights_instead_of_resetting_them():
    """when the select fails (the full scan's grid barrier timed out), `salun_mask_topk_thresholds` exports
    NaN and `salun_soft_threshold_step` must spread it — a plain comparison against NaN would silently write p0
    everywhere (every weight reset), the worst possible quiet outcome."""
    import ctypes
    from unlearn_saliency_amd import _lib, ops
    n = 10_000
    p0 = torch.randn(n, device="cuda")
    p = p0 + 0.01 * torch.randn(n, device="cuda")
    tau = torch.full((1,), float("nan"), device="cuda")
    ops.check(_lib.lib().salun_soft_threshold_step(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(p0.data_ptr()),
                                                   ctypes.c_void_p(tau.data_ptr()), ctypes.c_int64(n), ops._stream()),
              "salun_soft_threshold_step")
    assert bool(torch.isnan(p).all())
    # and the healthy path: a finite threshold behaves as before
    p = p0 + 0.01 * torch.randn(n, device="cuda")
    t = ops.proximal_step(p, p0, n // 2)
    assert bool(torch.isfinite(p).all()) and float(t) > 0 and int((p == p0).sum()) >= n // 2


def test_mask_topk_check_reports_a_healthy_select():
    from unlearn_saliency_amd import ops
    acc = torch.randn(50_000, device="cuda")
    m = ops.mask_topk(acc, [25_000], check=True)[0]                      # single-read route
    assert int(m.sum()) == 25_000 and ops.mask_topk_status(acc.device) == (1, 0)
    m = ops.mask_topk(acc[:5000], [2500], check=True)[0]                 # n < 8192: the (cooperative) full scan
    assert int(m.sum()) == 2500 and ops.mask_topk_status(acc.device) == (2, 0)
