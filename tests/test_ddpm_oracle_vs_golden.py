"""DDPM: oracle / CPU restatement vs golden vectors captured from the reference's DDPM code
(tests/golden/make_golden_ddpm.py).  CPU only; also pins the package's U-Net architecture (it is plain
PyTorch and runs on CPU) against the reference's forward output."""
import os

import numpy as np
import pytest
import torch

import ddpm_ref_cpu as R
from fixtures import ddpm_batch, ddpm_small_config, fill_params, flat_params

STRIDE = 997


@pytest.fixture(scope="module")
def core(golden_dir):
    return np.load(os.path.join(golden_dir, "ddpm_core.npz"))


def t_(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_beta_schedules_and_tables(core):
    from unlearn_saliency_amd.DDPM.runners.diffusion import get_beta_schedule
    for s in ("linear", "quad", "sigmoid"):
        b = get_beta_schedule(s, beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000)
        assert b.dtype == np.float64 and np.array_equal(b, core[f"betas_{s}"])
    a = (1 - t_(core["betas_linear"]).float()).cumprod(dim=0)
    assert np.array_equal(a.numpy(), core["alphas_cumprod"])
    assert np.array_equal(a.sqrt().numpy(), core["sqrt_ab"])


def test_qsample_bit_exact_vs_reference(oracle_mod, core):
    xt = oracle_mod.qsample(core["loss_x0"], core["loss_e"], core["sqrt_ab"], core["sqrt_1mab"], core["loss_t"])
    assert np.array_equal(xt.view(np.uint32), core["loss_xt"].view(np.uint32))


def test_eps_mse_value_and_gradient(oracle_mod, core):
    e, out = core["loss_e"], core["loss_out"]
    B = e.shape[0]
    loss, per, d = oracle_mod.sqerr_loss(e, out, 1.0 / B)
    assert abs(loss - float(core["loss_value"])) <= 1e-6 * abs(float(core["loss_value"]))
    assert np.allclose(per, core["loss_per_sample"], rtol=1e-6, atol=0)
    assert np.allclose(d, core["loss_dout"], rtol=1e-6, atol=1e-12)
    loss2, _, d2 = oracle_mod.sqerr_loss(core["mse_pseudo"], out, 1.0 / out.size)
    assert abs(loss2 - float(core["mse_value"])) <= 1e-6 * abs(float(core["mse_value"]))
    assert np.allclose(d2, core["mse_dout"], rtol=1e-6, atol=1e-12)


def test_parameter_tables_match_reference(core):
    """Names / shapes / order of the full CFG-DDPM U-Net = the mask keys and the flat ranking order."""
    import yaml
    from unlearn_saliency_amd.DDPM.functions import load_config
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = load_config(os.path.join(here, "..", "unlearn_saliency_amd", "DDPM", "configs", "cifar10_saliency_unlearn.yml"))
    m = Conditional_Model(cfg)
    assert [n for n, _ in m.named_parameters()] == list(core["full_param_names"])
    assert [str(tuple(p.shape)) for p in m.parameters()] == list(core["full_param_shapes"])
    assert sum(p.numel() for p in m.parameters()) == 38_632_323 and len(list(m.parameters())) == 334
    small = Conditional_Model(ddpm_small_config())
    assert [n for n, _ in small.named_parameters()] == list(core["small_param_names"])
    assert [p.numel() for p in small.parameters()] == list(core["small_param_numel"])


def test_unet_forward_matches_reference(core):
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    model = fill_params(Conditional_Model(ddpm_small_config()), 7000).eval()
    xb, cb = ddpm_batch(4, 200)
    xb = t_(2 * xb - 1)
    tb = torch.tensor([5.0, 400.0, 750.0, 999.0])
    with torch.no_grad():
        for key, kw in (("fwd_test_s2", dict(mode="test", cond_scale=2.0)),
                        ("fwd_train_nodrop", dict(mode="train", cond_drop_prob=0.0)),
                        ("fwd_train_alldrop", dict(mode="train", cond_drop_prob=1.0))):
            out = model(xb, tb, t_(cb), **kw).numpy()
            ref = core[key]
            assert np.allclose(out, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max()), key


def _batches(seed0, label=None):
    return [tuple(map(t_, ddpm_batch(4, seed0 + i, label=label))) for i in range(2)]


def test_generate_mask_restatement_vs_reference(oracle_mod, golden_dir):
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    g = np.load(os.path.join(golden_dir, "ddpm_generate_mask.npz"))
    cfg = ddpm_small_config()
    model = fill_params(Conditional_Model(cfg), 7000)
    with R.replay(randn=g["randn"], randint=g["randint"]):
        acc, mask = R.cpu_generate_mask(cfg, model, _batches(400, label=0))
    n = int(g["n"])
    assert acc.size == n
    ref_norm = float(g["acc_norm"])
    assert abs(np.linalg.norm(acc.astype(np.float64)) - ref_norm) <= 1e-5 * ref_norm
    assert np.allclose(acc[::STRIDE], g["acc_sample"], rtol=1e-4, atol=1e-5 * np.abs(g["acc_sample"]).max())
    ref_mask = np.unpackbits(g["mask_packed"])[:n]
    assert int(mask.sum()) == int(g["popcount"]) == oracle_mod.k_of(n, 0.5)
    # the two accumulators differ by float rounding, so only saliencies within rounding of the threshold may flip
    assert (mask != ref_mask).mean() < 1e-3
    assert [k for k in g["mask_keys"]][:2] == ["module.null_classes_emb", "module.temb.dense.0.weight"]


@pytest.mark.parametrize("method", ["rl", "ga"])
def test_saliency_unlearn_restatement_vs_reference(oracle_mod, golden_dir, method):
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    g = np.load(os.path.join(golden_dir, f"ddpm_unlearn_{method}.npz"))
    gm = np.load(os.path.join(golden_dir, "ddpm_generate_mask.npz"))
    n = int(gm["n"])
    mask = np.unpackbits(gm["mask_packed"])[:n].astype(np.uint8)
    cfg = ddpm_small_config()
    model = fill_params(Conditional_Model(cfg), 7000)
    before = flat_params(model)
    with R.replay(randn=g["randn"], randint=g["randint"], keep=g["keep"]):
        losses = R.cpu_unlearn(cfg, model, method, 1e-3, _batches(300), _batches(400, label=0), mask, n_iters=2)
    # the loop's loss scalar at every step (forget term + alpha * eps-MSE): 1e-5 relative (north_star)
    assert np.allclose(losses, g["step_loss"], rtol=1e-5, atol=0), (losses, g["step_loss"])
    # Adam's moments are linear / quadratic in the clipped, masked gradients: the per-weight quantities that can be
    # compared at fp32 round-off (the weights themselves move by ~lr * sign(g) in the first steps)
    m1, v = R.cpu_unlearn.last_moments
    s1, s2 = np.abs(g["exp_avg_sample"]).max(), np.abs(g["exp_avg_sq_sample"]).max()
    assert np.allclose(m1[::STRIDE], g["exp_avg_sample"], rtol=1e-4, atol=1e-5 * s1)
    assert np.allclose(v[::STRIDE], g["exp_avg_sq_sample"], rtol=2e-4, atol=1e-5 * s2)
    assert abs(np.linalg.norm(m1.astype(np.float64)) - float(g["exp_avg_norm"])) <= 1e-5 * float(g["exp_avg_norm"])
    assert abs(v.astype(np.float64).sum() - float(g["exp_avg_sq_sum"])) <= 2e-5 * float(g["exp_avg_sq_sum"])
    assert not m1[mask == 0].any() and not v[mask == 0].any()
    after = flat_params(model)
    # masked-out weights are bit-identical to the start (Adam state stays 0 there)
    assert np.array_equal(after[mask == 0].view(np.uint32), before[mask == 0].view(np.uint32))
    ref = g["param_sample"]
    got = after[::STRIDE]
    lr = cfg.optim.lr
    # Adam's first steps move each selected weight by ~lr*sign(g): compare the *update*, tolerance 2 % of lr
    close = np.abs(got - ref) <= 0.02 * lr + 1e-6 * np.abs(ref)
    assert close.mean() > 0.995, close.mean()
    assert np.abs(got - ref).max() <= 2.5 * lr * 2
    sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
    assert np.allclose(sums, g["tensor_sums"], rtol=1e-4, atol=2e-3)


def test_train_forget_restatement_vs_reference(oracle_mod, golden_dir):
    """EWC / Selective-Amnesia loop: 3 iterations, the fused EWC term entering before the clip."""
    from fixtures import fisher_fixture
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    g = np.load(os.path.join(golden_dir, "ddpm_train_forget.npz"))
    cfg = ddpm_small_config()
    cfg.training.gamma, cfg.training.lmbda = int(g["gamma"]), int(g["lmbda"])
    model = fill_params(Conditional_Model(cfg), 7000)
    F = np.concatenate([a.reshape(-1) for a in fisher_fixture([tuple(p.shape) for p in model.parameters()])])
    with R.replay(randn=g["randn"], randint=g["randint"], rand=g["rand"]):
        losses = R.cpu_train_forget(cfg, model, _batches(300), F, n_iters=int(g["n_iters"]))
    assert losses[0][1] == 0.0 and losses[2][1] > 0.0  # theta == theta* at the first step, then the anchor pulls
    lr = cfg.optim.lr
    got, ref = flat_params(model)[::STRIDE], g["param_sample"]
    close = np.abs(got - ref) <= 0.02 * lr + 1e-6 * np.abs(ref)
    assert close.mean() > 0.995, close.mean()
    assert np.abs(got - ref).max() <= 3 * lr * 2
    sums = np.array([float(p.detach().double().sum()) for p in model.parameters()])
    assert np.allclose(sums, g["tensor_sums"], rtol=1e-4, atol=2e-3)


def test_fim_restatement_vs_reference(oracle_mod, golden_dir):
    from unlearn_saliency_amd.DDPM.models.diffusion import Conditional_Model
    g = np.load(os.path.join(golden_dir, "ddpm_fim.npz"))
    cfg = ddpm_small_config(T=4)
    model = fill_params(Conditional_Model(cfg), 7000)
    samples = [tuple(map(t_, ddpm_batch(1, 500 + i))) for i in range(2)]
    with R.replay(randn=g["randn"], keep=g["keep"]):
        F = R.cpu_fim(cfg, model, samples, n_chunks=2)
    assert abs(F.astype(np.float64).sum() - float(g["F_sum"])) <= 1e-4 * float(g["F_sum"])
    assert np.allclose(F[::STRIDE], g["F_sample"], rtol=1e-3, atol=1e-5 * np.abs(g["F_sample"]).max())
    assert list(g["keys"])[0] == "module.null_classes_emb"
