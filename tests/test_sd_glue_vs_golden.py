"""A13 / A14 / F2 pinned: the oracle restatements of the Stable-Diffusion scripts (`oracle/torch_ref.py`:
`sd_saliency_gradients`, `masks_from_gradients_cpu`, `sd_unlearn`, `sd_proximal_unlearn` on `PlainLDM`) against
`tests/golden/sd_glue.npz` — outputs of the REFERENCE's `generate_nsfw_mask`, `generate_mask`, `nsfw_removal`,
`certain_label` and `proximal_gradient` executed by tests/golden/make_golden_sd_glue.py on the reference's `UNetModel`,
with every random draw recorded.  The device path is checked against the same file in tests/test_sd_parity_gpu.py.

Tolerances (fp32 on both sides, the same library ops, different module code): losses 1e-5 relative (north_star);
accumulators / Adam moments 1e-5 of the vector's scale; masks: bit-exact on the reference's own accumulator."""
import os

import numpy as np
import pytest
import torch

from fixtures import fill_params, replay_draws, sd_glue_config, sd_glue_loaders

STRIDE = 7


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "sd_glue.npz"))


def _ldm():
    from oracle import torch_ref
    from unlearn_saliency_amd.SD.unet import UNetModel
    return torch_ref.PlainLDM(fill_params(UNetModel(**sd_glue_config()), 9100))


def _flat(unet):
    return torch.cat([p.detach().reshape(-1) for p in unet.parameters()]).numpy()


def _moments(opt, unet):
    n = sum(p.numel() for p in unet.parameters())
    m1, m2, off = np.zeros(n, np.float32), np.zeros(n, np.float32), 0
    for p in unet.parameters():
        st = opt.state.get(p)
        if st:
            m1[off:off + p.numel()] = st["exp_avg"].reshape(-1).numpy()
            m2[off:off + p.numel()] = st["exp_avg_sq"].reshape(-1).numpy()
        off += p.numel()
    return m1, m2


def _close(a, b, rel, what):
    scale = float(np.abs(b).max())
    err = float(np.abs(a - b).max()) / scale
    assert err <= rel, f"{what}: {err:.2e} of scale"
    return err


def test_fixture_model_is_the_reference_model(g):
    ldm = _ldm()
    assert [n for n, _ in ldm.unet.named_parameters()] == list(g["param_names"])
    assert abs(float(_flat(ldm.unet).astype(np.float64).sum()) - float(g["init_sum"])) < 1e-9


@pytest.mark.parametrize("tag,kind", [("nsfw_mask", "nsfw_mask"), ("class_mask", "class_mask")])
def test_saliency_accumulator_and_mask_equal_the_reference_run(g, oracle_mod, tag, kind):
    from oracle import torch_ref
    ldm = _ldm()
    ri, rn = g[f"{tag}__randint"], g[f"{tag}__randn"]
    if tag == "nsfw_mask":
        assert len(ri) == 2 * len(rn)
        ri = ri[1::2]  # generate_nsfw_mask draws an unused t first (generate_mask.py:141-143): only the second is used
    with replay_draws(ri, rn):
        grads = torch_ref.sd_saliency_gradients(ldm, sd_glue_loaders(kind), 7.5)
    acc = np.abs(np.concatenate([v.reshape(-1).numpy() for v in grads.values()]))
    n = acc.size
    bits = np.unpackbits(g[f"{tag}__mask_bits"])[:n]
    assert int(bits.sum()) == int(n * 0.5)
    if tag == "nsfw_mask":
        ref_acc = g["nsfw_mask__abs_acc"]
        _close(acc, ref_acc, 1e-5, "accumulator")
        # the oracle's ranking on the REFERENCE's accumulator reproduces the reference's mask bit for bit
        assert bool(g["nsfw_mask__tau_unique"])
        m = oracle_mod.mask_topk(ref_acc, [int(n * 0.5)])[0]
        assert np.array_equal(m, bits)
        d = {f"t{i}": torch.from_numpy(ref_acc[i::3].copy()) for i in range(3)}  # the torch restatement as well
        mm = torch_ref.masks_from_gradients_cpu(d, [0.5])[0.5]
        want = np.concatenate([bits[i::3] for i in range(3)])
        assert np.array_equal(torch.cat([v.reshape(-1) for v in mm.values()]).numpy().astype(np.uint8), want)
    else:
        _close(acc[::STRIDE], g["class_mask__abs_acc_s"], 1e-5, "accumulator sample")
        assert abs(float(acc.astype(np.float64).sum()) - float(g["class_mask__abs_acc_sum"])) <= 1e-5 * float(g["class_mask__abs_acc_sum"])
    # own accumulator -> own mask: differs from the reference's only where two saliencies are within round-off
    own = oracle_mod.mask_topk(acc.astype(np.float32), [int(n * 0.5)])[0]
    assert float((own != bits).mean()) < 1e-3


@pytest.mark.parametrize("method", ["full", "xattn"])
def test_nsfw_removal_equals_the_reference_run(g, method):
    from oracle import torch_ref
    tag = f"nsfw_removal_{method}"
    ldm = _ldm()
    n = sum(p.numel() for p in ldm.unet.parameters())
    bits = np.unpackbits(g["nsfw_mask__mask_bits"])[:n].astype(np.int64)
    off, mask = 0, {}
    for name, p in ldm.unet.named_parameters():
        mask[name] = torch.from_numpy(bits[off:off + p.numel()]).view_as(p)
        off += p.numel()
    forget, remain = sd_glue_loaders("nsfw")
    init = _flat(ldm.unet).copy()
    with replay_draws(g[f"{tag}__randint"], g[f"{tag}__randn"]):
        losses, opt = torch_ref.sd_unlearn(ldm, forget, remain, 0.5, 1e-4, mask, method)
    ref = g[f"{tag}__losses"]
    assert np.abs(np.array(losses) - ref).max() <= 1e-5 * np.abs(ref).max(), (losses, ref)
    m1, m2 = _moments(opt, ldm.unet)
    _close(m1[::STRIDE], g[f"{tag}__exp_avg_s"], 1e-5, "exp_avg")
    _close(m2[::STRIDE], g[f"{tag}__exp_avg_sq_s"], 1e-5, "exp_avg_sq")
    w = _flat(ldm.unet)
    assert np.array_equal(w[bits == 0], init[bits == 0])  # masked-out weights never move
    # Adam's first steps move a weight by ~lr * sign(g): compare the MOVEMENT where the reference moved by more than
    # round-off, at 1e-3 of the step size (a gradient within 1e-6 of zero may flip a whole lr; those are listed)
    dw, dref = (w - init)[::STRIDE], g[f"{tag}__weights_s"] - init[::STRIDE]
    bad = np.abs(dw - dref) > 1e-3 * 3e-4 + 1e-3 * np.abs(dref)
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size} sampled weights moved differently"


def test_certain_label_equals_the_reference_run(g):
    from oracle import torch_ref
    ldm = _ldm()
    forget, remain = sd_glue_loaders("class", g["certain_label__remain_labels"])
    with replay_draws(g["certain_label__randint"], g["certain_label__randn"]):
        losses, opt = torch_ref.sd_unlearn(ldm, forget, remain, 0.5, 1e-4, None, "full", epochs=2)
    ref = g["certain_label__losses"]
    assert len(losses) == 6 and np.abs(np.array(losses) - ref).max() <= 1e-5 * np.abs(ref).max(), (losses, ref)
    m1, m2 = _moments(opt, ldm.unet)
    _close(m1[::STRIDE], g["certain_label__exp_avg_s"], 2e-5, "exp_avg")
    _close(m2[::STRIDE], g["certain_label__exp_avg_sq_s"], 2e-5, "exp_avg_sq")


def test_proximal_gradient_equals_the_reference_run_with_frozen_stages_counted(g):
    """the reference ranks over U-Net + frozen stages; with the stages ignored the thresholds differ and
    so does every later loss."""
    from oracle import torch_ref
    n_unet, n_all = int(g["proximal__n_unet"]), int(g["proximal__n_all"])
    assert "classes" in str(g["proximal__ended_with"])  # the reference dies after training on an undefined name (:200)
    forget, remain = sd_glue_loaders("class", g["certain_label__remain_labels"])
    ldm = _ldm()
    init = _flat(ldm.unet).copy()
    with replay_draws(g["proximal__randint"], g["proximal__randn"]):
        losses, _ = torch_ref.sd_proximal_unlearn(ldm, forget, remain, 0.5, 1e-4, float(g["proximal__mask_ratio"]),
                                                  n_frozen=n_all - n_unet, epochs=2)
    ref = g["proximal__losses"]
    assert np.abs(np.array(losses) - ref).max() <= 1e-5 * np.abs(ref).max(), (losses, ref)
    w = _flat(ldm.unet)
    # weights that sit exactly on theta_0 at the end: the time-embedding weights whose gradients are so small
    # (|g| << Adam's eps) that the update is below half an ulp — a round-off-level set, compared as a count (10 %)
    assert abs(int((w == init).sum()) - int(g["proximal__reset_count"])) <= 0.1 * int(g["proximal__reset_count"])
    dw, dref = (w - init)[::STRIDE], g["proximal__weights_s"] - init[::STRIDE]
    bad = np.abs(dw - dref) > 1e-3 * 6e-4 + 1e-3 * np.abs(dref)
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size}"
    # and the count matters: ranking over the U-Net alone pulls with larger thresholds at every step (the losses hardly
    # notice — Adam moves a weight by ~lr per step — but the weights do): far more weights end exactly on theta_0
    ldm2 = _ldm()
    with replay_draws(g["proximal__randint"], g["proximal__randn"]):
        torch_ref.sd_proximal_unlearn(ldm2, forget, remain, 0.5, 1e-4, float(g["proximal__mask_ratio"]), n_frozen=0,
                                      epochs=2)
    assert int((_flat(ldm2.unet) == init).sum()) > 2 * int(g["proximal__reset_count"])
