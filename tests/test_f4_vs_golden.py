"""Evaluation-side rows: the DDPM reverse-process samplers and the EMA helper against outputs of the
reference's own functions (tests/golden/make_golden_f4.py; DDPM/functions/denoising.py:11-131, DDPM/models/ema.py:5-51).
Device agnostic code: checked on the CPU here."""
import os

import numpy as np
import pytest
import torch

from unlearn_saliency_amd import rng


class StubEps(torch.nn.Module):
    """The deterministic eps model of the golden generator."""

    def forward(self, x, t, c=None, cond_scale=None, mode=None):
        g = 0.3 + 0.0005 * t.view(-1, 1, 1, 1).float()
        out = torch.tanh(x.flip(1)) * g + 0.1 * x
        if c is not None:
            out = out + 0.01 * c.view(-1, 1, 1, 1).float() * (1.0 if cond_scale is None else cond_scale)
        return out


class _ReplayRandn:
    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self.real = torch.randn_like
        torch.randn_like = lambda x, **k: torch.as_tensor(self.draws.pop(0)).to(x.device).reshape(x.shape)
        return self

    def __exit__(self, *a):
        torch.randn_like = self.real
        assert not self.draws, "recorded draws left over: the call order differs"


@pytest.mark.parametrize("name", ["ddim", "ddim_eta", "ddpm", "ddim_cond", "ddpm_cond"])
@pytest.mark.parametrize("keep", ["all", "last"])
def test_samplers_match_reference(golden_dir, name, keep):
    from unlearn_saliency_amd.DDPM.functions import denoising as DN
    g = np.load(os.path.join(golden_dir, "ddpm_f4.npz"))
    x = torch.from_numpy(g["x"])
    seq = [int(v) for v in g["seq"]]
    betas = torch.linspace(1e-4, 0.02, 1000)
    c = torch.tensor([1, 5, 9])
    model = StubEps()
    call = {"ddim": lambda: DN.generalized_steps(x, seq, model, betas, eta=0.0, keep=keep),
            "ddim_eta": lambda: DN.generalized_steps(x, seq, model, betas, eta=0.7, keep=keep),
            "ddpm": lambda: DN.ddpm_steps(x, seq, model, betas, keep=keep),
            "ddim_cond": lambda: DN.generalized_steps_conditional(x, c, seq, model, betas, cond_scale=2.0, eta=0.3, keep=keep),
            "ddpm_cond": lambda: DN.ddpm_step_conditional(x, c, seq, model, betas, 2.0, keep=keep)}[name]
    with _ReplayRandn(g[name + "_randn"]):
        xs, x0s = call()
    ref_xs, ref_x0 = g[name + "_xs"], g[name + "_x0"]
    if keep == "all":
        assert len(xs) == len(ref_xs) == len(seq) + 1 and len(x0s) == len(ref_x0) == len(seq)
        for a, b in zip(xs, ref_xs):
            assert np.allclose(a.numpy(), b, rtol=1e-5, atol=1e-6)
        for a, b in zip(x0s, ref_x0):
            assert np.allclose(a.numpy(), b, rtol=1e-5, atol=1e-6)
    else:
        assert len(xs) == 1 and len(x0s) == 1
    assert np.allclose(xs[-1].numpy(), ref_xs[-1], rtol=1e-5, atol=1e-6)  # what sample_image returns
    assert np.allclose(x0s[-1].numpy(), ref_x0[-1], rtol=1e-5, atol=1e-6)


def test_compute_alpha_matches_its_definition():
    from unlearn_saliency_amd.DDPM.functions.denoising import compute_alpha
    betas = torch.linspace(1e-4, 0.02, 1000)
    t = torch.tensor([-1, 0, 17, 999])
    a = compute_alpha(betas, t).view(-1)
    cp = (1 - betas).cumprod(0)
    assert a[0] == 1.0 and torch.allclose(a[1:], cp[[0, 17, 999]], rtol=1e-6)


def _ema_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))


@pytest.mark.parametrize("flat", [False, True])
def test_ema_matches_reference(golden_dir, flat):
    """Per-tensor route and the flat-arena route (one lerp over the whole vector) against the reference's EMAHelper."""
    from unlearn_saliency_amd.DDPM.models.ema import EMAHelper
    from unlearn_saliency_amd.flat import arena_of
    g = np.load(os.path.join(golden_dir, "ddpm_f4.npz"))
    lin = _ema_model()
    if flat:
        arena_of(lin)
    ema = EMAHelper(mu=0.9)
    ema.register(lin)
    assert list(ema.state_dict().keys()) == list(g["ema_keys"])
    for step in range(4):
        with torch.no_grad():
            for i, p in enumerate(lin.parameters()):
                p.add_(torch.from_numpy(rng.normal(p.numel(), 4100 + 10 * step + i, 0.0, 0.1)).view_as(p))
        ema.update(lin)
        got = np.concatenate([v.reshape(-1).numpy() for v in ema.state_dict().values()])
        assert np.allclose(got, g["ema_states"][step], rtol=1e-6, atol=1e-7), step
    # ema(): the shadow is copied into a module; state_dict round trip
    target = _ema_model()
    ema2 = EMAHelper(mu=0.9)
    ema2.register(target)
    ema2.load_state_dict({k: v.clone() for k, v in ema.state_dict().items()})
    ema2.ema(target)
    now = np.concatenate([p.detach().reshape(-1).numpy() for p in target.parameters()])
    assert np.allclose(now, g["ema_states"][-1], rtol=1e-6, atol=1e-7)


def test_runner_sample_image_picks_the_sampler_and_sequence():
    """Diffusion.sample_image (reference runners/diffusion.py:828-875): uniform / quad timestep sequences, DDIM with
    eta or ancestral sampling, final state only."""
    from types import SimpleNamespace
    from unlearn_saliency_amd.DDPM.functions import denoising as DN
    from unlearn_saliency_amd.DDPM.runners.diffusion import Diffusion
    r = Diffusion.__new__(Diffusion)
    r.num_timesteps = 1000
    r.betas = torch.linspace(1e-4, 0.02, 1000)
    x = torch.from_numpy(rng.normal(2 * 3 * 8 * 8, 1).reshape(2, 3, 8, 8))
    c = torch.tensor([2, 7])
    model = StubEps()
    r.args = SimpleNamespace(sample_type="generalized", skip_type="uniform", timesteps=8, eta=0.0)
    out = r.sample_image(x, model, c, 2.0)
    ref, _ = DN.generalized_steps_conditional(x, c, range(0, 1000, 125), model, r.betas, 2.0, eta=0.0)
    assert torch.equal(out, ref[-1])
    r.args = SimpleNamespace(sample_type="ddpm_noisy", skip_type="quad", timesteps=6, eta=0.0)
    torch.manual_seed(0)
    out = r.sample_image(x, model, c, 1.5)
    seq = [int(s) for s in list(np.linspace(0, np.sqrt(800.0), 6) ** 2)]
    torch.manual_seed(0)
    ref, _ = DN.ddpm_step_conditional(x, c, seq, model, r.betas, 1.5)
    assert torch.equal(out, ref[-1])
