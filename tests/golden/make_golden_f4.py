"""Golden vectors for the evaluation-side rows, produced by the REFERENCE's own functions imported
from /root/reference/DDPM (build container only):

    functions/denoising.py  generalized_steps[_conditional], ddpm_steps, ddpm_step_conditional   (with a stub eps model)
    models/ema.py           EMAHelper over a few updates

    python tests/golden/make_golden_f4.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_ddpm import import_reference_ddpm  # noqa: E402
from unlearn_saliency_amd import rng  # noqa: E402


class StubEps(torch.nn.Module):
    """Deterministic eps(x, t[, c]): a fixed mixing of the input with timestep / class dependent gains."""

    def forward(self, x, t, c=None, cond_scale=None, mode=None):
        g = 0.3 + 0.0005 * t.view(-1, 1, 1, 1).float()
        out = torch.tanh(x.flip(1)) * g + 0.1 * x
        if c is not None:
            out = out + 0.01 * c.view(-1, 1, 1, 1).float() * (1.0 if cond_scale is None else cond_scale)
        return out


def main():
    import_reference_ddpm()
    import functions.denoising as RDN
    import models.ema as REMA
    betas = torch.linspace(1e-4, 0.02, 1000)
    x = torch.from_numpy(rng.normal(3 * 3 * 8 * 8, 4000).reshape(3, 3, 8, 8))
    c = torch.tensor([1, 5, 9])
    seq = list(range(0, 1000, 125))
    out = {"x": x.numpy(), "seq": np.array(seq)}
    model = StubEps()
    for name, call in (("ddim", lambda: RDN.generalized_steps(x, seq, model, betas, eta=0.0)),
                       ("ddim_eta", lambda: RDN.generalized_steps(x, seq, model, betas, eta=0.7)),
                       ("ddpm", lambda: RDN.ddpm_steps(x, seq, model, betas)),
                       ("ddim_cond", lambda: RDN.generalized_steps_conditional(x, c, seq, model, betas, cond_scale=2.0, eta=0.3)),
                       ("ddpm_cond", lambda: RDN.ddpm_step_conditional(x, c, seq, model, betas, 2.0))):
        rec = []
        real = torch.randn_like

        def randn_like(t_, **k):
            r = real(t_, **k)
            rec.append(r.clone())
            return r

        torch.manual_seed(11)
        torch.randn_like = randn_like
        try:
            xs, x0s = call()
        finally:
            torch.randn_like = real
        out[name + "_xs"] = np.stack([t_.numpy() for t_ in xs])
        out[name + "_x0"] = np.stack([t_.numpy() for t_ in x0s])
        out[name + "_randn"] = np.stack([t_.numpy() for t_ in rec])
    # EMA
    torch.manual_seed(3)
    lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 2))
    ema = REMA.EMAHelper(mu=0.9)
    ema.register(lin)
    states = []
    for step in range(4):
        with torch.no_grad():
            for i, p in enumerate(lin.parameters()):
                p.add_(torch.from_numpy(rng.normal(p.numel(), 4100 + 10 * step + i, 0.0, 0.1)).view_as(p))
        ema.update(lin)
        states.append(np.concatenate([v.reshape(-1).numpy() for v in ema.state_dict().values()]))
    out["ema_states"] = np.stack(states)
    out["ema_keys"] = np.array(list(ema.state_dict().keys()))
    out["ema_init"] = np.concatenate([p.detach().reshape(-1).numpy() for p in lin.parameters()])
    np.savez_compressed(os.path.join(HERE, "ddpm_f4.npz"), **out)
    print("f4 fixtures written")


if __name__ == "__main__":
    main()
