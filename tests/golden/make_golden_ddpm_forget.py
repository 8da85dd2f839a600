"""Golden vectors for the EWC / Selective-Amnesia loop: runs the REFERENCE's
`Diffusion.train_forget()` (DDPM/runners/diffusion.py:273-396, imported from /root/reference/DDPM, build container
only) on the reduced U-Net for 3 iterations with a generated Fisher dictionary, recording every random draw.

    python tests/golden/make_golden_ddpm_forget.py
"""
from __future__ import annotations

import os
import pickle
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_ddpm as MD  # noqa: E402
from fixtures import ddpm_batch, ddpm_small_config, fill_params, fisher_fixture  # noqa: E402


def main():
    RD, RLoss, RM = MD.import_reference_ddpm()
    cfg = ddpm_small_config()
    cfg.training.n_iters = 3
    cfg.training.gamma, cfg.training.lmbda = 1, 10
    remain = MD.Loader([tuple(map(torch.from_numpy, ddpm_batch(4, 300 + i))) for i in range(2)])
    rec = dict(randn=[], randint=[], rand=[])
    real = (torch.randn_like, torch.randint, torch.rand, RD.get_optimizer)
    cap = {}

    def randn_like(x, **k):
        r = real[0](x, **k)
        rec["randn"].append(r.clone())
        return r

    def randint(*a, **k):
        r = real[1](*a, **k)
        rec["randint"].append(r.clone())
        return r

    def rand(*a, **k):
        r = real[2](*a, **k)
        rec["rand"].append(r.clone())
        return r

    def get_opt(config, params):
        params = list(params)
        cap["params"] = params
        return real[3](config, params)

    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "ckpts"))
        init = fill_params(RM.Conditional_Model(cfg), 7000)
        dp = torch.nn.DataParallel(init)
        torch.save([dp.state_dict(), None, 0], os.path.join(d, "ckpts/ckpt.pth"))
        names = [n for n, _ in dp.named_parameters()]
        shapes = [tuple(p.shape) for p in dp.parameters()]
        F = fisher_fixture(shapes)
        with open(os.path.join(d, "fisher_dict.pkl"), "wb") as f:
            pickle.dump({n: torch.from_numpy(a) for n, a in zip(names, F)}, f)
        cfg.ckpt_dir, cfg.log_dir = os.path.join(d, "out_ckpts"), os.path.join(d, "logs")
        os.makedirs(cfg.ckpt_dir)
        args = SimpleNamespace(ckpt_folder=d, label_to_forget=0, cond_scale=2.0, mask_path=None)
        RD.all_but_one_class_path_dataset = lambda c, p, l: remain
        RD.get_optimizer = get_opt
        torch.randn_like, torch.randint, torch.rand = randn_like, randint, rand
        try:
            torch.manual_seed(99)
            RD.Diffusion(args, cfg).train_forget()
        finally:
            torch.randn_like, torch.randint, torch.rand, RD.get_optimizer = real
    s = MD.summarize(SimpleNamespace(parameters=lambda: cap["params"]))
    np.savez_compressed(os.path.join(HERE, "ddpm_train_forget.npz"), param_sample=s["sample"],
                        tensor_sums=s["tensor_sums"], randn=np.stack([t.numpy() for t in rec["randn"]]),
                        randint=np.stack([t.numpy() for t in rec["randint"]]),
                        rand=np.stack([t.numpy() for t in rec["rand"]]), n_iters=3, gamma=1, lmbda=10)
    print("ddpm_train_forget.npz written:", len(rec["randn"]), "randn,", len(rec["rand"]), "rand draws")


if __name__ == "__main__":
    main()
