"""Golden vectors for the "next" rows, produced by calling the REFERENCE's own functions
(imported from /root/reference through the stubs of make_golden.py; build container only):

    boundary_shrink / boundary_expanding (Classification/unlearn/boundary_sh.py, boundary_ex.py)
    RL_proximal                          (Classification/unlearn/RL_pro.py)
    EWC term of train_forget             (DDPM/runners/diffusion.py:343-350, evaluated with autograd)

    python tests/golden/make_golden_next.py

Inputs come from the counter-based generator (seeds stored next to the outputs); fixtures are data only.
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (stubs + reference import; also puts the repo root on sys.path)
from fixtures import TinyCNN, next_rows_datasets, tiny_batches, tiny_state  # noqa: E402
from unlearn_saliency_amd import rng  # noqa: E402


def _args(**kw):
    base = dict(unlearn_lr=0.013, momentum=0.9, weight_decay=5e-4, decreasing_lr="91,136", rewind_epoch=0,
                imagenet_arch=False, unlearn_epochs=2, dataset="cifar10", num_classes=10, warmup=0, print_freq=50,
                batch_size=16)
    base.update(kw)
    return SimpleNamespace(**base)


def _mask_for(model, seed):
    sizes = [p.numel() for p in model.parameters()]
    mflat = (rng.u8(sum(sizes), seed) & 1).astype(np.int64)
    off = np.cumsum([0] + sizes)
    return mflat, {n: torch.from_numpy(mflat[off[i]:off[i + 1]]).view_as(p)
                   for i, (n, p) in enumerate(model.named_parameters())}


def main():
    _, ref_unlearn = MG.import_reference_classification()
    crit = nn.CrossEntropyLoss()

    # ---- F1 boundary_shrink (masked / unmasked): 2 forget batches, 2 epochs
    for tag, use_mask in (("masked", True), ("unmasked", False)):
        model = TinyCNN()
        model.load_state_dict(tiny_state(21))
        fb = tiny_batches(2, 16, 700)
        forget = MG._ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in fb])
        mflat, maskd = _mask_for(model, 900)
        ref_unlearn.boundary_shrink({"forget": forget}, model, crit, _args(unlearn="boundary_shrink"),
                                    maskd if use_mask else None)
        np.savez(os.path.join(HERE, f"boundary_shrink_{tag}.npz"),
                 mask=mflat.astype(np.uint8) if use_mask else np.zeros(0, np.uint8),
                 **{"sd_" + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- F1 boundary_expanding (the reference cannot take a mask here: old-shape mask x expanded layer)
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    fb = tiny_batches(2, 16, 700)
    forget = MG._ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in fb])
    torch.manual_seed(9)  # the new layer's initialisation
    ref_unlearn.boundary_expanding({"forget": forget}, model, crit, _args(unlearn="boundary_expanding"), None)
    np.savez(os.path.join(HERE, "boundary_expanding.npz"), init_seed=9,
             **{"sd_" + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- F1 GA / GA_l1 / FT / FT_l1 (Classification/unlearn/GA.py:44-206, FT.py:44-180): 2 epochs on the tiny BN
    # network; GA walks 2 forget batches, FT walks 3 retain batches.  The returned value of the epoch function (train
    # top-1) is captured through the print the reference makes of it.
    #   GA_l1 is declared WITHOUT the `mask` parameter (GA.py:156) while the epoch driver always passes one
    #   (impl.py:108-110): through the registry it raises TypeError in the reference.  Its golden is produced by
    #   driving the undecorated epoch function with the driver's own optimizer / scheduler construction.
    def run_plain(name, loaders, args, maskd):
        model = TinyCNN()
        model.load_state_dict(tiny_state(21))
        getattr(ref_unlearn, name)(loaders, model, crit, args, maskd)
        return model

    fb = tiny_batches(2, 16, 700)
    rb = tiny_batches(3, 16, 800)
    mk = lambda bs: MG._ListLoader([(torch.from_numpy(x), torch.from_numpy(y)) for x, y in bs])
    for name, key, batches in (("GA", "forget", fb), ("FT", "retain", rb), ("FT_l1", "retain", rb)):
        for tag, use_mask in (("masked", True), ("unmasked", False)):
            mflat, maskd = _mask_for(TinyCNN(), 900)
            model = run_plain(name, {key: mk(batches)}, _args(unlearn=name, alpha=2e-3, no_l1_epochs=0),
                              maskd if use_mask else None)
            np.savez(os.path.join(HERE, f"{name.lower()}_{tag}.npz"), alpha=2e-3, no_l1_epochs=0,
                     mask=mflat.astype(np.uint8) if use_mask else np.zeros(0, np.uint8),
                     **{"sd_" + k: v.numpy() for k, v in model.state_dict().items()})
    # GA_l1: the undecorated function + the driver's optimizer (impl.py:68-73, 96-99)
    raised = None
    try:
        run_plain("GA_l1", {"forget": mk(fb)}, _args(unlearn="GA_l1", alpha=2e-3), None)
    except TypeError as e:
        raised = str(e)
    assert raised and "positional argument" in raised, raised
    inner = ref_unlearn.GA_l1.__closure__[0].cell_contents
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    a = _args(unlearn="GA_l1", alpha=2e-3)
    opt = torch.optim.SGD(model.parameters(), a.unlearn_lr, momentum=a.momentum, weight_decay=a.weight_decay)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[91, 136], gamma=0.1)
    accs = []
    for epoch in range(a.unlearn_epochs):
        accs.append(float(inner({"forget": mk(fb)}, model, crit, opt, epoch, a)))
        sched.step()
    np.savez(os.path.join(HERE, "ga_l1_unmasked.npz"), alpha=2e-3, reference_registry_error=raised,
             train_acc=np.array(accs), mask=np.zeros(0, np.uint8),
             **{"sd_" + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- F2 RL_proximal: 24 forget + 40 retain uint8 samples, merged + shuffled by the reference's DataLoader
    fds, rds = next_rows_datasets()
    forget = torch.utils.data.DataLoader(fds, batch_size=16, shuffle=False)
    retain = torch.utils.data.DataLoader(rds, batch_size=16, shuffle=False)
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    thresholds = []
    real_topk = torch.topk

    def rec_topk(*a, **k):
        out = real_topk(*a, **k)
        thresholds.append(float(-out[0][-1]))
        return out

    np.random.seed(7)
    torch.manual_seed(7)
    torch.topk = rec_topk
    try:
        ref_unlearn.RL_proximal({"forget": forget, "retain": retain}, model, crit,
                                _args(unlearn="RL_proximal", mask_ratio=0.5), None)
    finally:
        torch.topk = real_topk
    np.savez(os.path.join(HERE, "rl_proximal.npz"), seed=7, mask_ratio=0.5, thresholds=np.array(thresholds, np.float32),
             **{"sd_" + k: v.numpy() for k, v in model.state_dict().items()})

    # ---- F2 step KAT: the reference's tensor expressions (RL_pro.py:53-58) on a flat pair, three ratios
    n = 5000
    p0 = rng.normal(n, 1200, 0.0, 0.05)
    p = (p0 + rng.normal(n, 1201, 0.0, 0.01)).astype(np.float32)
    outs, taus, ratios = [], [], [1, 1234, 4999]
    for ratio in ratios:
        params, init_params = torch.from_numpy(p.copy()), torch.from_numpy(p0)
        diff_params = params - init_params
        threshold = -torch.topk(-diff_params.abs(), ratio)[0][-1]
        params = torch.where(diff_params > threshold, params - threshold,
                             torch.where(diff_params < -threshold, params + threshold, init_params))
        outs.append(params.numpy())
        taus.append(float(threshold))
    np.savez(os.path.join(HERE, "proximal_step.npz"), n=n, seeds=np.array([1200, 1201]), ratios=np.array(ratios),
             out=np.stack(outs), tau=np.array(taus, np.float32))

    # ---- F3 EWC term: value and autograd gradient of the reference's per-tensor loop (diffusion.py:343-350)
    shapes = [(16, 8, 3, 3), (16,), (32, 16), (32,)]
    lam = 10.0
    ps, stars, Fs = [], [], []
    for i, s in enumerate(shapes):
        k = int(np.prod(s))
        stars.append(rng.normal(k, 1300 + i, 0.0, 0.05).reshape(s))
        ps.append((stars[-1].reshape(-1) + rng.normal(k, 1310 + i, 0.0, 0.01)).astype(np.float32).reshape(s))
        Fs.append(np.abs(rng.normal(k, 1320 + i, 0.0, 1.0)).astype(np.float32).reshape(s))
    params = [nn.Parameter(torch.from_numpy(a.copy())) for a in ps]
    loss = 0.0
    for prm, st, F in zip(params, stars, Fs):
        _loss = torch.from_numpy(F) * (prm - torch.from_numpy(st)) ** 2
        loss = loss + lam * _loss.sum()
    loss.backward()
    np.savez(os.path.join(HERE, "ewc_term.npz"), lam=lam, shapes=np.array([str(s) for s in shapes]),
             loss=float(loss.item()), grad=np.concatenate([q.grad.reshape(-1).numpy() for q in params]))
    print("next-row fixtures written")


if __name__ == "__main__":
    main()
