"""Data-parallel path on the real kernels, two ranks sharing the one GPU of the test box (backend gloo through
SALUN_DIST_BACKEND, since RCCL refuses two ranks on one device): fused ResNet blocks + in-kernel gradient
accumulation (gradsink) + bucketed overlapped all-reduce + fused masked SGD.  Ranks must stay bit-identical and
equal the single-process full-batch run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    if world > 1:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(port), SALUN_DIST_BACKEND="gloo")
    else:
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    import torch.nn as nn
    from unlearn_saliency_amd import dist as sdist, ops
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.flat import arena_of
    from unlearn_saliency_amd.norm import use_fused_bn
    from unlearn_saliency_amd.optim import FusedMaskedSGD
    rk, _, ws = sdist.init_from_env()
    torch.manual_seed(0)
    model = model_dict["resnet18"](num_classes=10).cuda()
    use_salun_convs(model)
    use_fused_bn(model)
    model.eval()  # batch statistics are per replica in train mode (documented); eval makes the identity exact
    arena = arena_of(model)
    opt = FusedMaskedSGD(arena, 0.013, momentum=0.9, weight_decay=5e-4)
    opt.set_mask(ops.mask_topk(ops.fill_normal(arena.n, 5, 0.0, 1e-3), [arena.n // 2])[0])
    g = torch.Generator().manual_seed(1)
    x = torch.rand(64, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (64,), generator=g)
    crit = nn.CrossEntropyLoss()
    if rk == 0:
        np.save(os.path.join(out_dir, f"init_w{world}.npy"), arena.params.cpu().numpy())
    for _ in range(2):
        opt.zero_grad()
        if ws > 1:
            lo, hi = sdist.shard_bounds(64)
            crit(model(x[lo:hi].cuda()), y[lo:hi].cuda()).backward()
        else:
            # the same arithmetic in one process: the two shards' mean-loss gradients accumulated with weight 1/2
            # (same kernel tilings as the ranks use, so only the order of the final average differs)
            for lo, hi in ((0, 32), (32, 64)):
                (0.5 * crit(model(x[lo:hi].cuda()), y[lo:hi].cuda())).backward()
        opt.step()
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"p_w{world}_r{rank}.npy"), arena.params.cpu().numpy())
    if ws > 1:
        sdist.barrier()
        torch.distributed.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    mp.spawn(_run, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_run, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = np.load(tmp_path / "p_w1_r0.npy")
    a, b = np.load(tmp_path / "p_w2_r0.npy"), np.load(tmp_path / "p_w2_r1.npy")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # replicas never diverge: no broadcast needed
    assert np.isfinite(one).all() and not np.array_equal(one, a * 0)
    init = np.load(tmp_path / "init_w1.npy")
    assert np.array_equal(init, np.load(tmp_path / "init_w2.npy"))
    # same update up to the rounding of the final average: compare the UPDATE, per 4096-element block, against its
    # own scale
    upd1, upd2 = (one - init).astype(np.float64), (a - init).astype(np.float64)
    n = upd1.size // 4096 * 4096
    scale = np.abs(upd1[:n]).reshape(-1, 4096).max(axis=1)
    err = np.abs(upd1[:n] - upd2[:n]).reshape(-1, 4096).max(axis=1)
    assert scale.max() > 0
    assert (err <= 1e-4 * scale + 1e-9).all(), float((err / (scale + 1e-30)).max())
