"""world_size-2 tests of the data-parallel scheme on CPU (backend gloo, 127.0.0.1).

The HIP kernels need a GPU, so what is exercised here is everything *around* them that makes the N>1 path
correct by construction: rendezvous from the torchrun environment, contiguous batch
sharding, the flat all-reduces, and the sharding algebra — per-rank (n_local/n_global)-weighted shard
gradients summed over ranks equal the single-process batch-mean gradient, for the saliency accumulator
(Phase A, eval mode) and for the training gradient (Phase B) — with the oracle standing in for the kernels.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, fn_name, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(WORLD), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from unlearn_saliency_amd import dist as sdist
    rk, lrk, ws = sdist.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, WORLD) and sdist.is_dist() and sdist.world_size() == WORLD
    try:
        globals()[fn_name](rank, out_dir)
    finally:
        sdist.barrier()
        torch.distributed.destroy_process_group()


def _run(fn_name, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(port, fn_name, str(tmp_path)), nprocs=WORLD, join=True)


# ----------------------------------------------------------------- collectives
def _collectives(rank, out_dir):
    from unlearn_saliency_amd import dist as sdist
    v = torch.arange(10, dtype=torch.float32) * (rank + 1)
    sdist.all_reduce_sum_(v)
    assert torch.equal(v, torch.arange(10, dtype=torch.float32) * 3)
    g = torch.full((7,), float(rank + 1))
    sdist.all_reduce_mean_(g)
    assert torch.allclose(g, torch.full((7,), 1.5))
    assert sdist.shard_bounds(10) == ((0, 5) if rank == 0 else (5, 10))
    assert sdist.shard_bounds(9) == ((0, 5) if rank == 0 else (5, 9))
    assert sdist.shard_bounds(1) == ((0, 1) if rank == 0 else (1, 1))


def test_collectives_and_shard_bounds(tmp_path):
    _run("_collectives", tmp_path)


# -------------------------------------------------------------- loader sharding
def _loader_shards(rank, out_dir):
    from unlearn_saliency_amd.Classification.dataset import ArrayDataset, BatchLoader, synthetic_cifar10
    (x, y), _ = synthetic_cifar10(n_train=1000, n_test=10)
    ds = ArrayDataset(x, y, transform="test")
    torch.manual_seed(7)  # same seed on every rank => same permutation => consistent shards
    got = [(xb.clone(), yb.clone()) for xb, yb in BatchLoader(ds, 256, True, rank=rank, world_size=WORLD)]
    torch.save(got, os.path.join(out_dir, f"shards_{rank}.pt"))


def test_batch_loader_shards_partition_each_global_batch(tmp_path):
    _run("_loader_shards", tmp_path)
    sys.path.insert(0, ROOT)
    from unlearn_saliency_amd.Classification.dataset import ArrayDataset, BatchLoader, synthetic_cifar10
    (x, y), _ = synthetic_cifar10(n_train=1000, n_test=10)
    torch.manual_seed(7)
    full = list(BatchLoader(ArrayDataset(x, y, transform="test"), 256, True))
    s0 = torch.load(tmp_path / "shards_0.pt")
    s1 = torch.load(tmp_path / "shards_1.pt")
    assert len(full) == len(s0) == len(s1) == 4
    for (xf, yf), (x0, y0), (x1, y1) in zip(full, s0, s1):
        assert torch.equal(torch.cat([x0, x1]), xf) and torch.equal(torch.cat([y0, y1]), yf)
    assert s0[-1][0].shape[0] + s1[-1][0].shape[0] == 1000 - 3 * 256  # ragged last global batch


# ------------------------------------------------------ sharded saliency == single process
def _sharded_saliency(rank, out_dir):
    import oracle
    from fixtures import TinyCNN, tiny_batches, tiny_state
    from unlearn_saliency_amd import dist as sdist
    model = TinyCNN()
    model.load_state_dict(tiny_state(11))
    model.eval()
    crit = nn.CrossEntropyLoss()
    n = sum(p.numel() for p in model.parameters())
    acc = np.zeros(n, np.float32)
    batches = tiny_batches(3, 16, 500)
    batches[-1] = (batches[-1][0][:9], batches[-1][1][:9])  # ragged: shards of 5 and 4
    for x, y in batches:
        lo, hi = sdist.shard_bounds(len(y))
        xs, ys = torch.from_numpy(x[lo:hi]), torch.from_numpy(y[lo:hi])
        cnt = torch.tensor([float(hi - lo)])
        sdist.all_reduce_sum_(cnt)
        model.zero_grad()
        (-crit(model(xs), ys)).backward()
        g = np.concatenate([p.grad.reshape(-1).numpy() for p in model.parameters()])
        oracle.saliency_accumulate(acc, g, (hi - lo) / float(cnt.item()))  # same rule as generate_mask.accumulate_saliency
    t = torch.from_numpy(acc)
    sdist.all_reduce_sum_(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "acc.npy"), t.numpy())


def test_sharded_saliency_equals_reference_accumulator(tmp_path, golden_dir):
    _run("_sharded_saliency", tmp_path)
    acc = np.load(tmp_path / "acc.npy")
    g = np.load(os.path.join(golden_dir, "saliency_tinycnn.npz"))
    # equal to the REFERENCE's single-process accumulator up to fp32 summation order
    assert np.allclose(acc, g["acc"], rtol=1e-5, atol=1e-5 * np.abs(g["acc"]).max())
    import oracle
    m = oracle.mask_topk(acc, [oracle.k_of(acc.size, 0.5)])[0]
    assert int(m.sum()) == int(g["mask_05"].sum()) and (m != g["mask_05"]).sum() <= 2


# -------------------------------------------------- sharded training gradient == single process
def _sharded_step(rank, out_dir):
    import oracle
    from fixtures import TinyCNN, tiny_batches, tiny_state
    from unlearn_saliency_amd import dist as sdist
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()  # BN statistics are per-replica in train mode (documented); eval makes the identity exact
    crit = nn.CrossEntropyLoss()
    x, y = tiny_batches(1, 16, 700)[0]
    lo, hi = sdist.shard_bounds(16)
    model.zero_grad()
    crit(model(torch.from_numpy(x[lo:hi])), torch.from_numpy(y[lo:hi])).backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    sdist.all_reduce_mean_(g)  # what FusedMaskedSGD.step() does before the kernel
    flat = np.concatenate([p.detach().reshape(-1).numpy() for p in model.parameters()]).astype(np.float32)
    buf = np.zeros_like(flat)
    mask = (np.arange(flat.size) % 2).astype(np.uint8)
    oracle.masked_sgd_step(flat, g.numpy(), buf, mask, 0.013, 0.9, 5e-4, True)
    np.save(os.path.join(out_dir, f"p_{rank}.npy"), flat)


def test_sharded_step_equals_single_process_and_ranks_agree(tmp_path):
    _run("_sharded_step", tmp_path)
    p0, p1 = np.load(tmp_path / "p_0.npy"), np.load(tmp_path / "p_1.npy")
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))  # replicas stay bit-identical: no broadcast needed
    sys.path.insert(0, ROOT)
    import oracle
    from fixtures import TinyCNN, tiny_batches, tiny_state
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()
    x, y = tiny_batches(1, 16, 700)[0]
    nn.CrossEntropyLoss()(model(torch.from_numpy(x)), torch.from_numpy(y)).backward()
    g = np.concatenate([p.grad.reshape(-1).numpy() for p in model.parameters()])
    flat = np.concatenate([p.detach().reshape(-1).numpy() for p in model.parameters()]).astype(np.float32)
    buf = np.zeros_like(flat)
    oracle.masked_sgd_step(flat, g, buf, (np.arange(flat.size) % 2).astype(np.uint8), 0.013, 0.9, 5e-4, True)
    assert np.allclose(p0, flat, rtol=1e-6, atol=1e-8)


# -------------------------------------------------- bucketed, overlapped gradient all-reduce
def _bucketed(rank, out_dir):
    from fixtures import TinyCNN, tiny_batches, tiny_state
    from unlearn_saliency_amd import dist as sdist
    from unlearn_saliency_amd.flat import FlatArena
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()
    arena = FlatArena.from_module(model, device="cpu")
    red = sdist.BucketedGradReducer(arena, num_buckets=3)
    assert red.bounds[0][0] == 0 and red.bounds[-1][1] == arena.n and len(red.bounds) >= 2
    assert all(a[1] == b[0] for a, b in zip(red.bounds, red.bounds[1:]))
    x, y = tiny_batches(1, 16, 700)[0]
    lo, hi = sdist.shard_bounds(16)
    for _ in range(2):  # two steps: state resets between them
        arena.zero_grad()
        nn.CrossEntropyLoss()(model(torch.from_numpy(x[lo:hi])), torch.from_numpy(y[lo:hi])).backward()
        assert any(red.launched)  # slices went out during backward
        red.finish()
        assert not any(red.launched) and not red.works
    np.save(os.path.join(out_dir, f"g_{rank}.npy"), arena.grads.numpy())
    # hooks firing in the opposite order (first parameters first) must still issue the slices last-first —
    # the one order every rank uses, so slices of different sizes always pair up across ranks
    keep = arena.grads.clone()
    for i in range(len(arena._params)):
        red._make_hook(i)(None)
        assert red._order == ([] if i < len(arena._params) - 1 else list(range(len(red.bounds) - 1, -1, -1)))
    red.finish()
    assert torch.allclose(arena.grads, keep)  # AVG of identical vectors


def test_bucketed_overlapped_allreduce_equals_full_batch_gradient(tmp_path):
    _run("_bucketed", tmp_path)
    g0, g1 = np.load(tmp_path / "g_0.npy"), np.load(tmp_path / "g_1.npy")
    assert np.array_equal(g0, g1)
    sys.path.insert(0, ROOT)
    from fixtures import TinyCNN, tiny_batches, tiny_state
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()
    x, y = tiny_batches(1, 16, 700)[0]
    nn.CrossEntropyLoss()(model(torch.from_numpy(x)), torch.from_numpy(y)).backward()
    g = np.concatenate([p.grad.reshape(-1).numpy() for p in model.parameters()])
    assert np.allclose(g0, g, rtol=1e-5, atol=1e-7)


# -------------------------------------------------- a rank with an EMPTY shard (tail batch smaller than the world)
def _empty_shard(rank, out_dir):
    """a rank whose shard of a ragged tail batch is empty runs no backward; its buckets must still pair
    up with the other rank's hook-launched slices (different sizes).  The reducer issues slices in ONE order on every
    rank — last slice first — whether a hook or `finish` issues them, also when the ragged step is the first step."""
    from fixtures import TinyCNN, tiny_batches, tiny_state
    from unlearn_saliency_amd import dist as sdist
    from unlearn_saliency_amd.flat import FlatArena
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()
    arena = FlatArena.from_module(model, device="cpu")
    red = sdist.BucketedGradReducer(arena, num_buckets=3)
    sizes = {hi - lo for lo, hi in red.bounds}
    assert len(sizes) > 1, "the case needs slices of different sizes"
    x, y = tiny_batches(1, 16, 700)[0]
    for first_step_empty in (False, True):
        if not first_step_empty:  # one complete step on both ranks first; then the ragged step is the FIRST of the run
            lo, hi = sdist.balanced_slice(16)
            arena.zero_grad()
            nn.CrossEntropyLoss()(model(torch.from_numpy(x[lo:hi])), torch.from_numpy(y[lo:hi])).backward()
            red.finish()
        # tail batch of ONE sample: balanced_slice gives rank 0 nothing, rank 1 the sample
        lo, hi = sdist.balanced_slice(1)
        assert (hi - lo) == (0 if rank == 0 else 1)
        arena.zero_grad()
        if hi > lo:
            w = (hi - lo) * WORLD / 1.0
            (nn.CrossEntropyLoss()(model(torch.from_numpy(x[:1])), torch.from_numpy(y[:1])) * w).backward()
        red.finish()
        np.save(os.path.join(out_dir, f"tail_{int(first_step_empty)}_{rank}.npy"), arena.grads.numpy().copy())


def test_empty_shard_rank_issues_its_buckets_in_the_hooks_order(tmp_path):
    _run("_empty_shard", tmp_path)
    sys.path.insert(0, ROOT)
    from fixtures import TinyCNN, tiny_batches, tiny_state
    model = TinyCNN()
    model.load_state_dict(tiny_state(21))
    model.eval()
    x, y = tiny_batches(1, 16, 700)[0]
    nn.CrossEntropyLoss()(model(torch.from_numpy(x[:1])), torch.from_numpy(y[:1])).backward()
    g = np.concatenate([p.grad.reshape(-1).numpy() for p in model.parameters()])
    for tag in (0, 1):
        g0, g1 = np.load(tmp_path / f"tail_{tag}_0.npy"), np.load(tmp_path / f"tail_{tag}_1.npy")
        assert np.array_equal(g0, g1)
        assert np.allclose(g0, g, rtol=1e-6, atol=1e-8)  # AVG over ranks of (2 x sample gradient, 0) = the batch mean
