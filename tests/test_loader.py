"""BatchLoader walks a dataset in exactly the order torch's DataLoader(shuffle=True) would under the same global
seed (a base-seed draw, then RandomSampler's seed draw, then randperm) — which is what lets runs of this package be
compared with runs of the reference batch for batch (RL_proximal golden, tests/test_next_oracle_vs_golden.py).  CPU."""
import numpy as np
import torch

from fixtures import next_rows_datasets
from unlearn_saliency_amd.Classification.dataset import ArrayDataset, BatchLoader


def test_shuffled_order_equals_torch_dataloader():
    fds, rds = next_rows_datasets()
    ds = ArrayDataset(np.concatenate([fds.data, rds.data]), np.concatenate([fds.targets, rds.targets]), transform="test")
    for seed in (0, 7, 123):
        torch.manual_seed(seed)
        ours = [(x.clone(), y.clone()) for _ in range(2) for x, y in BatchLoader(ds, 16, True)]  # two epochs
        torch.manual_seed(seed)
        dl = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True)
        ref = [(x.clone(), y.clone()) for _ in range(2) for x, y in dl]
        assert len(ours) == len(ref) == 8
        for (xa, ya), (xb, yb) in zip(ours, ref):
            assert torch.equal(xa, xb) and torch.equal(ya, yb)


def test_unshuffled_and_ragged_tail():
    fds, _ = next_rows_datasets()
    batches = list(BatchLoader(fds, 10, False))
    assert [b[0].shape[0] for b in batches] == [10, 10, 4] and len(BatchLoader(fds, 10, False)) == 3
    assert torch.equal(torch.cat([b[1] for b in batches]), torch.from_numpy(np.asarray(fds.targets)))


def test_rank_shards_partition_every_global_batch():
    fds, rds = next_rows_datasets()
    torch.manual_seed(3)
    full = list(BatchLoader(rds, 16, True))
    shards = []
    for r in range(3):
        torch.manual_seed(3)
        shards.append(list(BatchLoader(rds, 16, True, rank=r, world_size=3)))
    for i, (x, y) in enumerate(full):
        assert torch.equal(torch.cat([s[i][0] for s in shards]), x)
        assert torch.equal(torch.cat([s[i][1] for s in shards]), y)
