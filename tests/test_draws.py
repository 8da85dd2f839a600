"""draws.py on the host: global-batch draws sliced per shard, dropout keys, the shard weights (single process; the
world_size-2 runs of the code that uses them are in tests/test_dist_diffusion_gloo.py, the device kernel behind the
dropout keys in tests/test_kernels_gpu.py)."""
import numpy as np
import torch

from unlearn_saliency_amd import draws


def test_generator_draws_are_global_and_sliced():
    x = torch.zeros(3, 2, 4)
    torch.manual_seed(5)
    whole = torch.randn(7, 2, 4)
    torch.manual_seed(5)
    a = draws.randn_like(x, draws.Shard(0, 3, 7))
    torch.manual_seed(5)
    y = torch.zeros(4, 2, 4)
    b = draws.randn_like(y, draws.Shard(3, 7, 7))
    assert torch.equal(torch.cat([a, b]), whole)
    torch.manual_seed(9)
    t_whole = torch.randint(0, 1000, (7,))
    torch.manual_seed(9)
    with draws.scope(draws.Shard(3, 7, 7)):
        t = draws.randint(1000, 4, "cpu")
    assert torch.equal(t, t_whole[3:])
    # no scope / unsliced shard: plain local draws, same generator consumption as torch's own call
    torch.manual_seed(9)
    assert torch.equal(draws.randint(1000, 7, "cpu"), t_whole)
    torch.manual_seed(5)
    assert torch.equal(draws.randn_like(torch.zeros(7, 2, 4), draws.Shard(0, 7, 7, sliced=False)), whole)


def test_label_drop_mask_of_the_model_is_drawn_for_the_global_batch():
    from unlearn_saliency_amd.DDPM.models.diffusion import prob_mask_like
    fn = lambda n: prob_mask_like((n,), 0.7, "cpu")
    torch.manual_seed(3)
    whole = fn(10)
    parts = []
    for lo, hi in ((0, 5), (5, 10)):
        torch.manual_seed(3)
        with draws.scope(draws.Shard(lo, hi, 10)):
            parts.append(draws.batch_draw(hi - lo, fn))
    assert torch.equal(torch.cat(parts), whole)
    assert 0 < int(whole.sum()) < 10  # the two shards see different decisions, not one pattern repeated
    assert not torch.equal(parts[0], parts[1])


def test_dropout_keys_step_and_call_indexed_and_rank_independent():
    draws.seed(11)
    draws.next_step()
    k1, off1 = draws.dropout_key()
    k2, _ = draws.dropout_key()
    draws.next_step()
    k3, _ = draws.dropout_key()
    assert len({k1, k2, k3}) == 3 and off1 == 0
    draws.seed(11)
    draws.next_step()
    with draws.scope(draws.Shard(6, 9, 12)):
        k1b, off = draws.dropout_key()
    assert (k1b, off) == (k1, 6)  # same key on every rank; only the global sample offset differs
    draws.seed(None)
    torch.manual_seed(77)
    draws.next_step()
    a = draws.dropout_key()[0]
    torch.manual_seed(78)
    draws.set_state((None, 1, 0))
    assert draws.dropout_key()[0] != a  # follows torch.manual_seed when no explicit base is set


def test_counter_dropout_host_path_concatenates_over_shards():
    d = draws.CounterDropout(0.3).train()
    x = torch.from_numpy(np.arange(6 * 5, dtype=np.float32).reshape(6, 5) + 1.0)
    torch.manual_seed(21)
    whole = d(x)
    parts = []
    for lo, hi in ((0, 2), (2, 6)):
        torch.manual_seed(21)
        with draws.scope(draws.Shard(lo, hi, 6)):
            parts.append(d(x[lo:hi]))
    assert torch.equal(torch.cat(parts), whole)
    kept = whole != 0
    assert 0 < int(kept.sum()) < kept.numel()
    assert torch.allclose(whole[kept], (x / 0.7)[kept])
    assert d.eval()(x) is x


def test_shard_weights():
    s = draws.Shard(2, 5, 8)
    assert s.n == 3 and s.share == 3 / 8 and s.weight == 1.0  # single process: AVG over one rank is the identity
    assert draws.Shard(0, 4, 4, sliced=False).share == 1.0
