"""GPU parity: every HIP kernel, called through the C-ABI (unlearn_saliency_amd.ops ->
libsalun.so), against the CPU oracle on the same seeded inputs.

Bars (north_star): mask indices bit-exact; element-wise updates bit-exact against the
oracle (identical rounding order by construction), reductions within 1e-6 relative.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N18 = 11_173_962  # ResNet-18 (CIFAR) parameter count, SURVEY.md §0 fact 8
ND = 38_632_323   # CFG-DDPM U-Net


@pytest.fixture(scope="module")
def ops():
    from unlearn_saliency_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint32)


# ------------------------------------------------------------------- generators
@pytest.mark.parametrize("n", [0, 1, 5, 4096, 100_003])
def test_generators_bit_exact(ops, oracle_mod, n):
    assert np.array_equal(bits(ops.fill_uniform(n, 7, -1.0, 3.0).cpu().numpy()),
                          bits(oracle_mod.fill_uniform(n, 7, -1.0, 3.0)))
    assert np.array_equal(bits(ops.fill_normal(n, 11, 0.5, 2.0).cpu().numpy()),
                          bits(oracle_mod.fill_normal(n, 11, 0.5, 2.0)))
    assert np.array_equal(ops.fill_u8(n, 3).cpu().numpy(), oracle_mod.fill_u8(n, 3))


def test_normal_generator_moments(oracle_mod):
    z = oracle_mod.fill_normal(1_000_000, 123)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3


# ------------------------------------------------------------- counter-based dropout
@pytest.mark.parametrize("shape,off", [((4, 6), 0), ((3, 7), 5), ((5, 3, 8, 8), 2), ((16, 128, 16, 16), 48)])
@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_bit_exact_and_is_its_own_backward(ops, oracle_mod, shape, off, p):
    x = oracle_mod.fill_normal(int(np.prod(shape)), 21).reshape(shape)
    key = 0x9E3779B97F4A7C15 ^ (off * 1315423911)
    want = oracle_mod.dropout(x, p, key, off)
    got = ops.dropout(dev(x), p, key, off)
    assert np.array_equal(bits(got.cpu().numpy()), bits(want))
    drop = float((want == 0).mean())
    assert abs(drop - p) < (0.25 if x.size < 1000 else 0.01)
    # unaligned view -> scalar path, same bits; in place
    buf = torch.zeros(x.size + 1, device="cuda")
    buf[1:] = dev(x).reshape(-1)
    v = buf[1:].view(*shape)
    ops.dropout(v, p, key, off, out=v)
    assert np.array_equal(bits(v.cpu().numpy()), bits(want))
    # autograd: d/dx = the same keep mask and scale
    xd = dev(x).requires_grad_(True)
    ops.dropout_fn(xd, p, key, off).backward(torch.ones_like(xd))
    assert np.array_equal(bits(xd.grad.cpu().numpy()), bits(oracle_mod.dropout(np.ones_like(x), p, key, off)))


def test_dropout_shards_concatenate_to_the_global_batch(ops):
    """DDPM activation at bench size: rows [lo, hi) computed with sample_offset = lo ARE rows lo..hi-1 of the whole."""
    x = ops.fill_normal(128 * 128 * 32 * 32, 5).view(128, 128, 32, 32)
    whole = ops.dropout(x, 0.1, 777, 0)
    parts = [ops.dropout(x[lo:hi].contiguous(), 0.1, 777, lo) for lo, hi in ((0, 16), (16, 64), (64, 65), (65, 128))]
    assert torch.equal(torch.cat(parts), whole)
    assert abs(float((whole == 0).float().mean()) - 0.1) < 1e-3
    assert not torch.equal(ops.dropout(x, 0.1, 778, 0), whole)
    seed_dev = torch.tensor([5], dtype=torch.int64, device="cuda")  # device-resident part of the key (graph replays)
    assert torch.equal(ops.dropout(x, 0.1, 772, 0, seed_dev=seed_dev), whole)


# --------------------------------------------------------------------------- K1
@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4099, 1_000_003])
@pytest.mark.parametrize("scale", [1.0, 0.37])
def test_saliency_accumulate(ops, oracle_mod, n, scale):
    acc = oracle_mod.fill_normal(n, 1, 0, 1e-3)
    g = oracle_mod.fill_normal(n, 2, 0, 1e-2)
    d_acc, d_g = dev(acc), dev(g)
    for _ in range(3):
        ops.saliency_accumulate(d_acc, d_g, scale)
        oracle_mod.saliency_accumulate(acc, g, scale)
    assert np.array_equal(bits(d_acc.cpu().numpy()), bits(acc))


def test_saliency_accumulate_device_clip(ops, oracle_mod):
    n = 50_001
    acc = np.zeros(n, np.float32)
    g = oracle_mod.fill_normal(n, 5, 0, 0.05)  # norm ~ 11 > 1 => clipped
    d_acc, d_g = dev(acc), dev(g)
    sq = ops.grad_sqnorm(d_g)
    ops.saliency_accumulate(d_acc, d_g, sqnorm=sq, max_norm=1.0)
    coef = oracle_mod.clip_coef(float(sq.item()), 1.0)
    assert coef < 1.0
    oracle_mod.saliency_accumulate(acc, g, coef)
    assert np.array_equal(bits(d_acc.cpu().numpy()), bits(acc))


def test_unaligned_views_take_scalar_path(ops, oracle_mod):
    n = 10_007
    acc = oracle_mod.fill_normal(n + 1, 1)
    g = oracle_mod.fill_normal(n + 3, 2)
    d_acc, d_g = dev(acc)[1:], dev(g)[3:]  # 4-byte aligned only
    ops.saliency_accumulate(d_acc, d_g, 1.0)
    a2 = acc[1:].copy()
    oracle_mod.saliency_accumulate(a2, np.ascontiguousarray(g[3:]), 1.0)
    assert np.array_equal(bits(d_acc.cpu().numpy()), bits(a2))


# --------------------------------------------------------------------------- K2
def _check_topk(ops, oracle_mod, acc, ks):
    want = oracle_mod.mask_topk(acc, ks)
    got = ops.mask_topk(dev(acc), ks)
    n = acc.size
    for k, w, g in zip(ks, want, got):
        g = g.cpu().numpy()
        assert int(g.sum()) == min(max(int(k), 0), n), (k, int(g.sum()))
        assert np.array_equal(g, w), f"mask differs for k={k}: {(g != w).sum()} positions"


@pytest.mark.parametrize("n", [1, 2, 7, 64, 4095, 4096, 4097, 123_457])
def test_topk_small_sizes_all_ratios(ops, oracle_mod, n):
    acc = oracle_mod.fill_normal(n, 42 + n, 0, 1e-3)
    ks = [oracle_mod.k_of(n, r / 10) for r in range(1, 11)]
    _check_topk(ops, oracle_mod, acc, ks)
    # the numpy double-argsort restatement agrees with the C oracle too
    for a, b in zip(oracle_mod.mask_topk(acc, ks), oracle_mod.mask_topk_numpy(acc, ks)):
        assert np.array_equal(a, b)


def test_topk_edge_ks(ops, oracle_mod):
    n = 10_000
    acc = oracle_mod.fill_normal(n, 9)
    _check_topk(ops, oracle_mod, acc, [0, 1, 2, n - 1, n, n + 5, -3])


def test_topk_ties_zero_block(ops, oracle_mod):
    """tau inside a block of exact zeros -> stable rule (lowest index first)."""
    n = 300_000
    acc = oracle_mod.fill_normal(n, 3, 0, 1e-3)
    acc[50_000:150_000] = 0.0
    acc[200_000:200_100] = -0.0
    ks = [oracle_mod.k_of(n, r) for r in (0.2, 0.5, 0.7, 0.8, 0.95, 1.0)]
    _check_topk(ops, oracle_mod, acc, ks)


def test_topk_ties_quantised(ops, oracle_mod):
    n = 250_001
    acc = np.round(oracle_mod.fill_normal(n, 4) * 8).astype(np.float32) / 8  # ~40 distinct values
    ks = [oracle_mod.k_of(n, r / 10) for r in range(1, 11)] + [12345, 99_999]
    _check_topk(ops, oracle_mod, acc, ks[:12])


def test_topk_all_equal(ops, oracle_mod):
    n = 70_000
    acc = np.full(n, 0.25, np.float32)
    acc[::2] *= -1  # sign must not matter
    _check_topk(ops, oracle_mod, acc, [1, 4095, 4096, 4097, 35_000, n - 1])


def test_topk_nan_inf_denormal(ops, oracle_mod):
    n = 20_000
    acc = oracle_mod.fill_normal(n, 8)
    acc[5] = np.nan
    acc[77] = -np.nan
    acc[100] = np.inf
    acc[101] = -np.inf
    acc[200:300] = 1e-42  # denormals
    acc[300:400] = 0.0
    _check_topk(ops, oracle_mod, acc, [1, 2, 3, 10_000, n - 150, n - 3, n - 2, n - 1, n])


def test_topk_does_not_modify_input_and_unaligned(ops, oracle_mod):
    n = 33_333
    acc = oracle_mod.fill_normal(n + 1, 17)
    d = dev(acc)[1:]
    before = d.clone()
    k = [n // 2]
    got = ops.mask_topk(d, k)[0].cpu().numpy()
    assert torch.equal(d, before)
    assert np.array_equal(got, oracle_mod.mask_topk(np.ascontiguousarray(acc[1:]), k)[0])


def test_topk_resnet18_size_all_ratios(ops, oracle_mod):
    """Full N18 vector, the reference's 10 thresholds in one call; k table of SURVEY Appendix C."""
    acc = oracle_mod.fill_normal(N18, 2024, 0, 1e-3)
    ratios = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
    ks = [oracle_mod.k_of(N18, r) for r in ratios]
    assert ks[4] == 5_586_981 and ks[0] == 1_117_396 and ks[9] == N18
    _check_topk(ops, oracle_mod, acc, ks)


def test_topk_ddpm_size_property(ops, oracle_mod):
    """N_D vector: popcount == k and every selected |x| >= every unselected |x| (size-independent property)."""
    d = ops.fill_normal(ND, 99, 0.0, 1e-3)
    k = int(ND * 0.5)
    assert k == 19_316_161
    m = ops.mask_topk(d, [k])[0]
    assert ops.mask_popcount(m) == k
    a = d.abs()
    lo_sel = a[m.bool()].min().item()
    hi_unsel = a[~m.bool()].max().item()
    assert lo_sel >= hi_unsel
    tau = ops.mask_topk_thresholds(d.device, 1)[0].item()
    assert tau == lo_sel


# ----------------------------------------------------------- single-read route vs full scan (csrc/salun_topk.hip)
FULL, VALUES_ONLY = 1, 2   # include/salun.h SALUN_TOPK_*


def _route_cases(oracle_mod, n):
    from fixtures import saliency_vector_wide
    uniq = saliency_vector_wide(n, 77)                       # almost surely unique threshold: the single-read route publishes
    ties = oracle_mod.fill_normal(n, 5, 0, 1e-3)             # <= 786 K distinct values: every threshold sits in a run of ties
    nans = uniq.copy(); nans[::1000] = np.nan; nans[7] = np.inf
    periodic = np.tile(np.array([1e-3, 5.0, 2e-3, 7.0], np.float32), n // 4 + 1)[:n]  # four values in a fixed cycle
    ramp = (np.arange(n, dtype=np.float32) / n)              # sorted: index and rank fully correlated
    return {"unique": uniq, "ties": ties, "nans": nans, "periodic": periodic, "ramp": ramp}


def _check_tau(ops, d, acc, want):
    tau = ops.mask_topk_thresholds(d.device, 1)[0].item()
    sel = np.abs(acc[want.astype(bool)])
    kth = np.nan if np.isnan(sel).any() else sel.min()
    assert (np.isnan(tau) and np.isnan(kth)) or tau == kth, (tau, kth)


@pytest.mark.parametrize("n", [8192, 70_001, 1_000_003, N18])
def test_topk_single_read_route_bit_exact(ops, oracle_mod, n):
    """Default route for n >= 8192 (sample -> brackets -> ONE pass -> exact resolution among the candidates) against
    the oracle, one threshold per call: published by the single-read route when the threshold is resolvable among the
    candidates (route 1), redone by the persistent full scan on the device otherwise (heavy ties, four-valued data:
    route 2) — bit-identical masks and thresholds either way, and equal to the forced full scan."""
    cases = _route_cases(oracle_mod, n)
    plan = [(name, k) for name in cases for k in (1, 7, n // 10, n // 2, n - n // 7, n - 1, n)]
    if n == N18:  # the CPU oracle sorts 11 M values per call: keep the list short
        plan = [("unique", 1), ("unique", n // 2), ("unique", n - 1), ("periodic", n // 2), ("nans", n // 3),
                ("ties", n // 2)]
    routes = {}
    for name, k in plan:
        acc = cases[name]
        d = dev(acc)
        got = ops.mask_topk(d, [k])[0]
        route, err = ops.mask_topk_status(d.device)
        assert err == 0
        routes.setdefault(name, set()).add(route)
        want = oracle_mod.mask_topk(acc, [k])[0]
        assert np.array_equal(got.cpu().numpy(), want), (name, k, route)
        _check_tau(ops, d, acc, want)
        full = ops.mask_topk(d, [k], flags=FULL)[0]
        assert ops.mask_topk_status(d.device) == (2, 0)
        assert torch.equal(full, got), (name, k)
        _check_tau(ops, d, acc, want)
    assert routes["unique"] == {1}, routes          # generic data never needs the fallback
    assert 2 in routes["periodic"], routes          # 25 % of the vector ties at the threshold: candidate slabs overflow


def test_topk_single_read_ten_thresholds_and_values_only(ops, oracle_mod):
    """All ten ratios of the reference in one call on the single-read route; VALUES_ONLY publishes the same thresholds
    without touching the masks."""
    from fixtures import saliency_vector_wide
    n = 2_000_003
    acc = saliency_vector_wide(n, 5)
    d = dev(acc)
    ks = [oracle_mod.k_of(n, r / 10) for r in range(1, 11)] + [1, n - 1]
    got = ops.mask_topk(d, ks)
    assert ops.mask_topk_status(d.device) == (1, 0)
    taus = ops.mask_topk_thresholds(d.device, len(ks)).cpu().numpy()
    want = oracle_mod.mask_topk(acc, ks)
    for k, g, w, t in zip(ks, got, want, taus):
        assert np.array_equal(g.cpu().numpy(), w), k
        assert t == np.abs(acc[w.astype(bool)]).min(), k
    keep = [g.clone() for g in got]
    assert ops.mask_topk(d, ks, flags=VALUES_ONLY) == []
    assert ops.mask_topk_status(d.device) == (1, 0)
    assert np.array_equal(ops.mask_topk_thresholds(d.device, len(ks)).cpu().numpy(), taus)
    for a, b in zip(keep, got):
        assert torch.equal(a, b)
    ops.mask_topk(d, ks, flags=VALUES_ONLY | FULL)
    assert ops.mask_topk_status(d.device) == (2, 0)
    assert np.array_equal(ops.mask_topk_thresholds(d.device, len(ks)).cpu().numpy(), taus)


def test_topk_moderate_ties_stay_on_the_single_read_route(ops, oracle_mod):
    """A few thousand ties at the threshold (more than the exact-ranking stage holds) are ordered by flat index inside
    k_finish (radix select on the index) without falling back."""
    n = 3_000_000
    acc = oracle_mod.fill_normal(n, 11, 0, 1e-3) * (1.0 + oracle_mod.fill_uniform(n, 12, 0.0, 0.5))
    srt = np.sort(np.abs(acc))[::-1]
    k = n // 2
    tau = srt[k - 1]
    rs = np.random.RandomState(0)
    idx = rs.choice(n, 5000, replace=False)
    acc[idx] = tau * np.where(rs.rand(5000) < 0.5, 1, -1)   # 5000 more elements exactly at the threshold
    d = dev(acc)
    for kk in (k, k + 1234, k + 4000):
        got = ops.mask_topk(d, [kk])[0]
        assert ops.mask_topk_status(d.device) == (1, 0)
        want = oracle_mod.mask_topk(acc, [kk])[0]
        assert np.array_equal(got.cpu().numpy(), want), kk
        _check_tau(ops, d, acc, want)


def test_topk_two_level_route_at_scale(ops, oracle_mod):
    """2^27 + 3 elements: the brackets come from the exact selection on a 2^20 sample (two-level route).  popcount == k,
    selected >= unselected, threshold exported; same mask as the full scan."""
    n = (1 << 27) + 3
    d = ops.fill_normal(n, 123, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 124, 0.0, 0.5))
    k = int(n * 0.5)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    res = {}
    for i, (name, flags) in enumerate((("two-level", 0), ("full scan", FULL))):
        m = ops.mask_topk(d, [k], flags=flags)[0]
        ev[2 * i].record(); m = ops.mask_topk(d, [k], flags=flags)[0]; ev[2 * i + 1].record()
        torch.cuda.synchronize()
        route, err = ops.mask_topk_status(d.device)
        assert err == 0 and route == (2 if flags == FULL else 1), (name, route, err)
        res[name] = (m, ops.mask_topk_thresholds(d.device, 1)[0].item(), ev[2 * i].elapsed_time(ev[2 * i + 1]))
    m, tau, _ = res["two-level"]
    assert ops.mask_popcount(m) == k
    a = d.abs()
    assert a[m.bool()].min().item() >= a[~m.bool()].max().item()
    assert tau == a[m.bool()].min().item()
    assert torch.equal(m, res["full scan"][0]) and res["full scan"][1] == tau
    print("top-k at N = 2^27+3: " + ", ".join(f"{k_} {v[2]:.3f} ms" for k_, v in res.items()))


def _check_route1(ops, oracle_mod, acc, ks, what=""):
    """Masks bit-exact against the oracle, thresholds = the k-th |acc|, and the call stayed on the single-read route."""
    d = dev(acc)
    got = ops.mask_topk(d, ks)
    assert ops.mask_topk_status(d.device) == (1, 0), (what, ks)
    taus = ops.mask_topk_thresholds(d.device, len(ks)).cpu().numpy()
    want = oracle_mod.mask_topk(acc, ks)
    n = acc.size
    for k, g, w, t in zip(ks, got, want, taus):
        g = g.cpu().numpy()
        assert int(g.sum()) == min(max(int(k), 0), n), (what, k)
        assert np.array_equal(g, w), (what, k, int((g != w).sum()))
        if 0 < k <= n:
            sel = np.abs(acc[w.astype(bool)])
            kth = np.nan if np.isnan(sel).any() else sel.min()
            assert (np.isnan(t) and np.isnan(kth)) or t == kth, (what, k, t, kth)
    return got


@pytest.mark.parametrize("n", [8192, 300_001, 3_000_000])
def test_topk_zero_block_stays_on_the_single_read_route(ops, oracle_mod, n):
    """A real accumulator holds several per cent of exact zeros (Classification/generate_mask.py:46-80 ranks them like
    any other value): thresholds inside the zero block are resolved by the tie pass (lowest flat index first), k >= n is
    all ones without a select — none of them falls back to the full scan."""
    acc = oracle_mod.fill_normal(n, 31, 0, 1e-3) * (1.0 + oracle_mod.fill_uniform(n, 32, 0.0, 0.5))
    rs = np.random.RandomState(n % 1000)
    z = rs.rand(n) < 0.30                       # scattered zeros ...
    acc[z] = 0.0
    acc[n // 3: n // 3 + n // 10] = 0.0         # ... and a contiguous block of them
    acc[5::97] *= np.where(acc[5::97] == 0, -1.0, 1.0)  # some of them -0.0
    nz = int((acc != 0).sum())
    assert 0.5 * n < nz < 0.7 * n
    for k in (nz - 1, nz, nz + 1, nz + 2, (nz + n) // 2, n - 2, n - 1, n, n + 7):
        _check_route1(ops, oracle_mod, acc, [k], f"n={n}")
    # one call: ordinary thresholds, two different zero budgets, everything, nothing
    _check_route1(ops, oracle_mod, acc, [n // 10, n // 2, nz + 5, n - 100, n, 0, n // 4, nz - 3], f"n={n} mixed")


@pytest.mark.parametrize("n", [3_000_000, N18])
def test_topk_local_concentration_spills_instead_of_falling_back(ops, oracle_mod, n):
    """A real accumulator is not i.i.d. along the flat index: a layer whose magnitudes sit right at a threshold gives the
    workgroups that stream it many times the average share of candidates (on the bench's ResNet-18 accumulator two
    workgroups exceeded a 1.5x slab at ratios 0.8 / 0.9 and sent all ten ratios to the full scan).  What a slab cannot
    take goes to the threshold's spill row: still the single-read route, still bit-exact."""
    acc = oracle_mod.fill_normal(n, 51, 0, 1e-3) * (1.0 + oracle_mod.fill_uniform(n, 52, 0.0, 0.5))
    med = float(np.median(np.abs(acc)))
    blk = slice(n // 3, n // 3 + 40_000)                     # ~10 chunks whose every element is a candidate at ratio 0.5
    acc[blk] = med * (1.0 + 1e-4 * oracle_mod.fill_normal(40_000, 53))
    q8 = float(np.quantile(np.abs(acc), 0.2))                # ... and a block sitting at ratio 0.8
    blk2 = slice(2 * n // 3, 2 * n // 3 + 30_000)
    acc[blk2] = q8 * (1.0 + 1e-4 * oracle_mod.fill_normal(30_000, 54))
    _check_route1(ops, oracle_mod, acc, [n // 2], f"n={n} one ratio")
    if n <= 3_000_000:
        _check_route1(ops, oracle_mod, acc, [int(n * r / 10) for r in range(1, 11)], f"n={n} ten ratios")
    else:
        _check_route1(ops, oracle_mod, acc, [int(n * 0.8)], f"n={n} ratio 0.8")


def test_topk_zero_block_with_nans_falls_back_only_when_the_rank_reaches_them(ops, oracle_mod):
    n = 200_000
    acc = oracle_mod.fill_normal(n, 41, 0, 1e-3) * (1.0 + oracle_mod.fill_uniform(n, 42, 0.0, 0.5))
    acc[10_000:60_000] = 0.0
    acc[[3, 77_777, 150_000]] = np.nan
    nz = int(np.count_nonzero(acc[~np.isnan(acc)]))
    _check_route1(ops, oracle_mod, acc, [nz + 1000, n - 3, n])      # zeros only / every number / everything
    for k in (n - 2, n - 1):                                        # the k-th largest is a NaN: full scan, same rule
        d = dev(acc)
        got = ops.mask_topk(d, [k])[0].cpu().numpy()
        assert ops.mask_topk_status(d.device) == (2, 0)
        assert np.array_equal(got, oracle_mod.mask_topk(acc, [k])[0])


def test_topk_reference_ratio_list_on_the_bench_accumulator(ops, oracle_mod):
    """The accumulator bench.py builds (ResNet-18, Kaiming seed 1, one pass over the 4,500 forget samples) ranked for the
    reference's ten ratios 0.1 ... 1.0 (Classification/generate_mask.py:49-80) in one call: single-read route, bit-exact
    against the oracle, every threshold published."""
    import torch.nn as nn
    import bench
    from unlearn_saliency_amd.Classification.generate_mask import THRESHOLD_LIST, accumulate_saliency
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.norm import use_fused_bn
    device = torch.device("cuda", torch.cuda.current_device())
    model, forget_loader, _ = bench.build_workload(device, 0, 1, 256)
    use_salun_convs(model)
    use_fused_bn(model)
    acc = accumulate_saliency(forget_loader, model, nn.CrossEntropyLoss())
    assert acc.numel() == N18
    a = acc.cpu().numpy()
    zeros = int((a == 0).sum())
    print(f"bench accumulator: {zeros} exact zeros of {N18} ({100.0 * zeros / N18:.2f} %)")
    ks = [int(N18 * r) for r in THRESHOLD_LIST]
    assert ks[-1] == N18
    _check_route1(ops, oracle_mod, a, ks, "bench accumulator")
    # a ratio inside the zero block, if there is one
    if zeros > 1000:
        _check_route1(ops, oracle_mod, a, [N18 - zeros // 2, N18 - 1], "bench accumulator, zero block")


def test_topk_two_level_route_with_a_zero_block(ops):
    """2^27 + 5 elements, 60 % exact zeros, ratio 0.5 (the threshold is a zero): the two-level bracket lands on the zero
    key and the tie pass admits the first k - #nonzero zeros.  Size-independent properties (the oracle would sort 134 M
    values): popcount, every non-zero selected, the selected zeros are a prefix of the zeros in flat-index order."""
    n = (1 << 27) + 5
    d = ops.fill_normal(n, 321, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 322, 0.0, 0.5))
    d[ops.fill_uniform(n, 323, 0.0, 1.0) < 0.6] = 0.0
    nz = int((d != 0).sum().item())
    k = int(n * 0.5)
    assert nz < k
    m = ops.mask_topk(d, [k])[0]
    assert ops.mask_topk_status(d.device) == (1, 0)
    assert ops.mask_popcount(m) == k
    assert bool(m[d != 0].all())
    zsel = m[d == 0]
    assert int(zsel.sum().item()) == k - nz
    assert bool(zsel[:k - nz].all()) and not bool(zsel[k - nz:].any())
    assert ops.mask_topk_thresholds(d.device, 1)[0].item() == 0.0
    full = ops.mask_topk(d, [k], flags=FULL)[0]
    assert torch.equal(full, m)


def test_mask_format_roundtrip(ops, oracle_mod):
    n = 100_003
    m = (oracle_mod.fill_u8(n, 5) & 1).astype(np.uint8)
    d = dev(m)
    i64 = ops.mask_u8_to_i64(d)
    assert i64.dtype == torch.int64 and np.array_equal(i64.cpu().numpy(), m.astype(np.int64))
    back = ops.mask_i64_to_u8(i64 * 7)  # any non-zero -> 1
    assert np.array_equal(back.cpu().numpy(), m)
    assert ops.mask_popcount(d) == int(m.sum())


# ------------------------------------------------------------------------ K3+K4
def _sgd_inputs(oracle_mod, n, seed):
    p = oracle_mod.fill_normal(n, seed, 0, 0.05)
    g = oracle_mod.fill_normal(n, seed + 1, 0, 1e-3)
    buf = oracle_mod.fill_normal(n, seed + 2, 0, 1e-3)
    m = (oracle_mod.fill_u8(n, seed + 3) & 1).astype(np.uint8)
    return p, g, buf, m


@pytest.mark.parametrize("n", [1, 5, 4096, 16_385, 1_000_003])
@pytest.mark.parametrize("wd", [5e-4, 0.0])
@pytest.mark.parametrize("mu", [0.9, 0.0])
@pytest.mark.parametrize("masked", [True, False])
def test_masked_sgd_bit_exact(ops, oracle_mod, n, wd, mu, masked):
    p, g, buf, m = _sgd_inputs(oracle_mod, n, 10)
    if not masked:
        m = None
    buf[...] = 0
    dp, dg, db = dev(p), dev(g), dev(buf)
    dm = dev(m) if masked else None
    for step in range(3):
        first = step == 0
        ops.masked_sgd_step(dp, dg, db if mu else None, dm, 0.013, mu, wd, first)
        oracle_mod.masked_sgd_step(p, g, buf if mu else None, m, 0.013, mu, wd, first)
    assert np.array_equal(bits(dp.cpu().numpy()), bits(p))
    if mu:
        assert np.array_equal(bits(db.cpu().numpy()), bits(buf))


def test_masked_sgd_equals_reference_sequence(ops, oracle_mod):
    """Fused kernel == mask-multiply -> SGD -> restore (RL.py:11-34) incl. the invariants
    p[m==0] bit-identical to theta0 and buf[m==0] == 0."""
    n = 200_001
    p, g, buf, m = _sgd_inputs(oracle_mod, n, 20)
    buf[...] = 0
    theta0 = p.copy()
    p_ref, buf_ref = p.copy(), buf.copy()
    dp, db, dm = dev(p), dev(buf), dev(m)
    for step in range(4):
        g = oracle_mod.fill_normal(n, 100 + step, 0, 1e-3)
        ops.masked_sgd_step(dp, dev(g), db, dm, 0.013, 0.9, 5e-4, step == 0)
        oracle_mod.masked_sgd_step_reference(p_ref, g, buf_ref, m, theta0, 0.013, 0.9, 5e-4, step == 0)
    got_p, got_b = dp.cpu().numpy(), db.cpu().numpy()
    assert np.array_equal(bits(got_p), bits(p_ref))
    assert np.array_equal(bits(got_b[m == 1]), bits(buf_ref[m == 1]))
    assert np.array_equal(bits(got_p[m == 0]), bits(theta0[m == 0]))
    assert not got_b[m == 0].any() and not buf_ref[m == 0].any()


def test_masked_sgd_resnet18_size_properties(ops):
    """Full-size property check without the CPU oracle: frozen weights untouched, momentum zeroed."""
    p = ops.fill_normal(N18, 1, 0, 0.05)
    g = ops.fill_normal(N18, 2, 0, 1e-3)
    buf = torch.zeros(N18, device="cuda")
    acc = ops.fill_normal(N18, 3, 0, 1e-3)
    m = ops.mask_topk(acc, [N18 // 2])[0]
    p0 = p.clone()
    ops.masked_sgd_step(p, g, buf, m, 0.013, 0.9, 5e-4, True)
    frozen = ~m.bool()
    assert torch.equal(p[frozen], p0[frozen])
    assert not buf[frozen].any()
    sel = m.bool()
    want = (g.double() + 5e-4 * p0.double())[sel]  # d = g + wd*p
    assert torch.allclose(buf[sel].double(), want, rtol=1e-6, atol=1e-10)
    assert torch.allclose(p[sel].double(), (p0.double() - 0.013 * buf.double())[sel], rtol=1e-6, atol=1e-9)


# --------------------------------------------------------------------------- K5
@pytest.mark.parametrize("n", [1, 7, 4096, 1_000_003, ND])
def test_grad_sqnorm(ops, oracle_mod, n):
    g = oracle_mod.fill_normal(n, 31, 0, 1e-2)
    got = float(ops.grad_sqnorm(dev(g)).item())
    want = oracle_mod.grad_sqnorm(g)
    assert abs(got - want) <= 1e-6 * abs(want)
    # deterministic run-to-run
    assert got == float(ops.grad_sqnorm(dev(g)).item())


@pytest.mark.parametrize("n", [1, 6, 4096, 500_001])
@pytest.mark.parametrize("wd", [0.0, 1e-2])
@pytest.mark.parametrize("masked", [True, False])
def test_masked_adam_bit_exact(ops, oracle_mod, n, wd, masked):
    p = oracle_mod.fill_normal(n, 40, 0, 0.05)
    m1 = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    m = (oracle_mod.fill_u8(n, 43) & 1).astype(np.uint8) if masked else None
    dp, dm1, dv = dev(p), dev(m1), dev(v)
    dm = dev(m) if masked else None
    for step in range(1, 4):
        g = oracle_mod.fill_normal(n, 50 + step, 0, 1e-3)
        ops.masked_adam_step(dp, dev(g), dm1, dv, dm, 1e-4, 0.9, 0.999, 1e-8, wd, step, gscale=0.75)
        oracle_mod.masked_adam_step(p, g, m1, v, m, 0.75, 1e-4, 0.9, 0.999, 1e-8, wd, step)
    assert np.array_equal(bits(dp.cpu().numpy()), bits(p))
    assert np.array_equal(bits(dm1.cpu().numpy()), bits(m1))
    assert np.array_equal(bits(dv.cpu().numpy()), bits(v))


def test_masked_adam_with_a_device_resident_step_counter(ops, oracle_mod):
    """salun_adam_coefficients + salun_masked_adam_step_coef (whole-step HIP graphs: the host cannot pass t) against the
    host-scalar entry: bit-identical parameters and moments over four steps."""
    n = 100_003
    p0, g = oracle_mod.fill_normal(n, 1, 0, 0.05), oracle_mod.fill_normal(n, 2, 0, 1e-2)
    mask = (oracle_mod.fill_u8(n, 3) & 1).astype(np.uint8)
    a = [dev(p0.copy()), dev(np.zeros(n, np.float32)), dev(np.zeros(n, np.float32))]
    b = [dev(p0.copy()), dev(np.zeros(n, np.float32)), dev(np.zeros(n, np.float32))]
    dg, dm = dev(g), dev(mask)
    step_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    coef = torch.zeros(2, dtype=torch.float32, device="cuda")
    for step in range(1, 5):
        ops.masked_adam_step(a[0], dg, a[1], a[2], dm, 1e-4, 0.9, 0.999, 1e-8, 0.0, step)
        ops.adam_coefficients(step_dev, 1e-4, 0.9, 0.999, coef)
        ops.masked_adam_step_coef(b[0], dg, b[1], b[2], dm, coef, 0.9, 0.999, 1e-8, 0.0)
        for x, y in zip(a, b):
            assert torch.equal(x, y), step
    assert int(step_dev.item()) == 4


def test_masked_adam_device_clip_and_frozen_weights(ops, oracle_mod):
    """clip -> mask -> Adam order (runners/diffusion.py:582-593); masked-out weights never move."""
    n = 300_007
    p = oracle_mod.fill_normal(n, 60, 0, 0.05)
    p0 = p.copy()
    m1, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    m = (oracle_mod.fill_u8(n, 61) & 1).astype(np.uint8)
    dp, dm1, dv, dm = dev(p), dev(m1), dev(v), dev(m)
    for step in range(1, 4):
        g = oracle_mod.fill_normal(n, 70 + step, 0, 0.02)
        dg = dev(g)
        sq = ops.grad_sqnorm(dg)
        ops.masked_adam_step(dp, dg, dm1, dv, dm, 1e-4, 0.9, 0.999, 1e-8, 0.0, step, sqnorm=sq, max_norm=1.0)
        coef = oracle_mod.clip_coef(float(sq.item()), 1.0)
        assert coef < 1.0
        oracle_mod.masked_adam_step(p, g, m1, v, m, coef, 1e-4, 0.9, 0.999, 1e-8, 0.0, step)
    got = dp.cpu().numpy()
    assert np.array_equal(bits(got), bits(p))
    assert np.array_equal(bits(got[m == 0]), bits(p0[m == 0]))


# --------------------------------------------------------------------------- K6
def test_qsample_bit_exact(ops, oracle_mod):
    B, chw, T = 128, 3 * 32 * 32, 1000
    x0 = oracle_mod.fill_uniform(B * chw, 1, -1, 1).reshape(B, 3, 32, 32)
    e = oracle_mod.fill_normal(B * chw, 2).reshape(B, 3, 32, 32)
    betas = np.linspace(1e-4, 0.02, T, dtype=np.float64).astype(np.float32)
    ab = np.cumprod(1 - betas, dtype=np.float32)
    sa, sb = np.sqrt(ab), np.sqrt(1 - ab)
    t = (oracle_mod.fill_uniform(B, 3) * T).astype(np.int64)
    got = ops.qsample(dev(x0), dev(e), dev(sa), dev(sb), dev(t)).cpu().numpy()
    want = oracle_mod.qsample(x0, e, sa, sb, t)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("B,shape", [(128, (3, 32, 32)), (8, (4, 64, 64)), (3, (3, 5, 7)), (1, (1, 1, 1))])
def test_sqerr_loss_and_grad(ops, oracle_mod, B, shape):
    n = B * int(np.prod(shape))
    a = oracle_mod.fill_normal(n, 4).reshape((B,) + shape)
    b = oracle_mod.fill_normal(n, 5).reshape((B,) + shape)
    for coef in (1.0 / B, 1.0 / n):
        loss, per, d = ops.sqerr_loss(dev(a), dev(b), coef, want_per_sample=True)
        wl, wper, wd = oracle_mod.sqerr_loss(a, b, coef)
        assert abs(loss.item() - wl) <= 1e-6 * abs(wl)
        assert np.allclose(per.cpu().numpy(), wper, rtol=1e-6, atol=0)
        assert np.array_equal(bits(d.cpu().numpy()), bits(wd))


def test_eps_mse_autograd_matches_torch(ops):
    torch.manual_seed(0)
    e = torch.randn(16, 3, 32, 32, device="cuda")
    out = torch.randn(16, 3, 32, 32, device="cuda", requires_grad=True)
    loss = ops.eps_mse(e, out) * 0.5
    loss.backward()
    out2 = out.detach().clone().requires_grad_(True)
    ref = (e - out2).square().sum(dim=(1, 2, 3)).mean(dim=0) * 0.5
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert torch.allclose(out.grad, out2.grad, rtol=1e-5, atol=1e-7)
    pseudo = torch.randn_like(e)
    out3 = out.detach().clone().requires_grad_(True)
    l3 = ops.mse_loss(pseudo, out3)
    l3.backward()
    out4 = out.detach().clone().requires_grad_(True)
    r4 = torch.nn.MSELoss()(pseudo, out4)
    r4.backward()
    assert abs(l3.item() - r4.item()) <= 1e-5 * abs(r4.item())
    assert torch.allclose(out3.grad, out4.grad, rtol=1e-5, atol=1e-9)


# --------------------------------------------------------------------------- K7
@pytest.mark.parametrize("n", [1, 4097, 777_777])
def test_fim_square_accumulate(ops, oracle_mod, n):
    F = np.zeros(n, np.float32)
    dF = dev(F)
    for s in range(3):
        tmp = oracle_mod.fill_normal(n, 80 + s, 0, 0.1)
        dt = dev(tmp)
        ops.fim_square_accumulate(dF, dt, 5000)
        oracle_mod.fim_square_accumulate(F, tmp, 5000)
        assert not dt.any() and not tmp.any()
    assert np.array_equal(bits(dF.cpu().numpy()), bits(F))


# --------------------------------------------------------------------------- K0
def test_image_batch(ops, oracle_mod):
    num, H, W, C, B = 500, 32, 32, 3, 256
    data = oracle_mod.fill_u8(num * H * W * C, 1).reshape(num, H, W, C)
    rng = np.random.RandomState(0)
    idx = rng.randint(0, num, B)
    crop = rng.randint(0, 9, (B, 2)).astype(np.int32)
    flip = rng.randint(0, 2, B).astype(np.uint8)
    got = ops.image_batch(dev(data), dev(idx.astype(np.int64)), dev(crop), dev(flip), pad=4).cpu().numpy()
    want = oracle_mod.image_batch(data, idx, crop, flip, 4)
    assert np.array_equal(bits(got), bits(want))
    plain = ops.image_batch(dev(data), dev(idx.astype(np.int64))).cpu().numpy()
    assert np.array_equal(plain, data[idx].transpose(0, 3, 1, 2).astype(np.float32) / np.float32(255))


# ----------------------------------------------------------------------- errors
def test_cpu_tensors_are_rejected(ops):
    with pytest.raises(RuntimeError):
        ops.saliency_accumulate(torch.zeros(4), torch.zeros(4))
    with pytest.raises(TypeError):
        ops.saliency_accumulate(torch.zeros(4, device="cuda", dtype=torch.float64), torch.zeros(4, device="cuda"))
    with pytest.raises(ValueError):
        ops.mask_topk(torch.zeros(4, device="cuda"), list(range(17)))
