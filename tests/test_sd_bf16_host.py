"""Host side of the bf16 SD configuration (no GPU): which convolutions of the U-Net go to the bf16 MFMA kernels (K11),
which stay on the fp32 kernels, that parameters / state_dict are untouched by the re-classing, and that the weight-pack
cache is keyed on the parameter epoch the fused optimizers bump."""
import torch

from fixtures import sd_tiny_config


def test_reclassing_covers_every_convolution_and_keeps_the_state_dict():
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    from unlearn_saliency_amd.conv_bf16 import SalunConv2dBF16, _Fp32Island, use_salun_convs_bf16
    with torch.device("meta"):
        m = UNetModel(**V1_UNET_CONFIG)
    names = [n for n, _ in m.named_parameters()]
    n16 = use_salun_convs_bf16(m)
    convs = [mod for mod in m.modules() if isinstance(mod, torch.nn.Conv2d)]
    assert all(isinstance(c, (SalunConv2dBF16, _Fp32Island)) for c in convs)
    islands = [c for c in convs if isinstance(c, _Fp32Island)]
    assert n16 == len(convs) - len(islands) == 96
    # only the 4-channel latent head and tail stay fp32
    assert sorted((c.in_channels, c.out_channels) for c in islands) == [(4, 320), (320, 4)]
    assert [n for n, _ in m.named_parameters()] == names


def test_pack_cache_key_follows_param_epoch():
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.conv_bf16 import SalunConv2dBF16
    calls = []
    real = ops.conv2d_bf16_pack
    ops.conv2d_bf16_pack = lambda w, out=None: calls.append(1) or torch.zeros(1)
    try:
        c = torch.nn.Conv2d(32, 32, 3, padding=1)
        c.__class__ = SalunConv2dBF16
        c.packed_weight(); c.packed_weight()
        assert len(calls) == 1                      # cached
        ops.PARAM_EPOCH[0] += 1                     # what masked_sgd_step / masked_adam_step / proximal_step do
        c.packed_weight()
        assert len(calls) == 2
        with torch.no_grad():
            c.weight.mul_(2.0)                      # torch's own version counter
        c.packed_weight()
        assert len(calls) == 3
    finally:
        ops.conv2d_bf16_pack = real


def test_tiny_config_is_inside_the_bf16_kernels_domain():
    from unlearn_saliency_amd.SD.unet import UNetModel
    from unlearn_saliency_amd.conv_bf16 import use_salun_convs_bf16
    m = UNetModel(**sd_tiny_config())
    assert use_salun_convs_bf16(m) >= 8


def test_bf16_kernels_refuse_host_tensors_loudly():
    """No CPU path behind K11-K14: host tensors are an error, not a fallback."""
    import pytest
    from unlearn_saliency_amd import norm, ops
    x = torch.randn(1, 8, 8, 32).to(torch.bfloat16)
    wp = torch.zeros(32, 9, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv2d_bf16_forward(x, wp, 3, 1, 1)
    q = torch.randn(1, 16, 2, 40).to(torch.bfloat16)
    with pytest.raises(TypeError, match="device tensor"):
        ops.attention(q, q, q, 40 ** -0.5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layer_norm_bf16(torch.randn(4, 32).to(torch.bfloat16), torch.nn.LayerNorm(32))
    # a bf16 host tensor through the GroupNorm dispatcher takes the library ops (is_cuda gate), it never reaches K12
    gn = torch.nn.GroupNorm(4, 32)
    y = norm.fused_gn_act(torch.randn(2, 32, 4, 4), gn, silu=True)
    assert y.shape == (2, 32, 4, 4)


def test_attention_view_contract():
    """[B, tokens, H, D] views: heads must be adjacent runs of D contiguous channels (checked before any launch)."""
    import pytest
    from unlearn_saliency_amd import ops

    class FakeCuda(torch.Tensor):
        @property
        def is_cuda(self):
            return True

    t = torch.randn(2, 6, 4, 8).to(torch.bfloat16)
    bad = t.transpose(1, 2).as_subclass(FakeCuda)  # [B, H, tokens, D]: head stride is not D
    with pytest.raises(ValueError, match="adjacent runs"):
        ops._tok_view(bad, "q")
    ok = t.as_subclass(FakeCuda)
    ptr, bs, ld = ops._tok_view(ok, "q")
    assert bs.value == 6 * 4 * 8 and ld.value == 4 * 8


def test_pack_cache_sees_in_place_writes_on_the_flat_arena():
    """FlatArena binds parameters with `p.data = flat[o:o+k].view(...)`; a torch write on `arena.params`
    (a model reset, a copy from a checkpoint vector) does not bump the parameter's own version counter — the pack cache
    must key on the flat vector's counter as well, or forward / backward-data keep a stale bf16 weight image."""
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.conv_bf16 import SalunConv2dBF16
    from unlearn_saliency_amd.flat import FlatArena
    calls = []
    real = ops.conv2d_bf16_pack
    ops.conv2d_bf16_pack = lambda w, out=None: calls.append(1) or torch.zeros(1)
    try:
        c = torch.nn.Conv2d(32, 32, 3, padding=1)
        c.__class__ = SalunConv2dBF16
        arena = FlatArena.from_module(c, device="cpu")
        c.packed_weight(); c.packed_weight()
        assert len(calls) == 1
        v = c.weight._version
        with torch.no_grad():
            arena.params.mul_(0.5)                  # in place on the flat vector
        assert c.weight._version == v               # ... which the parameter's own counter does not see
        c.packed_weight()
        assert len(calls) == 2
        with torch.no_grad():
            arena.params[3:7].zero_()               # a slice shares the flat vector's counter
        c.packed_weight()
        assert len(calls) == 3
        c.packed_weight()
        assert len(calls) == 3
    finally:
        ops.conv2d_bf16_pack = real


def test_linear_layers_of_the_transformer_blocks_are_reclassed():
    """Round 3: the transformer blocks' Linear layers go to the K11 1x1 kernels in the bf16 configuration — 10 per block
    (attn1 / attn2: to_q, to_k, to_v, to_out; the GEGLU and output projections) x 16 blocks; the time-embedding Linears
    stay; names / state_dict untouched; on the host (fp32, no autocast) the module is plain F.linear."""
    from unlearn_saliency_amd.SD.unet import UNetModel, V1_UNET_CONFIG
    from unlearn_saliency_amd.conv_bf16 import SalunLinearBF16, use_salun_linears_bf16
    with torch.device("meta"):
        m = UNetModel(**V1_UNET_CONFIG)
    names = [n for n, _ in m.named_parameters()]
    assert use_salun_linears_bf16(m) == 160
    assert [n for n, _ in m.named_parameters()] == names
    assert not isinstance(m.time_embed[0], SalunLinearBF16)
    lin = torch.nn.Linear(64, 96)
    ref = lin(torch.ones(3, 5, 64))
    lin.__class__ = SalunLinearBF16
    assert torch.equal(lin(torch.ones(3, 5, 64)), ref)
    assert torch.equal(lin(torch.ones(3, 5, 64), addend=torch.ones(3, 5, 96)), ref + 1)


def test_forget_and_target_on_host_tensors_is_the_plain_sequence():
    """`train_scripts.forget_and_target` (nsfw_removal.py:131-140 of the reference: forget pass, then the target pass
    under no_grad) only uses a second stream for device tensors; on the host it is the reference's two calls in the
    reference's order."""
    import torch
    from unlearn_saliency_amd.SD import train_scripts as TS
    calls = []

    class Stub:
        def apply_model(self, z, t, c):
            calls.append((c, torch.is_grad_enabled()))
            return z * 2.0 + c.mean()

    z = torch.randn(2, 4, 8, 8, requires_grad=True)
    t = torch.tensor([1, 2])
    c_f, c_t = torch.ones(2, 3), torch.zeros(2, 3)
    out, tgt = TS.forget_and_target(Stub(), z, t, c_f, c_t)
    assert [g for _, g in calls] == [True, False] and calls[0][0] is c_f and calls[1][0] is c_t
    assert out.requires_grad and not tgt.requires_grad
    assert torch.equal(out.detach(), z.detach() * 2.0 + 1.0) and torch.equal(tgt, z.detach() * 2.0)


def test_registered_modules_repack_together_once_per_parameter_epoch():
    """Round 6: after an optimizer step every weight image of a model is stale; the first one asked for re-packs ALL of
    them in one batch (conv_bf16._repack_stale -> ops.bf16_pack_batch), later requests are cache hits; a second model's
    modules on another device are not touched; a torch write on one weight re-packs that weight alone."""
    from unlearn_saliency_amd import conv_bf16, ops
    from unlearn_saliency_amd.SD.unet import UNetModel
    m = UNetModel(**sd_tiny_config())
    n_conv = conv_bf16.use_salun_convs_bf16(m)
    n_lin = conv_bf16.use_salun_linears_bf16(m)
    mods = [x for x in m.modules() if isinstance(x, (conv_bf16.SalunConv2dBF16, conv_bf16.SalunLinearBF16))]
    assert len(mods) == n_conv + n_lin and all(getattr(x, "_salun_pack_registered", False) for x in mods)
    batches = []
    real = ops.bf16_pack_batch
    ops.bf16_pack_batch = lambda jobs: batches.append(list(jobs)) or 1
    try:
        first = mods[0].packed_weight()
        assert len(batches) == 1
        mine = [j for j in batches[0] if any(j[0].data_ptr() == x.weight.data_ptr() for x in mods)]
        assert len(mine) == n_conv + 2 * n_lin          # every Linear: the [K, C] image and the transposed one
        assert all(j[5] in (False, True) and j[1].dtype == torch.bfloat16 for j in mine)
        for x in mods:                                   # all fresh now: no further batch
            x.packed_weight()
            if isinstance(x, conv_bf16.SalunLinearBF16):
                assert tuple(x.packed_weight_t().shape) == (x.in_features, x.out_features)
        assert len(batches) == 1 and mods[0].packed_weight() is first
        ops.PARAM_EPOCH[0] += 1                          # an optimizer step through raw pointers
        mods[3].packed_weight()
        assert len(batches) == 2 and len([j for j in batches[1] if any(j[0].data_ptr() == x.weight.data_ptr() for x in mods)]) == len(mine)
        assert mods[0].packed_weight() is first          # buffers are reused
        with torch.no_grad():
            mods[1].weight.mul_(2.0)                     # torch's own version counter: this weight alone
        mods[1].packed_weight()
        assert len(batches) == 3
        assert [j[0].data_ptr() for j in batches[2]].count(mods[1].weight.data_ptr()) == len(batches[2])
    finally:
        ops.bf16_pack_batch = real
