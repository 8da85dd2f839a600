"""Stable-Diffusion v1 (LDM) U-Net, state_dict-compatible with the reference's `UNetModel`
(SD/ldm/modules/diffusionmodules/openaimodel.py:428-847 with SD/ldm/modules/attention.py:149-303),
restricted to what `configs/stable-diffusion/v1-inference.yaml` instantiates: 2-D, spatial transformers
with cross-attention (context 768), `legacy=False`, GEGLU feed-forward, GroupNorm(32) evaluated in fp32.

Built from a block table so `named_parameters()` yields the reference's 686 names / shapes / order
(`time_embed.0.weight`, `input_blocks.1.1.transformer_blocks.0.attn2.to_q.weight`, …): that order is the flat
index of the saliency ranking and the key set of the mask files.

MI355X notes: every attention (4096 / 1024 / 256 / 64 tokens × 8 heads, cross-attention over 77 tokens)
is one `scaled_dot_product_attention` call — the reference's einsum+softmax materialises a
(B·8)×4096×4096 fp32 score tensor per block — and the model runs under bf16 autocast
with fp32 master weights in the flat arena for the 8-GPU config (BASELINE.json configs[4]).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint as _ckpt

from .. import ops
from ..conv_bf16 import SalunConv2dBF16, SalunLinearBF16
from ..norm import _GN_TYPES, fused_gn_act


def timestep_embedding(timesteps, dim, max_period=10000):
    """[cos | sin] sinusoidal embedding (note: cosine first, unlike the DDPM model)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        if x.dtype == torch.float64:  # a float64 evaluation of the module (parity tests) stays in float64
            return super().forward(x)
        return super().forward(x.float()).type(x.dtype)


_GN_TYPES.add(GroupNorm32)  # fp32 evaluation is what the fused kernel does anyway


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class TimestepEmbedSequential(nn.Sequential):
    """Children get (x, emb), (x, context) or (x) depending on their kind."""

    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, ResBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode="nearest"))


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


import os as _os
_EMB_FP32 = _os.environ.get("SALUN_SD_EMB_FP32", "1") != "0"   # A/B switch: 0 = Linear(SiLU(emb)) under autocast per block


def _emb_term(lin, emb):
    """`emb_layers(emb)` = Linear(SiLU(emb)) of a ResBlock for the bf16 kernels, whose epilogue takes it in fp32: computed in
    fp32 outside autocast — 8 rows; under autocast every call cast the fp32 weight, the bias and the result (three copy
    launches around a 3 us product, x 113 calls per SD step) — from ONE SiLU(emb) per U-Net pass (`UNetModel.forward`
    leaves it on the tensor; the 22 ResBlocks of a pass and their checkpoint recomputes share it)."""
    if not _EMB_FP32:
        return lin(F.silu(emb)).float()
    with torch.autocast("cuda", enabled=False):
        act = getattr(emb, "_salun_silu", None)
        if act is None:
            act = F.silu(emb.float())
        return F.linear(act, lin.weight, lin.bias)


class ResBlock(nn.Module):
    """GN-SiLU-conv3x3, + Linear(SiLU(emb)), GN-SiLU-dropout-conv3x3(zero-init), 1x1 skip if widths differ."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_checkpoint=False):
        super().__init__()
        out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.preserve_rng = dropout > 0  # no random draw inside the block otherwise: nothing to replay on recompute
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(nn.Conv2d(out_channels, out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if out_channels == channels else nn.Conv2d(channels, out_channels, 1)

    def _forward(self, x, emb):
        # GroupNorm32 -> SiLU as one kernel (norm.fused_gn_act; falls back to the library ops under autocast / for
        # shapes outside the kernel's domain); the remaining layers of each Sequential run as they are
        conv1, conv2 = self.in_layers[2], self.out_layers[3]
        if isinstance(conv1, SalunConv2dBF16) and isinstance(conv2, SalunConv2dBF16):
            # bf16 kernels: the time-embedding term and the residual branch ride in the convolutions' epilogues
            h = conv1(fused_gn_act(x, self.in_layers[0], silu=True), nbias=_emb_term(self.emb_layers[1], emb))
            o = fused_gn_act(h, self.out_layers[0], silu=True)
            return conv2(self.out_layers[2](o), addend=self.skip_connection(x))
        h = conv1(fused_gn_act(x, self.in_layers[0], silu=True))
        h = h + self.emb_layers(emb).type(h.dtype)[..., None, None]
        o = fused_gn_act(h, self.out_layers[0], silu=True)
        return self.skip_connection(x) + conv2(self.out_layers[2](o))

    def forward(self, x, emb):
        if self.use_checkpoint and torch.is_grad_enabled():
            return _ckpt(self._forward, x, emb, use_reentrant=False, preserve_rng_state=self.preserve_rng)
        return self._forward(x, emb)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head, self.scale = heads, dim_head, dim_head ** -0.5
        self.own_gemm = False  # set by conv.use_salun_convs(model) (fp32 configuration): attention on K15
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    def _out(self, o, residual):
        """to_out = Linear -> Dropout(0); with the own bf16 GEMM the block's residual rides in the Linear's epilogue."""
        lin = self.to_out[0]
        if residual is not None and isinstance(lin, SalunLinearBF16) and o.dtype == torch.bfloat16 \
                and residual.dtype == torch.bfloat16:
            return self.to_out[1](lin(o, addend=residual))
        y = self.to_out(o)
        return y if residual is None else y + residual

    def forward(self, x, context=None, residual=None):
        """`residual`: the tensor the caller adds to the result (BasicTransformerBlock's `attn(x) + x`)."""
        context = x if context is None else context
        b, n, _ = x.shape
        q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
        if q.is_cuda and q.dtype == torch.bfloat16 and ops.attn_supported(self.dim_head):
            # bf16 configuration: fused attention of csrc/salun_attn.hip reading the projections in place ([b, n, h, d] views)
            heads = lambda t: t.view(b, t.shape[1], self.heads, self.dim_head)
            o = ops.attention(heads(q), heads(k.to(torch.bfloat16)), heads(v.to(torch.bfloat16)), self.scale)
            return self._out(o.view(b, n, self.heads * self.dim_head), residual)
        split = lambda t: t.view(b, t.shape[1], self.heads, self.dim_head).transpose(1, 2)  # (b, h, tokens, d)
        if self.own_gemm:
            from .. import gemm
            qs, ks, vs = split(q), split(k), split(v)
            if gemm.attention_supported(qs, ks, vs):
                # fp32 configuration: GEMM -> row softmax -> GEMM on the fp32 matrix-core kernel over the [b, h, n, d]
                # views of the projections (no head-splitting copies; o comes back in q's layout)
                o = gemm.attention_f32(qs, ks, vs, self.scale)
                return self._out(o.transpose(1, 2).reshape(b, n, self.heads * self.dim_head), residual)
        o = F.scaled_dot_product_attention(split(q), split(k), split(v), scale=self.scale)
        return self._out(o.transpose(1, 2).reshape(b, n, self.heads * self.dim_head), residual)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h = self.proj(x)
        if h.is_cuda and h.dtype == torch.bfloat16 and h.shape[-1] % 16 == 0:
            return ops.geglu_bf16(h)  # one pass (csrc/salun_tok_bf16.hip) instead of chunk -> gelu -> mul
        x, gate = h.chunk(2, dim=-1)
        return x * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, dropout=0.0):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(dropout), nn.Linear(dim * mult, dim))

    def forward(self, x, residual=None):
        """`residual`: added to the result — in the output Linear's epilogue when that runs on the own bf16 GEMM."""
        h = self.net[1](self.net[0](x))
        lin = self.net[2]
        if residual is not None and isinstance(lin, SalunLinearBF16) and h.dtype == torch.bfloat16 \
                and residual.dtype == torch.bfloat16:
            return lin(h, addend=residual)
        y = lin(h)
        return y if residual is None else y + residual


class BasicTransformerBlock(nn.Module):
    """self-attention, cross-attention on the text context, GEGLU feed-forward; pre-LayerNorm residuals."""

    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, use_checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head, dropout)
        self.ff = FeedForward(dim, dropout=dropout)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.use_checkpoint = use_checkpoint
        self.preserve_rng = dropout > 0

    @staticmethod
    def _ln(ln, x):
        # bf16 tokens: LayerNorm in one pass with bf16 output (autocast would cast to fp32, normalise, and leave an fp32
        # tensor for the next Linear to cast back)
        if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048:
            return ops.layer_norm_bf16(x, ln)
        return ln(x)

    def _forward(self, x, context):
        x = self.attn1(self._ln(self.norm1, x), residual=x)
        x = self.attn2(self._ln(self.norm2, x), context, residual=x)
        return self.ff(self._ln(self.norm3, x), residual=x)

    def forward(self, x, context=None):
        if self.use_checkpoint and torch.is_grad_enabled():
            return _ckpt(self._forward, x, context, use_reentrant=False, preserve_rng_state=self.preserve_rng)
        return self._forward(x, context)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, use_checkpoint=True):
        super().__init__()
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim, use_checkpoint) for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, 1))

    def forward(self, x, context=None):
        b, c, h, w = x.shape
        t = self.proj_in(fused_gn_act(x, self.norm, silu=False)).flatten(2).transpose(1, 2).contiguous()  # b (h w) c
        for blk in self.transformer_blocks:
            t = blk(t, context)
        t = t.transpose(1, 2).reshape(b, -1, h, w)  # channels_last view of the token tensor
        if isinstance(self.proj_out, SalunConv2dBF16):  # NHWC kernels: no copy, the residual rides in the epilogue
            return self.proj_out(t, addend=x)
        return self.proj_out(t.contiguous()) + x


class UNetModel(nn.Module):
    def __init__(self, image_size=32, in_channels=4, model_channels=320, out_channels=4, num_res_blocks=2,
                 attention_resolutions=(4, 2, 1), dropout=0.0, channel_mult=(1, 2, 4, 4), num_heads=8,
                 use_spatial_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=False,
                 legacy=False, **unused):
        super().__init__()
        if not use_spatial_transformer or legacy:
            raise NotImplementedError("only the v1-inference.yaml variant (spatial transformer, legacy=False) is in scope")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.image_size, self.context_dim = image_size, context_dim
        self.use_checkpoint = use_checkpoint
        ted = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))

        def res(cin, cout):
            return ResBlock(cin, ted, dropout, out_channels=cout, use_checkpoint=use_checkpoint)

        def attn(ch):
            return SpatialTransformer(ch, num_heads, ch // num_heads, depth=transformer_depth, context_dim=context_dim,
                                      use_checkpoint=use_checkpoint)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        skip_chans, ch, ds = [model_channels], model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers.append(attn(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch), res(ch, ch))
        self.output_blocks = nn.ModuleList()
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                layers = [res(ch + skip_chans.pop(), model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers.append(attn(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(GroupNorm32(32, ch), nn.SiLU(),
                                 zero_module(nn.Conv2d(model_channels, out_channels, 3, padding=1)))

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        emb = self.time_embed(timestep_embedding(timesteps, self.model_channels).to(self.time_embed[0].weight.dtype))
        if _EMB_FP32 and emb.is_cuda and isinstance(self.middle_block[0].in_layers[2], SalunConv2dBF16):
            with torch.autocast("cuda", enabled=False):
                emb._salun_silu = F.silu(emb.float())  # shared by every ResBlock of this pass (_emb_term)
        hs, h = [], x
        for module in self.input_blocks:
            h = module(h, emb, context)
            hs.append(h)
        h = self.middle_block(h, emb, context)
        for module in self.output_blocks:
            h = module(torch.cat([h, hs.pop()], dim=1), emb, context)
        return self.out[2](fused_gn_act(h.type(x.dtype), self.out[0], silu=True))


def set_activation_checkpointing(model: nn.Module, enabled: bool) -> int:
    """Flip `use_checkpoint` on every block of a built U-Net; returns how many blocks changed.  The reference's
    v1-inference.yaml turns activation checkpointing on (`use_checkpoint: True`,
    SD/ldm/modules/diffusionmodules/openaimodel.py:245-247, attention.py:209-210): every ResBlock / transformer block is
    run a second time inside the backward pass so that its activations need not be kept — a device of 24 – 80 GB-class
    accelerators.  At batch 8 the whole step keeps ~50 GB of activations; with 288 GB of HBM they can simply stay
    resident (`enabled = False`): same kernels on the same inputs in the same order for everything that is kept,
    bit-identical parameters after the step (tests/test_sd_gpu.py), two of the nine forward-equivalents of a
    nsfw_removal step gone."""
    n = 0
    for m in model.modules():
        if isinstance(m, (ResBlock, BasicTransformerBlock, SpatialTransformer, UNetModel)) and hasattr(m, "use_checkpoint"):
            if bool(m.use_checkpoint) != bool(enabled):
                m.use_checkpoint = bool(enabled)
                n += 1
    return n


V1_UNET_CONFIG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1),
                      num_res_blocks=2, channel_mult=(1, 2, 4, 4), num_heads=8, use_spatial_transformer=True,
                      transformer_depth=1, context_dim=768, use_checkpoint=True, legacy=False)
