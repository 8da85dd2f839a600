"""CompVis (LDM) U-Net state_dict -> Diffusers `UNet2DConditionModel` key layout — the conversion the reference's
`savemodelDiffusers` performs on the unlearned model before it is evaluated with the Diffusers pipeline
(SD/train-scripts/convertModels.py:348-591, :1006-1128; only the U-Net is converted there too).

A rule table instead of the reference's path-renaming passes: every CompVis key is parsed into (section, block index,
sub-module, tail) and mapped directly.
    time_embed.{0,2}                      -> time_embedding.linear_{1,2}
    input_blocks.0.0                      -> conv_in
    input_blocks.i.0  (ResBlock)          -> down_blocks.b.resnets.l         b, l = divmod(i - 1, layers_per_block + 1)
    input_blocks.i.1  (SpatialTransformer)-> down_blocks.b.attentions.l
    input_blocks.i.0.op (Downsample)      -> down_blocks.b.downsamplers.0.conv
    middle_block.{0,1,2}                  -> mid_block.resnets.0 / attentions.0 / resnets.1
    output_blocks.i.0                     -> up_blocks.b.resnets.l           b, l = divmod(i, layers_per_block + 1)
    output_blocks.i.1 (transformer)       -> up_blocks.b.attentions.l
    output_blocks.i.{1|2}.conv (Upsample) -> up_blocks.b.upsamplers.0.conv
    out.{0,2}                             -> conv_norm_out / conv_out
  inside a ResBlock: in_layers.0 -> norm1, in_layers.2 -> conv1, emb_layers.1 -> time_emb_proj, out_layers.0 -> norm2,
  out_layers.3 -> conv2, skip_connection -> conv_shortcut; transformer sub-keys keep their names.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import torch

UNET_PREFIX = "model.diffusion_model."

_RES = {"in_layers.0": "norm1", "in_layers.2": "conv1", "emb_layers.1": "time_emb_proj", "out_layers.0": "norm2",
        "out_layers.3": "conv2", "skip_connection": "conv_shortcut"}


def _resnet_tail(tail: str) -> str:
    for old, new in _RES.items():
        if tail.startswith(old + "."):
            return new + tail[len(old):]
    raise KeyError(f"unexpected ResBlock key tail {tail!r}")


def _is_transformer(tail: str) -> bool:
    return tail.split(".")[0] in ("norm", "proj_in", "transformer_blocks", "proj_out")


def unet_key_to_diffusers(key: str, layers_per_block: int) -> str:
    parts = key.split(".")
    sec = parts[0]
    if sec == "time_embed":
        return f"time_embedding.linear_{ {'0': 1, '2': 2}[parts[1]] }." + ".".join(parts[2:])
    if sec == "out":
        return {"0": "conv_norm_out", "2": "conv_out"}[parts[1]] + "." + ".".join(parts[2:])
    if sec == "middle_block":
        sub, tail = int(parts[1]), ".".join(parts[2:])
        if sub == 1:
            return "mid_block.attentions.0." + tail
        return f"mid_block.resnets.{0 if sub == 0 else 1}." + _resnet_tail(tail)
    if sec == "input_blocks":
        i, sub, tail = int(parts[1]), int(parts[2]), ".".join(parts[3:])
        if i == 0:
            return "conv_in." + tail
        b, l = divmod(i - 1, layers_per_block + 1)
        if tail.startswith("op."):
            return f"down_blocks.{b}.downsamplers.0.conv." + tail[3:]
        if sub == 0:
            return f"down_blocks.{b}.resnets.{l}." + _resnet_tail(tail)
        return f"down_blocks.{b}.attentions.{l}." + tail
    if sec == "output_blocks":
        i, sub, tail = int(parts[1]), int(parts[2]), ".".join(parts[3:])
        b, l = divmod(i, layers_per_block + 1)
        if sub == 0:
            return f"up_blocks.{b}.resnets.{l}." + _resnet_tail(tail)
        if _is_transformer(tail):
            return f"up_blocks.{b}.attentions.{l}." + tail
        if tail.startswith("conv."):
            return f"up_blocks.{b}.upsamplers.0.conv." + tail[5:]
    raise KeyError(f"unexpected U-Net key {key!r}")


def convert_ldm_unet_checkpoint(checkpoint: Dict[str, torch.Tensor], layers_per_block: int = 2
                                ) -> "OrderedDict[str, torch.Tensor]":
    """`checkpoint`: a CompVis state_dict (keys `model.diffusion_model.*`; other entries are ignored) or a bare U-Net
    state_dict.  Returns the Diffusers-layout U-Net state_dict (tensors are shared, not copied)."""
    checkpoint = checkpoint.get("state_dict", checkpoint)
    has_prefix = any(k.startswith(UNET_PREFIX) for k in checkpoint)
    out = OrderedDict()
    for k, v in checkpoint.items():
        if has_prefix:
            if not k.startswith(UNET_PREFIX):
                continue
            k = k[len(UNET_PREFIX):]
        out[unet_key_to_diffusers(k, layers_per_block)] = v
    return out


def savemodelDiffusers(name, compvis_config_file=None, diffusers_config_file=None, device="cpu", layers_per_block=2):
    """models/{name}/{name}.pt (CompVis) -> models/{name}/{name with compvis->diffusers}.pt holding the converted
    U-Net state_dict, as the reference does (convertModels.py:1006-1128)."""
    src = f"models/{name}/{name}.pt"
    dst = f"models/{name}/{name.replace('compvis', 'diffusers')}.pt"
    ckpt = torch.load(src, map_location=device, weights_only=False)
    torch.save(convert_ldm_unet_checkpoint(ckpt, layers_per_block), dst)
    return dst
