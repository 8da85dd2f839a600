"""SalUn for Stable Diffusion: the four functions of the reference's SD/train-scripts —

    generate_mask / generate_nsfw_mask      SD/train-scripts/generate_mask.py:8-108, :111-211
    certain_label                           SD/train-scripts/random_label.py:13-156
    nsfw_removal                            SD/train-scripts/nsfw_removal.py:33-175
    proximal_gradient                       SD/train-scripts/proximal_gradient.py:14-186

— with the reference's positional parameters.  Data loaders yield latents + context embeddings
(ldm_lite.py explains why); `model=` / `*_dl=` keyword arguments inject a prepared model and loaders.

Differences underneath: gradients stay on the device in one flat accumulator (the reference copies 3.4 GB of
gradients to the CPU per iteration, generate_mask.py:66-69), the 859.5 M-element top-k is a radix select on the
GPU (the reference argsorts twice on the CPU), the mask is a resident 0.86 GB u8 vector instead of a 6.9 GB
int64 dict uploaded every step (random_label.py:132-137), Adam is the fused masked kernel, and the
`sleep(0.1)` per step (random_label.py:142) is not reproduced.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import yaml

from .. import dist as sdist
from .. import draws, hostperf, ops
from ..flat import FlatArena
from ..optim import FusedMaskedAdam
from ..streams import concurrent_stream
from .ldm_lite import SD_V1_FROZEN_PARAMS, LatentDiffusionLite  # noqa: F401 (re-exported)
from .unet import V1_UNET_CONFIG

UNET_PREFIX = "model.diffusion_model."

def setup_model(config_path, ckpt_path, device, bf16=False, resident_activations=False) -> LatentDiffusionLite:
    """YAML (`model.params.unet_config.params`, e.g. configs/stable-diffusion/v1-inference.yaml) + optional CompVis
    checkpoint (`state_dict` with `model.diffusion_model.*` keys).  No OmegaConf needed.  `resident_activations`: ignore
    the config's `use_checkpoint` and keep every activation in HBM (unet.set_activation_checkpointing)."""
    cfg = dict(V1_UNET_CONFIG)
    if config_path and os.path.exists(config_path):
        with open(config_path) as f:
            params = yaml.safe_load(f)["model"]["params"]["unet_config"]["params"]
        cfg.update({k: (tuple(v) if isinstance(v, list) else v) for k, v in params.items()})
    model = LatentDiffusionLite(cfg, bf16=bf16).to(device)
    if ckpt_path and os.path.exists(ckpt_path):
        sd = torch.load(ckpt_path, map_location=device, weights_only=False)
        sd = sd.get("state_dict", sd)
        unet_sd = {k[len(UNET_PREFIX):]: v for k, v in sd.items() if k.startswith(UNET_PREFIX)}
        model.model.diffusion_model.load_state_dict(unet_sd, strict=True)
    if torch.device(device).type == "cuda":
        model.use_mfma_convs()
    if resident_activations:
        from .unet import set_activation_checkpointing
        set_activation_checkpointing(model.model.diffusion_model, False)
    return model


class ShardedBatches:
    """Data parallel over GLOBAL batches: an iterable of tuples of tensors (batch in dimension 0) -> this rank's
    contiguous balanced shard of each, with `last_shard = (lo, hi, b)` for the draws (draws.py: timesteps and noise are
    drawn for the global batch and sliced) and the loss weights.  The reference's scripts are single-GPU; a plain list
    of per-rank batches (what `--synthetic` builds) keeps meaning "every rank has its own batches"."""

    def __init__(self, batches, rank: Optional[int] = None, world_size: Optional[int] = None):
        self.batches = list(batches)
        self.rank = sdist.rank() if rank is None else rank
        self.world_size = sdist.world_size() if world_size is None else world_size
        self.last_shard = None

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for batch in self.batches:
            b = batch[0].shape[0]
            lo, hi = sdist.balanced_slice(b, self.rank, self.world_size)
            self.last_shard = (lo, hi, b)
            yield tuple(t[lo:hi] for t in batch)


def _unet_arena(model) -> FlatArena:
    a = getattr(model, "_salun_unet_arena", None)
    if a is None:
        a = FlatArena(model.model.diffusion_model.named_parameters())  # keys relative to the U-Net, as in the masks
        object.__setattr__(model, "_salun_unet_arena", a)
    return a


def _saliency_mask(model, batches, c_guidance, mask_path, ratio=0.5):
    """Shared body of generate_mask / generate_nsfw_mask: loss = -MSE(noise, (1+g)·eps(z_t,c) - g·eps(z_t,∅)),
    uniform t, no clipping; Σ grads -> |.| -> top-`ratio` -> int64 dict saved as with_{ratio}.pt."""
    arena = _unet_arena(model)
    acc = arena.new_like()
    model.eval()
    hostperf.freeze_gc()
    for z, c_forget, c_null in batches:
        z, c_forget, c_null = z.to(model.device), c_forget.to(model.device), c_null.to(model.device)
        sh = draws.shard_of(batches, z.shape[0])
        with draws.scope(sh):
            t = draws.randint(model.num_timesteps, z.shape[0], model.device).long()
            noise = draws.randn_like(z)
            z_noisy = model.q_sample(x_start=z, t=t, noise=noise)
            forget_out = model.apply_model(z_noisy, t, c_forget)
            null_out = model.apply_model(z_noisy, t, c_null)
        preds = (1 + c_guidance) * forget_out - c_guidance * null_out
        loss = -ops.mse_loss(noise, preds)
        if sh.share != 1.0:  # a shard of a global batch: its share of the global-batch mean (accumulators are SUMmed)
            loss = loss * sh.share
        arena.zero_grad()
        loss.backward()
        ops.saliency_accumulate(acc, arena.grads, 1.0)
    sdist.all_reduce_sum_(acc)
    object.__setattr__(model, "_salun_last_saliency", acc)  # sum of gradients before |.| (diagnostics / parity tests)
    mask = ops.mask_topk(acc, [int(arena.n * ratio)], check=True)[0]  # raises instead of saving a garbage mask
    if mask_path and sdist.rank() == 0:
        os.makedirs(mask_path, exist_ok=True)
        torch.save(arena.unpack_mask(mask), os.path.join(mask_path, f"with_{str(ratio)}.pt"))
    return mask


def generate_mask(classes, c_guidance, batch_size, epochs, lr, config_path, ckpt_path, diffusers_config_path, device,
                  image_size=512, num_timesteps=1000, *, model=None, forget_dl=None):
    model = model or setup_model(config_path, ckpt_path, device)
    if forget_dl is None:
        raise ValueError("forget_dl: iterable of (latents, class-prompt context, empty-prompt context) batches")
    return _saliency_mask(model, forget_dl, c_guidance, os.path.join("mask", str(classes)))


def generate_nsfw_mask(c_guidance, batch_size, epochs, lr, config_path, ckpt_path, diffusers_config_path, device,
                       image_size=512, num_timesteps=1000, *, model=None, forget_dl=None):
    model = model or setup_model(config_path, ckpt_path, device)
    if forget_dl is None:
        raise ValueError("forget_dl: iterable of (latents, 'a photo of a nude person' context, empty context) batches")
    mask = _saliency_mask(model, forget_dl, c_guidance, None)
    if sdist.rank() == 0:
        os.makedirs("mask", exist_ok=True)
        torch.save(_unet_arena(model).unpack_mask(mask), os.path.join("mask", "nude_0.5.pt"))
    return mask


# SALUN_SD_TARGET_OVERLAP=0: run the no-grad target pass on the main stream, after the forget pass
TARGET_OVERLAP = os.environ.get("SALUN_SD_TARGET_OVERLAP", "1") != "0"
_target_streams: dict = {}


def forget_and_target(model, z_noisy, t, c_forget, c_target):
    """`forget_out = eps(z_t, t, c_forget)` (differentiated) and `target = eps(z_t, t, c_target)` under no_grad — the
    two U-Net passes of nsfw_removal.py:131-140 / random_label.py:112-121.  They depend on nothing of each other, and
    at batch 8 most of the U-Net's kernels leave the chip partly empty, so on the device the target pass is issued on a
    second stream beside the forget pass.  Host call order (forget first) is the reference's.  Safe only while no
    cached weight image is (re)written during the two passes: the packed bf16 weights are produced by the first pass
    after an optimizer step — the remain pass, in every loop of this file; if the forget pass had to pack anything
    (cold cache) the target pass is issued on the main stream after it instead."""
    dev = z_noisy.device
    # (Also under data parallel.  While the gradient slices' collectives had a launch stream of their own, this second
    # stream made the data-parallel step slower — 174.8 ms with it, 167.4 without — and was switched off there; with the
    # collectives issued from the backward-weight stream it gains what it gains single-process: 164.4 -> 158.8 ms.)
    if not (TARGET_OVERLAP and z_noisy.is_cuda):
        out = model.apply_model(z_noisy, t, c_forget)
        with torch.no_grad():
            return out, model.apply_model(z_noisy, t, c_target)
    side = _target_streams.get(dev)
    if side is None:
        side = _target_streams[dev] = concurrent_stream(dev)
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    packs = ops.PACK_CALLS[0]
    out = model.apply_model(z_noisy, t, c_forget)
    if ops.PACK_CALLS[0] != packs:
        with torch.no_grad():
            return out, model.apply_model(z_noisy, t, c_target)
    with torch.no_grad(), torch.cuda.stream(side):
        target = model.apply_model(z_noisy, t, c_target)
    for v in (z_noisy, t, c_target):
        v.record_stream(side)
    main.wait_stream(side)
    target.record_stream(main)
    return out, target


def _trainable_mask(arena: FlatArena, train_method: str) -> Optional[torch.Tensor]:
    """"xattn": only parameters whose name contains attn2 are optimised (random_label.py:47-54).  Expressed as a
    u8 mask over the flat arena so the same fused kernel serves both methods."""
    if train_method == "full":
        return None
    if train_method != "xattn":
        raise ValueError(f"train_method {train_method!r} (full | xattn)")
    m = torch.zeros(arena.n, dtype=torch.uint8, device=arena.device)
    for name, off, k in zip(arena.names, arena.offsets, arena.numels):
        if "attn2" in name:
            m[off:off + k] = 1
    return m


def _unlearn(model, forget_dl, remain_dl, alpha, epochs, lr, mask_path, train_method, proximal_ratio=None):
    """Loop body shared by certain_label / nsfw_removal / proximal_gradient: forget batches (z, c_forget, c_pseudo),
    remain batches (z, c); loss = MSE(eps(z_t^f, c_forget), eps(z_t^f, c_pseudo).detach()) + alpha * LDM-loss(remain).
    `proximal_ratio` (proximal_gradient.py's `mask_ratio`): after every optimizer step the weights are soft-thresholded
    towards their initial values so that `ratio_t` of them are reset exactly (one diff + select + threshold pass over
    the flat arena, K9)."""
    arena = _unet_arena(model)
    hostperf.freeze_gc()
    init_params = arena.params.clone() if proximal_ratio is not None else None
    # the reference ranks over the whole LatentDiffusion: U-Net + the frozen first stage and text encoder
    n_frozen = int(getattr(model, "frozen_param_count", 0)) if proximal_ratio is not None else 0
    n_total = arena.n + n_frozen
    if proximal_ratio is not None:
        print(f"proximal ranking over {n_total} parameters: {arena.n} U-Net + {n_frozen} frozen (first stage + text encoder)")
    steps_per_epoch = len(forget_dl) + len(remain_dl) if proximal_ratio is not None else 0  # proximal_gradient.py:73
    total_steps = epochs * steps_per_epoch
    opt = FusedMaskedAdam(arena, lr=lr)  # torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no clipping
    mask_u8 = _trainable_mask(arena, train_method)
    if mask_path:
        saliency = arena.pack_mask(torch.load(mask_path, map_location=arena.device, weights_only=False))
        mask_u8 = saliency if mask_u8 is None else (saliency & mask_u8)
    opt.set_mask(mask_u8)
    object.__setattr__(model, "_salun_last_optimizer", opt)  # checkpointing / parity tests read the Adam state
    model.train()
    losses = []
    for epoch in range(epochs):
        remain_iter = iter(remain_dl)
        for i, (z_f, c_forget, c_pseudo) in enumerate(forget_dl):
            try:
                z_r, c_r = next(remain_iter)
            except StopIteration:
                remain_iter = iter(remain_dl)
                z_r, c_r = next(remain_iter)
            opt.zero_grad()
            draws.next_step()
            sh_r = draws.shard_of(remain_dl, z_r.shape[0])
            with draws.scope(sh_r):
                remain_loss = model.shared_step({"z": z_r, "c": c_r})[0]
            z_f, c_forget, c_pseudo = z_f.to(model.device), c_forget.to(model.device), c_pseudo.to(model.device)
            sh_f = draws.shard_of(forget_dl, z_f.shape[0])
            with draws.scope(sh_f):
                t = draws.randint(model.num_timesteps, z_f.shape[0], model.device).long()
                noise = draws.randn_like(z_f)
                z_noisy = model.q_sample(x_start=z_f, t=t, noise=noise)
                forget_out, pseudo_out = forget_and_target(model, z_noisy, t, c_forget, c_pseudo)
            forget_loss = ops.mse_loss(pseudo_out, forget_out)
            if sh_f.weight != 1.0:  # shard means -> shares of the global-batch means under the AVG of the gradients
                forget_loss = forget_loss * sh_f.weight
            if sh_r.weight != 1.0:
                remain_loss = remain_loss * sh_r.weight
            loss = forget_loss + alpha * remain_loss
            loss.backward()
            opt.step()
            losses.append(loss.detach())
            if proximal_ratio is not None:
                # proximal_gradient.py:141-176: ratio of weights pulled back onto theta_0, linearly decaying schedule
                ratio = int(proximal_ratio * ((total_steps - (epoch * steps_per_epoch + i + 1)) / total_steps * n_total))
                if ratio < 1:
                    raise IndexError("index -1 is out of bounds for dimension 0 with size 0")  # topk(.., 0)[0][-1]
                # the n_frozen zeros of the frozen stages are the smallest |theta - theta_0|: the ratio-th smallest
                # over the whole model is 0 while ratio <= n_frozen (soft-threshold by 0 = identity), else the
                # (ratio - n_frozen)-th smallest over the U-Net
                if ratio > n_frozen:
                    ops.proximal_step(arena.params, init_params, ratio - n_frozen)
    model.eval()
    return [float(v) for v in torch.stack(losses).cpu()] if losses else []


def certain_label(class_to_forget, train_method, alpha, batch_size, epochs, lr, config_path, ckpt_path, mask_path,
                  diffusers_config_path, device, image_size=512, ddim_steps=50, *, model=None, forget_dl=None,
                  remain_dl=None):
    model = model or setup_model(config_path, ckpt_path, device)
    if forget_dl is None or remain_dl is None:
        raise ValueError("forget_dl: (latents, class context, pseudo-class context) batches; remain_dl: (latents, context)")
    return model, _unlearn(model, forget_dl, remain_dl, alpha, epochs, lr, mask_path, train_method)


def proximal_gradient(class_to_forget, train_method, alpha, batch_size, epochs, lr, config_path, ckpt_path, mask_ratio,
                      diffusers_config_path, device, image_size=512, ddim_steps=50, second_device=None, *, model=None,
                      forget_dl=None, remain_dl=None):
    """SD/train-scripts/proximal_gradient.py:14-186: random-label unlearning without a saliency mask, followed after
    every step by the proximal (soft-threshold) pull towards the initial weights with ratio
    mask_ratio * (remaining steps / total steps) * n_params.  The reference ships the 1-B-parameter vector to a second
    GPU for the top-k every step (`second_device`, :70-72,157-167); here the select runs in place on the flat arena
    (`second_device` is accepted and ignored)."""
    model = model or setup_model(config_path, ckpt_path, device)
    if forget_dl is None or remain_dl is None:
        raise ValueError("forget_dl: (latents, class context, pseudo-class context) batches; remain_dl: (latents, context)")
    return model, _unlearn(model, forget_dl, remain_dl, alpha, epochs, lr, None, train_method, proximal_ratio=mask_ratio)


def nsfw_removal(train_method, alpha, batch_size, epochs, lr, config_path, ckpt_path, mask_path,
                 diffusers_config_path, device, image_size=512, ddim_steps=50, *, model=None, forget_dl=None,
                 remain_dl=None):
    """forget prompt "a photo of a nude person" vs pseudo prompt "a photo of a person wearing clothes"
    (nsfw_removal.py:96-105,140-144); same loop as certain_label."""
    model = model or setup_model(config_path, ckpt_path, device)
    if forget_dl is None or remain_dl is None:
        raise ValueError("forget_dl: (latents, nude context, clothed context) batches; remain_dl: (latents, context)")
    return model, _unlearn(model, forget_dl, remain_dl, alpha, epochs, lr, mask_path, train_method)
