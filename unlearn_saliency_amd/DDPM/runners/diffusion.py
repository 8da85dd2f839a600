"""`Diffusion` runner: the three hot entry points of the reference's DDPM/runners/diffusion.py —

    generate_mask()     Phase A for the CFG-DDPM U-Net           (reference :933-1039)
    saliency_unlearn()  masked ε-MSE unlearning loop (rl / ga)   (reference :482-619)
    save_fim()          diagonal empirical Fisher                 (reference :101-191)

— same constructor (`Diffusion(args, config)`), same `args` / YAML fields, same artefacts
(`results/cifar10/mask/{label}/with_0.5.pt` with `module.`-prefixed int64 tensors; `ckpts/ckpt.pth` =
`[model_state, optimizer_state, step]`; `fisher_dict.pkl`).  Sampling / pre-training / FID stay out of
scope.

What is different underneath (MI355X-first):
  * one process per GPU + RCCL instead of nn.DataParallel's per-step parameter broadcast / output gather;
    state_dict keys still carry the `module.` prefix so checkpoints and masks interoperate;
  * gradients never leave the device: the reference moves all 334 gradients to the CPU every batch
    (154.5 MB D2H, :995) and the int64 mask back to the GPU every step (309 MB H2D, :592);
    here Σ grads is a flat fp32 vector updated by `salun_saliency_accumulate` with the per-batch clip
    coefficient read from device memory, and the mask is a resident flat u8 vector;
  * clip → mask → Adam is `salun_grad_sqnorm` + `salun_masked_adam_step`; q-sample and ε-MSE are
    `salun_qsample` / `salun_sqerr_loss`.
"""
from __future__ import annotations

import logging
import os
import pickle
import time
from collections import OrderedDict

import numpy as np
import torch

from ... import dist as sdist
from ... import draws, hostperf, ops
from ...flat import FlatArena, arena_of
from ...streams import concurrent_stream
from ..datasets import data_transform, get_forget_dataset
from ..functions import cycle, get_optimizer
from ..functions.losses import loss_registry_conditional, q_sample
from ..models.diffusion import Conditional_Model

DP_PREFIX = "module."  # reference checkpoints / masks come from nn.DataParallel-wrapped models


# SALUN_DDPM_TARGET_OVERLAP=0: run the RL method's no-grad target pass on the main stream, after the forget pass
PSEUDO_OVERLAP = os.environ.get("SALUN_DDPM_TARGET_OVERLAP", "1") != "0"
_target_streams: dict = {}


def _target_stream(device) -> "torch.cuda.Stream":
    s = _target_streams.get(device)
    if s is None:
        s = _target_streams[device] = concurrent_stream(device)
    return s


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """float64 numpy β_1..β_T (reference :36-66)."""
    T = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(T, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        betas = 1 / (np.exp(-np.linspace(-6, 6, T)) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


def antithetic_timesteps(n: int, num_timesteps: int, device) -> torch.Tensor:
    """t ~ randint(T) for n//2+1 draws, mirrored with T-t-1, truncated to n (reference :530-533)."""
    t = torch.randint(low=0, high=num_timesteps, size=(n // 2 + 1,))  # host generator, as the reference draws them
    if torch.device(device).type == "cuda":
        # pinned + non_blocking: a pageable host-to-device copy blocks the host until everything queued before it has
        # run — twice per unlearning step, which kept the host from ever running ahead of the device
        t = t.pin_memory().to(device, non_blocking=True)
    else:
        t = t.to(device)
    return torch.cat([t, num_timesteps - t - 1], dim=0)[:n]


class ShardDraws:
    """Per-batch randomness under data parallel (draws.py): noise, timesteps, the label-drop mask and the dropout
    keep-decisions are functions of the GLOBAL batch — drawn on every rank from identically seeded generators (like the
    reference's single process feeding nn.DataParallel, runners/diffusion.py:124) and sliced to this rank's shard
    [lo, hi), or keyed by the global sample index (dropout).  `with ShardDraws(...) as d:` makes the model calls
    inside see the shard.  Single-process: plain draws."""

    def __init__(self, loader, n_local: int):
        self.shard = draws.shard_of(loader, n_local)
        self.lo, self.hi, self.b = self.shard.lo, self.shard.hi, self.shard.b
        self._scope = draws.scope(self.shard)

    def __enter__(self):
        self._scope.__enter__()
        return self

    def __exit__(self, *exc):
        return self._scope.__exit__(*exc)

    def randn_like(self, x: torch.Tensor) -> torch.Tensor:
        return draws.randn_like(x, self.shard)

    def timesteps(self, num_timesteps: int, device) -> torch.Tensor:
        return antithetic_timesteps(self.b, num_timesteps, device)[self.lo:self.hi]

    @property
    def weight(self) -> float:
        """shard-mean loss -> share of the global-batch mean under the AVG all-reduce of the gradients"""
        return self.shard.weight


def strip_prefix(state: dict, prefix: str = DP_PREFIX) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k[len(prefix):] if k.startswith(prefix) else k, v) for k, v in state.items())


def add_prefix(state: dict, prefix: str = DP_PREFIX) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((prefix + k, v) for k, v in state.items())


class Diffusion(object):
    def __init__(self, args, config):
        self.args, self.config = args, config
        if not torch.cuda.is_available():
            raise RuntimeError("the DDPM hot path needs a ROCm device (no CPU fallback)")
        draws.seed(None)  # a run's dropout keys restart from (torch seed, step 0): main.py seeds torch before this
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model_var_type = config.model.var_type
        betas = get_beta_schedule(beta_schedule=config.diffusion.beta_schedule,
                                  beta_start=config.diffusion.beta_start, beta_end=config.diffusion.beta_end,
                                  num_diffusion_timesteps=config.diffusion.num_diffusion_timesteps)
        self.betas = torch.from_numpy(betas).float().to(self.device)
        self.num_timesteps = self.betas.shape[0]
        alphas_cumprod = (1.0 - self.betas).cumprod(dim=0)
        prev = torch.cat([torch.ones(1, device=self.device), alphas_cumprod[:-1]], dim=0)
        posterior_variance = self.betas * (1.0 - prev) / (1.0 - alphas_cumprod)
        if self.model_var_type == "fixedlarge":
            self.logvar = self.betas.log()
        elif self.model_var_type == "fixedsmall":
            self.logvar = posterior_variance.clamp(min=1e-20).log()

    # ------------------------------------------------------------------ helpers
    def _load_model(self):
        """Conditional_Model on this GPU with `states[0]` of `{ckpt_folder}/ckpts/ckpt.pth` loaded (keys with or
        without the DataParallel prefix).  `args.init_model` (a state_dict) or a missing checkpoint with
        `args.synthetic` keeps the random initialisation (benchmarks: no checkpoints offline)."""
        args = self.args
        model = Conditional_Model(self.config).to(self.device)
        path = os.path.join(args.ckpt_folder, "ckpts/ckpt.pth") if getattr(args, "ckpt_folder", None) else None
        if path and os.path.exists(path):
            print("Loading checkpoints {}".format(args.ckpt_folder))
            states = torch.load(path, map_location=self.device, weights_only=False)
            model.load_state_dict(strip_prefix(states[0]), strict=True)
        elif not getattr(args, "synthetic", False):
            raise FileNotFoundError(f"{path}: original-model checkpoint not found (pass --synthetic to benchmark "
                                    "a randomly initialised U-Net)")
        if not getattr(args, "library_conv", False):
            from ...conv import use_salun_convs
            use_salun_convs(model)  # fp32 MFMA convolution kernels instead of the library's heuristics
        return model

    def _loaders(self):
        self._remain_loader, self._forget_loader = get_forget_dataset(
            self.args, self.config, self.args.label_to_forget, device=self.device,
            synthetic=True if getattr(self.args, "synthetic", False) else None)
        return self._remain_loader, self._forget_loader

    # --------------------------------------------------------- Phase A: mask
    def accumulate_saliency(self, model, forget_loader, arena: FlatArena = None) -> torch.Tensor:
        """Σ_batches clip_1.0(∇ ε-MSE of the CFG-combined prediction), model in eval mode (reference :957-996).
        Exact under data parallel: each rank back-propagates its shard's share of the global-batch mean, the
        flat gradient is all-reduced per batch (the clip needs the global norm), then accumulated."""
        args, config = self.args, self.config
        arena = arena or arena_of(model)
        acc = arena.new_like()
        sq = torch.zeros(1, device=self.device)
        model.eval()
        hostperf.freeze_gc()
        ws = sdist.world_size()
        for x, forget_c in forget_loader:
            n = x.size(0)
            sd = ShardDraws(forget_loader, n)
            x = data_transform(config, x.to(self.device))
            e = sd.randn_like(x)
            t = sd.timesteps(self.num_timesteps, self.device)
            xt = q_sample(x, t, e, self.betas)
            draws.next_step()
            with sd:
                output = model(xt, t.float(), forget_c, cond_scale=args.cond_scale, mode="test")
            loss = ops.eps_mse(e, output)
            if ws > 1:  # local mean -> share of the global-batch mean (the gradients are SUMmed below)
                loss = loss * (n / float(sd.b))
            arena.zero_grad()
            loss.backward()
            sdist.all_reduce_sum_(arena.grads)
            clip = getattr(config.optim, "grad_clip", None)
            if clip:
                ops.grad_sqnorm(arena.grads, sq)
                ops.saliency_accumulate(acc, arena.grads, sqnorm=sq, max_norm=clip)
            else:
                ops.saliency_accumulate(acc, arena.grads, 1.0)
        return acc

    def generate_mask(self):
        args, config = self.args, self.config
        logging.info("Generating mask of diffusion to achieve gradient sparsity. "
                     f"Gamma: {config.training.gamma}, lambda: {config.training.lmbda}")
        _, forget_loader = self._loaders()
        model = self._load_model()
        arena = arena_of(model)
        t0 = time.time()
        acc = self.accumulate_saliency(model, forget_loader, arena)
        threshold_list = [0.5]  # reference :1006
        masks = ops.mask_topk(acc, [int(arena.n * r) for r in threshold_list], check=True)
        torch.cuda.synchronize()
        logging.info(f"saliency + top-k: {time.time() - t0:.2f}s")
        mask_path = os.path.join(getattr(args, "mask_dir", "results/cifar10/mask"), str(args.label_to_forget))
        if sdist.rank() == 0:
            os.makedirs(mask_path, exist_ok=True)
            for r, m in zip(threshold_list, masks):
                print(r)
                torch.save(arena.unpack_mask(m, prefix=DP_PREFIX), os.path.join(mask_path, f"with_{str(r)}.pt"))
        sdist.barrier()
        return dict(zip(threshold_list, masks))

    # ------------------------------------------------- Phase B: masked unlearning
    def unlearn_step(self, model, optimizer, remain_batch, forget_batch):
        """One iteration of the reference loop body (:520-593); returns the loss tensor (no host sync)."""
        args, config = self.args, self.config
        b = self.betas
        remain_x, remain_c = remain_batch
        n = remain_x.size(0)
        draws.next_step()  # dropout keys: (step, call index) — identical on every rank
        sd = ShardDraws(getattr(self, "_remain_loader", None), n)  # global-batch draws, sliced (data parallel)
        remain_x = data_transform(config, remain_x.to(self.device))
        e = sd.randn_like(remain_x)
        t = sd.timesteps(self.num_timesteps, self.device)
        with sd:
            remain_loss = loss_registry_conditional[config.model.type](model, remain_x, t, remain_c, e, b)
        if sd.weight != 1.0:
            remain_loss = remain_loss * sd.weight

        forget_x, forget_c = forget_batch
        n = forget_x.size(0)
        sd = ShardDraws(getattr(self, "_forget_loader", None), n)
        forget_x = data_transform(config, forget_x.to(self.device))
        e = sd.randn_like(forget_x)
        t = sd.timesteps(self.num_timesteps, self.device)
        forget_weight = sd.weight
        if args.method == "ga":
            with sd:
                forget_loss = -loss_registry_conditional[config.model.type](model, forget_x, t, forget_c, e, b)
        elif args.method == "rl":
            xt = q_sample(forget_x, t, e, b)
            pseudo_c = torch.full(forget_c.shape, (args.label_to_forget + 1) % 10, device=forget_c.device)
            tf = t.float()
            # (also under data parallel: it was switched off there while the gradient slices' collectives had a launch stream
            # of their own — 122 vs 107 ms at world size 1 with both; with the collectives issued from the backward-weight
            # stream, dist.BucketedGradReducer, it gains 2 %: 107.6 -> 105.5 ms)
            side = _target_stream(xt.device) if (PSEUDO_OVERLAP and xt.is_cuda) else None
            if side is not None:
                # the no-grad target pass depends only on (xt, t): it runs on its own stream NEXT TO the forward pass
                # that is differentiated, filling the phases where one pass leaves the chip half empty (4x4 / 8x8
                # levels, 1x1 projections, GroupNorm, attention).  Host call order — hence the order in which dropout /
                # conditioning-drop draws are taken — is unchanged: `output` first, then the target.
                main = torch.cuda.current_stream(xt.device)
                side.wait_stream(main)
            with sd:
                output = model(xt, tf, forget_c, mode="train")
                if side is not None:
                    with torch.no_grad(), torch.cuda.stream(side):
                        pseudo = model(xt, tf, pseudo_c, mode="train")
                    for v in (xt, tf, pseudo_c):
                        v.record_stream(side)
                    main.wait_stream(side)
                    pseudo.record_stream(main)
                else:
                    with torch.no_grad():
                        pseudo = model(xt, tf, pseudo_c, mode="train")
            forget_loss = ops.mse_loss(pseudo, output)
        else:
            raise ValueError(f"unknown --method {args.method!r} (rl | ga)")
        if forget_weight != 1.0:
            forget_loss = forget_loss * forget_weight
        loss = forget_loss + args.alpha * remain_loss

        optimizer.zero_grad()
        loss.backward()
        optimizer.clip_grad_norm_(config.optim.grad_clip)  # clip -> mask -> Adam, fused into step()
        optimizer.step()
        return loss

    def saliency_unlearn(self):
        args, config = self.args, self.config
        remain_loader, forget_loader = self._loaders()
        remain_iter, forget_iter = cycle(remain_loader), cycle(forget_loader)
        mask = torch.load(args.mask_path, map_location=self.device, weights_only=False) if args.mask_path else None
        model = self._load_model()
        arena = arena_of(model)
        optimizer = get_optimizer(config, arena=arena)
        if mask:
            optimizer.set_mask(arena.pack_mask(strip_prefix(mask)))
        ema_helper = self._ema(model)  # reference :509-513 (off in cifar10_saliency_unlearn.yml)
        model.train()
        start = time.time()
        self.last_optimizer = optimizer
        self.step_losses = []  # device scalars of the last `keep_losses` steps (no host sync; diagnostics / parity tests)
        keep_losses = int(getattr(args, "keep_losses", 64))
        hostperf.freeze_gc()
        for step in range(0, config.training.n_iters):
            model.train()
            loss = self.unlearn_step(model, optimizer, next(remain_iter), next(forget_iter))
            self.step_losses.append(loss.detach())
            if len(self.step_losses) > keep_losses:
                self.step_losses.pop(0)
            if ema_helper is not None:
                ema_helper.update(model)
            if (step + 1) % config.training.log_freq == 0:
                end = time.time()
                logging.info(f"step: {step}, loss: {loss.item()}, time: {end - start}")
                start = time.time()
            if (step + 1) % config.training.snapshot_freq == 0 and sdist.rank() == 0:
                states = [add_prefix(model.state_dict()), optimizer.state_dict(), step]
                if ema_helper is not None:
                    states.append(ema_helper.state_dict())  # reference :607-608
                torch.save(states, os.path.join(config.ckpt_dir, "ckpt.pth"))
        return model

    def _ema(self, model):
        """EMAHelper(mu=config.model.ema_rate) registered on the model when config.model.ema is set (reference :205-209)."""
        if not getattr(self.config.model, "ema", False):
            return None
        from ..models.ema import EMAHelper
        helper = EMAHelper(mu=self.config.model.ema_rate)
        helper.register(model)
        return helper

    # ------------------------------------------------------------------ sampling (evaluation of an unlearned model)
    def sample_image(self, x, model, c, cond_scale, last=True):
        """Reverse process from noise `x` for classes `c` (reference :828-875): `args.sample_type` "generalized" (DDIM,
        `args.eta`) or "ddpm_noisy" (ancestral), `args.skip_type` "uniform" | "quad", `args.timesteps` steps.
        (The reference's "ddpm_noisy" branch imports a misspelt name, `ddpm_steps_conditional`, and fails; here it
        runs `ddpm_step_conditional`.)  `last=True` keeps only the final state on the device."""
        from ..functions.denoising import ddpm_step_conditional, generalized_steps_conditional
        args = self.args
        if args.skip_type == "uniform":
            seq = range(0, self.num_timesteps, self.num_timesteps // args.timesteps)
        elif args.skip_type == "quad":
            seq = [int(s) for s in list(np.linspace(0, np.sqrt(self.num_timesteps * 0.8), args.timesteps) ** 2)]
        else:
            raise NotImplementedError
        keep = "last" if last else "all"
        if args.sample_type == "generalized":
            out = generalized_steps_conditional(x, c, seq, model, self.betas, cond_scale, eta=args.eta, keep=keep)
        elif args.sample_type == "ddpm_noisy":
            out = ddpm_step_conditional(x, c, seq, model, self.betas, cond_scale, keep=keep)
        else:
            raise NotImplementedError
        return out[0][-1] if last else out

    # ---------------------------------------------- EWC / Selective Amnesia
    def forget_step(self, model, optimizer, remember_batch, fisher_flat, params_mle_flat):
        """One iteration of the reference's `train_forget` body (runners/diffusion.py:313-365):
        eps-MSE on uniform-noise images labelled with the class to forget + gamma * eps-MSE on the remember batch
        (both without label dropout, one shared antithetic t), plus  lmbda * sum F (theta - theta*)^2.
        The EWC term is not built in autograd: after the U-Net backward, `salun_ewc_penalty_grad` adds its gradient
        into the flat gradient (one launch instead of 334 x 4 tensor ops) — before the clip, as in the reference,
        where the clip sees the sum of both gradients.  Returns (loss, forgetting_loss, ewc_loss) tensors."""
        args, config = self.args, self.config
        b = self.betas
        draws.next_step()  # dropout keys are (step, call index): every training step advances the step
        x_remember, c_remember = remember_batch
        x_remember, c_remember = x_remember.to(self.device), c_remember.to(self.device)
        x_remember = data_transform(config, x_remember)
        n = x_remember.size(0)
        c_forget = (torch.ones(n, dtype=int) * args.label_to_forget).to(self.device)
        x_forget = (torch.rand((n, config.data.channels, config.data.image_size, config.data.image_size),
                               device=self.device) - 0.5) * 2.0
        e_remember = torch.randn_like(x_remember)
        e_forget = torch.randn_like(x_forget)
        t = antithetic_timesteps(n, self.num_timesteps, self.device)
        loss_fn = loss_registry_conditional[config.model.type]
        forgetting = loss_fn(model, x_forget, t, c_forget, e_forget, b, cond_drop_prob=0.0) + \
            config.training.gamma * loss_fn(model, x_remember, t, c_remember, e_remember, b, cond_drop_prob=0.0)
        optimizer.zero_grad()
        forgetting.backward()
        arena = optimizer.arena
        ewc = ops.ewc_penalty_grad(arena.params, params_mle_flat, fisher_flat, arena.grads, config.training.lmbda)
        optimizer.clip_grad_norm_(config.optim.grad_clip)
        optimizer.step()
        return forgetting.detach() + ewc[0], forgetting.detach(), ewc[0]

    def train_forget(self, remember_loader=None, fisher_dict=None):
        """`--mode forget`: contrastive forgetting with the EWC anchor (reference :273-396).  `fisher_dict` defaults
        to {ckpt_folder}/fisher_dict.pkl (written by save_fim); `remember_loader` defaults to the all-but-one-class
        sample folder (synthetic remain set with --synthetic)."""
        args, config = self.args, self.config
        logging.info(f"Training diffusion forget with contrastive and EWC. Gamma: {config.training.gamma}, "
                     f"lambda: {config.training.lmbda}")
        if remember_loader is None:
            remember_loader, _ = self._loaders()
        remember_iter = cycle(remember_loader)
        model = self._load_model()
        arena = arena_of(model)
        optimizer = get_optimizer(config, arena=arena)
        ema_helper = self._ema(model)
        if fisher_dict is None:
            with open(os.path.join(args.ckpt_folder, "fisher_dict.pkl"), "rb") as f:
                fisher_dict = pickle.load(f)
        fisher_flat = arena.pack(strip_prefix(fisher_dict))
        params_mle_flat = arena.params.clone()
        hostperf.freeze_gc()
        for step in range(0, config.training.n_iters):
            model.train()
            loss, forgetting_loss, ewc_loss = self.forget_step(model, optimizer, next(remember_iter), fisher_flat,
                                                               params_mle_flat)
            if ema_helper is not None:
                ema_helper.update(model)
            if (step + 1) % config.training.log_freq == 0:
                logging.info(f"step: {step}, loss: {loss.item()}, forgetting loss: {forgetting_loss.item()}, "
                             f"ewc loss: {ewc_loss.item()}")
            if (step + 1) % config.training.snapshot_freq == 0 and sdist.rank() == 0:
                states = [add_prefix(model.state_dict()), optimizer.state_dict(), step]
                if ema_helper is not None:
                    states.append(ema_helper.state_dict())
                torch.save(states, os.path.join(config.ckpt_dir, "ckpt.pth"))
        return model

    # ------------------------------------------------------------------ Fisher
    def save_fim(self, samples=None):
        """F = (1/N) Σ_samples (Σ_t ∇ℓ_t(x, c))²  (reference :101-191): per sample, the ε-MSE gradient is summed
        over all T timesteps in `n_chunks` chunks (one backward per chunk), then squared and accumulated by
        `salun_fim_square_accumulate`.  Samples shard over ranks; F is all-reduced once at the end.
        `samples`: optional iterable of (x[1,3,H,W] in [0,1], c[1]) — default: {ckpt_folder}/class_samples."""
        args, config = self.args, self.config
        model = self._load_model()
        model.eval()
        arena = arena_of(model)
        if samples is None:
            samples = _image_folder_samples(os.path.join(args.ckpt_folder, "class_samples"), self.device)
        samples = list(samples)
        n_data = len(samples)
        lo, hi = sdist.shard_bounds(n_data)
        F = arena.new_like()
        tmp = arena.new_like()
        ts = torch.chunk(torch.arange(0, self.num_timesteps), args.n_chunks)
        for x, c in samples[lo:hi]:
            x, c = x.to(self.device), c.to(self.device)  # NB: not rescaled to [-1,1], as in the reference (:106-109)
            for _t in ts:
                loss = 0
                for i in range(len(_t)):
                    e = torch.randn_like(x)
                    t = torch.tensor([_t[i]]).expand(x.size(0)).to(self.device)
                    loss = loss + loss_registry_conditional[config.model.type](model, x, t, c, e, self.betas,
                                                                              keepdim=True)
                arena.zero_grad()
                loss.sum().backward()  # one sample per pass => the per-sample gradient
                ops.saliency_accumulate(tmp, arena.grads, 1.0)
            ops.fim_square_accumulate(F, tmp, float(n_data))
        sdist.all_reduce_sum_(F)
        fisher_dict = arena.view_dict(F, prefix=DP_PREFIX)
        if sdist.rank() == 0 and getattr(args, "ckpt_folder", None):
            with open(os.path.join(args.ckpt_folder, "fisher_dict.pkl"), "wb") as f:
                pickle.dump(OrderedDict((k, v.clone()) for k, v in fisher_dict.items()), f)
        return fisher_dict


def _image_folder_samples(root, device):
    """Minimal ImageFolder(ToTensor) reader: root/<class>/*.png|jpg -> (x[1,3,H,W], c[1]), needs PIL."""
    from PIL import Image
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    for ci, cname in enumerate(classes):
        for fn in sorted(os.listdir(os.path.join(root, cname))):
            img = np.asarray(Image.open(os.path.join(root, cname, fn)).convert("RGB"))
            x = torch.from_numpy(img).permute(2, 0, 1).float().div(255)[None]
            yield x.to(device), torch.tensor([ci], device=device)
