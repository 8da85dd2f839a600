"""Optimizer factory, YAML -> Namespace config, run-directory layout (reference DDPM/functions/__init__.py)."""
from __future__ import annotations

import argparse
import os
from datetime import datetime

import yaml

from ...flat import FlatArena
from ...optim import FusedMaskedAdam, FusedMaskedSGD


def get_optimizer(config, parameters=None, arena: FlatArena = None):
    """`get_optimizer(config, model.parameters())` in the reference (functions/__init__.py:9-28) returns
    torch.optim.Adam(lr, weight_decay, betas=(beta1, 0.999), amsgrad, eps).  Here the caller passes the
    model's flat arena and gets the fused equivalent; grad clipping (config.optim.grad_clip) is folded in."""
    if arena is None:
        raise ValueError("get_optimizer needs the model's FlatArena (arena=arena_of(model))")
    o = config.optim
    if o.optimizer == "Adam":
        return FusedMaskedAdam(arena, lr=o.lr, betas=(o.beta1, 0.999), eps=o.eps, weight_decay=o.weight_decay,
                               amsgrad=o.amsgrad, grad_clip=getattr(o, "grad_clip", None))
    if o.optimizer == "SGD":
        return FusedMaskedSGD(arena, lr=o.lr, momentum=0.9)
    raise NotImplementedError("Optimizer {} not understood.".format(o.optimizer))


def dict2namespace(config: dict) -> argparse.Namespace:
    ns = argparse.Namespace()
    for key, value in config.items():
        setattr(ns, key, dict2namespace(value) if isinstance(value, dict) else value)
    return ns


def load_config(filename: str) -> argparse.Namespace:
    with open(filename, "r") as fp:
        return dict2namespace(yaml.safe_load(fp))


def _make_run_dirs(config, root):
    config.exp_root_dir = root
    config.log_dir = os.path.join(root, "logs")
    config.ckpt_dir = os.path.join(root, "ckpts")
    os.makedirs(config.log_dir, exist_ok=True)
    os.makedirs(config.ckpt_dir, exist_ok=True)
    with open(os.path.join(config.log_dir, "config.yaml"), "w") as fp:
        yaml.dump(config, fp)
    return config


def mask_tag(mask_path) -> str:
    """Run-dir tag derived from the mask file name (reference functions/__init__.py:58-68).  The
    reference crashes on `None` ("origin" in None); here no mask maps to "full"."""
    if mask_path:
        for tag in ("origin", "inverted", "random", "without"):
            if tag in mask_path:
                return tag
    return "full"


def get_mask_config_and_setup_dirs(args, filename: str):
    """results/<dataset>/forget/<method>/<alpha>_<masktag>/<timestamp>/{logs,ckpts}"""
    config = load_config(filename)
    stamp = datetime.now().strftime("%Y_%m_%d_%H%M%S")
    root = os.path.join("./results", config.data.dataset.lower(), "forget", args.method,
                        f"{args.alpha}_{mask_tag(args.mask_path)}", stamp)
    return _make_run_dirs(config, root)


def get_config_and_setup_dirs(filename: str):
    config = load_config(filename)
    stamp = datetime.now().strftime("%Y_%m_%d_%H%M%S")
    return _make_run_dirs(config, os.path.join("./results", config.data.dataset.lower(), stamp))


def cycle(dl):
    while True:
        for data in dl:
            yield data
