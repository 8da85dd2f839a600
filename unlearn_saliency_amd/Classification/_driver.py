"""Shared body of main_random.py (masked: SalUn / Phase B) and main_forget.py (unmasked
baselines).  Follows the reference drivers (Classification/main_random.py:15-188,
main_forget.py:15-183): seed, model + marked dataset, forget/retain split, checkpoint
load, ``unlearn_method(loaders, model, criterion, args[, mask])``, checkpoint save,
accuracies on retain/forget/val/test (UA = 100 - forget accuracy), optional SVC-MIA."""
from __future__ import annotations

import os
import time
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import dist as sdist
from . import arg_parser, unlearn, utils
from .dataset import BatchLoader, split_marked
from .trainer import validate


def run(argv=None, use_mask: bool = True):
    args = arg_parser.parse_args(argv)
    rk, lrk, ws = sdist.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("the unlearning step needs a ROCm device: the fused kernels have no CPU fallback")
    if ws == 1:
        torch.cuda.set_device(int(args.gpu))
    device = torch.device("cuda", torch.cuda.current_device())
    os.makedirs(args.save_dir, exist_ok=True)
    if args.seed:
        utils.setup_seed(args.seed)
    seed = args.seed
    model, train_loader_full, val_loader, test_loader, marked_loader = utils.setup_model_dataset(args)
    model.to(device)
    if not args.library_conv:
        from ..conv import use_salun_convs
        use_salun_convs(model)  # convolutions on the fp32 MFMA kernels (csrc/salun_conv.hip)
        from ..norm import use_fused_bn
        use_fused_bn(model)  # BN(+add)+ReLU as one node (csrc/salun_norm.hip); SyncBatchNorm layers are left alone
    if ws > 1 and args.sync_bn:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)

    def replace_loader_dataset(dataset, batch_size=args.batch_size, seed=1, shuffle=True):
        utils.setup_seed(seed)
        return BatchLoader(dataset, batch_size, shuffle, device_resident=bool(args.device_loader), device=device,
                           rank=rk, world_size=ws)

    forget_dataset, retain_dataset = split_marked(marked_loader.dataset)
    forget_loader = replace_loader_dataset(forget_dataset, seed=seed, shuffle=True)
    retain_loader = replace_loader_dataset(retain_dataset, seed=seed, shuffle=True)
    assert len(forget_dataset) + len(retain_dataset) == len(train_loader_full.dataset)
    print(f"number of retain dataset {len(retain_dataset)}")
    print(f"number of forget dataset {len(forget_dataset)}")
    unlearn_data_loaders = OrderedDict(retain=retain_loader, forget=forget_loader, val=val_loader, test=test_loader)
    criterion = nn.CrossEntropyLoss()

    evaluation_result = None
    checkpoint = unlearn.load_unlearn_checkpoint(model, device, args) if args.resume else None
    if args.resume and checkpoint is not None:
        model, evaluation_result = checkpoint
    else:
        if args.model_path:
            ckpt = torch.load(args.model_path, map_location=device, weights_only=False)
            if "state_dict" in ckpt.keys():
                ckpt = ckpt["state_dict"]
            if args.unlearn != "retrain":
                model.load_state_dict(ckpt, strict=False)
        elif not args.synthetic:
            raise ValueError("--model_path is required (the original model's checkpoint)")
        mask = None
        if use_mask:
            if not args.mask_path:
                # the reference dies with NameError here (main_random.py:133-140)
                raise NameError("name 'mask' is not defined  [main_random needs --mask_path]")
            mask = torch.load(args.mask_path, map_location=device, weights_only=False)
        unlearn_method = unlearn.get_unlearn_method(args.unlearn)
        t0 = time.time()
        if use_mask:
            unlearn_method(unlearn_data_loaders, model, criterion, args, mask)
        else:
            unlearn_method(unlearn_data_loaders, model, criterion, args)
        torch.cuda.synchronize()
        print(f"unlearning wall time {time.time() - t0:.2f}s")
        if rk == 0:
            unlearn.save_unlearn_checkpoint(model, None, args)

    if evaluation_result is None:
        evaluation_result = {}
    if "new_accuracy" not in evaluation_result:
        accuracy = {}
        for name, loader in unlearn_data_loaders.items():
            utils.dataset_convert_to_test(loader.dataset, args)
            eval_loader = BatchLoader(loader.dataset, args.batch_size, False, device_resident=loader.device_resident,
                                      device=device)
            val_acc = validate(eval_loader, model, criterion, args)
            accuracy[name] = val_acc
            print(f"{name} acc: {val_acc}")
        evaluation_result["accuracy"] = accuracy
        if rk == 0:
            unlearn.save_unlearn_checkpoint(model, evaluation_result, args)

    for deprecated in ("MIA", "SVC_MIA", "SVC_MIA_forget"):
        evaluation_result.pop(deprecated, None)
    if "SVC_MIA_forget_efficacy" not in evaluation_result:
        try:
            from .evaluation import SVC_MIA
        except Exception as e:  # sklearn missing
            print(f"SVC_MIA skipped: {e}")
        else:
            test_len = len(test_loader.dataset)
            utils.dataset_convert_to_test(retain_dataset, args)
            utils.dataset_convert_to_test(forget_dataset, args)
            utils.dataset_convert_to_test(test_loader.dataset, args)
            shadow_train = BatchLoader(_head(retain_dataset, test_len), args.batch_size, False)
            evaluation_result["SVC_MIA_forget_efficacy"] = SVC_MIA(
                shadow_train=shadow_train, shadow_test=BatchLoader(test_loader.dataset, args.batch_size, False),
                target_train=None, target_test=BatchLoader(forget_dataset, args.batch_size, False), model=model)
    if rk == 0:
        unlearn.save_unlearn_checkpoint(model, evaluation_result, args)
    return evaluation_result


def _head(dataset, n):
    import copy
    d = copy.copy(dataset)
    d.data, d.targets = dataset.data[:n], dataset.targets[:n]
    return d
