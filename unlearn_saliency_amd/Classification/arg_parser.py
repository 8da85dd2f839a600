"""Command-line surface of the Classification entry points.

Flag names, types and defaults follow the reference's single parser
(Classification/arg_parser.py:4-145) so existing command lines keep working; flags that
only steer out-of-scope components (pruning, ImageNet) are accepted and ignored.
Additions of this build are grouped at the end (all optional).
"""
import argparse

# (flag, kwargs) — one row per reference flag, in the reference's order.
_REFERENCE_FLAGS = [
    # dataset
    ("--data", dict(type=str, default="../data", help="location of the data corpus")),
    ("--dataset", dict(type=str, default="cifar10", help="dataset")),
    ("--input_size", dict(type=int, default=32, help="size of input images")),
    ("--data_dir", dict(type=str, default="./tiny-imagenet-200", help="dir to tiny-imagenet")),
    ("--num_workers", dict(type=int, default=4)),
    ("--num_classes", dict(type=int, default=10)),
    # architecture
    ("--arch", dict(type=str, default="resnet18", help="model architecture")),
    ("--imagenet_arch", dict(action="store_true", help="architecture for imagenet size samples")),
    ("--train_y_file", dict(type=str, default="./labels/train_ys.pth")),
    ("--val_y_file", dict(type=str, default="./labels/val_ys.pth")),
    # general
    ("--seed", dict(type=int, default=2, help="random seed")),
    ("--train_seed", dict(type=int, default=1, help="seed for training (default value same as args.seed)")),
    ("--gpu", dict(type=int, default=0, help="gpu device id")),
    ("--workers", dict(type=int, default=4, help="number of workers in dataloader")),
    ("--resume", dict(action="store_true", help="resume from checkpoint")),
    ("--checkpoint", dict(type=str, default=None, help="checkpoint file")),
    ("--save_dir", dict(type=str, default=None, help="The directory used to save the trained models")),
    ("--model_path", dict(type=str, default=None, help="the path of original model")),
    # training
    ("--batch_size", dict(type=int, default=256, help="batch size")),
    ("--lr", dict(type=float, default=0.1, help="initial learning rate")),
    ("--momentum", dict(type=float, default=0.9, help="momentum")),
    ("--weight_decay", dict(type=float, default=5e-4, help="weight decay")),
    ("--epochs", dict(type=int, default=182, help="number of total epochs to run")),
    ("--warmup", dict(type=int, default=0, help="warm up epochs")),
    ("--print_freq", dict(type=int, default=50, help="print frequency")),
    ("--decreasing_lr", dict(default="91,136", help="decreasing strategy")),
    ("--no-aug", dict(action="store_true", default=False, help="No augmentation in training dataset")),
    ("--no-l1-epochs", dict(type=int, default=0, help="non l1 epochs")),
    # pruning (accepted for CLI compatibility; pruning baselines are out of scope)
    ("--prune", dict(type=str, default="omp")),
    ("--pruning_times", dict(type=int, default=1)),
    ("--rate", dict(type=float, default=0.95)),
    ("--prune_type", dict(type=str, default="rewind_lt")),
    ("--random_prune", dict(action="store_true")),
    ("--rewind_epoch", dict(type=int, default=0)),
    ("--rewind_pth", dict(type=str, default=None)),
    # unlearning
    ("--unlearn", dict(type=str, default="retrain", help="method to unlearn")),
    ("--unlearn_lr", dict(type=float, default=0.01, help="initial learning rate")),
    ("--unlearn_epochs", dict(type=int, default=10, help="number of total epochs for unlearn to run")),
    ("--num_indexes_to_replace", dict(type=int, default=None, help="Number of data to forget")),
    ("--class_to_replace", dict(type=int, default=-1, help="Specific class to forget")),
    ("--indexes_to_replace", dict(type=list, default=None, help="Specific index data to forget")),
    ("--alpha", dict(type=float, default=0.2, help="unlearn noise")),
    ("--mask_path", dict(type=str, default=None, help="the path of saliency map")),
]

# Extensions of the MI355X build.
_BUILD_FLAGS = [
    ("--synthetic", dict(action="store_true",
                         help="use the counter-based synthetic CIFAR-shaped set (no dataset files needed)")),
    ("--device_loader", dict(action="store_true",
                             help="keep the uint8 dataset resident in HBM and assemble batches on the device")),
    ("--mask_ratio", dict(type=float, default=0.5,
                          help="RL_proximal: fraction of weights reset per step at the start of the schedule (read as "
                               "args.mask_ratio by the reference's RL_pro.py:13, whose parser never defines the flag)")),
    ("--sync_bn", dict(action="store_true", help="SyncBatchNorm under multi-GPU data parallel")),
    ("--library_conv", dict(action="store_true",
                            help="use the library (MIOpen) convolutions instead of the fp32 MFMA kernels")),
    ("--thresholds", dict(type=str, default=None,
                          help="comma list of mask ratios for generate_mask (default: the reference's 0.1..1.0)")),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Classification of SalUn Experiments (MI355X-native hot path)")
    for flag, kw in _REFERENCE_FLAGS + _BUILD_FLAGS:
        parser.add_argument(flag, **kw)
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)
