"""`model_dict` registry, as the reference exposes it (Classification/models/__init__.py:6-14).
Only ResNet-18 is on the benchmarked path; other names raise with a pointer to the scope table."""
from .resnet_cifar import NormalizeByChannelMeanStd, ResNetCifar, resnet18


class _ModelDict(dict):
    def __missing__(self, key):
        raise KeyError(f"architecture {key!r} is outside the hot-path scope of this build (SURVEY.md §2 C7: "
                       "only resnet18 is in the benchmark configs)")


model_dict = _ModelDict(resnet18=resnet18)
