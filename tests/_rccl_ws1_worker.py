"""Worker of tests/test_rccl_ws1_gpu.py: a few fused ResNet-18 unlearning steps on cuda:0, either single-process or —
with SALUN_FORCE_COLLECTIVES=1 — through the RCCL process group at world size 1 (dist.init_from_env with
`device_id=`, the async AVG gradient buckets of BucketedGradReducer, their join before the fused update, the
side-stream wgrad join the data-parallel path uses).  Prints one JSON line: digest of the parameters, losses, what ran."""
import hashlib
import json
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from unlearn_saliency_amd import dist as sdist
    from unlearn_saliency_amd import ops
    from unlearn_saliency_amd.Classification.models import model_dict
    from unlearn_saliency_amd.conv import use_salun_convs
    from unlearn_saliency_amd.flat import arena_of
    from unlearn_saliency_amd.norm import use_fused_bn
    from unlearn_saliency_amd.optim import FusedMaskedSGD
    rk, lrk, ws = sdist.init_from_env()
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(1)
    model = model_dict["resnet18"](num_classes=10).to(dev)
    assert use_salun_convs(model) == 20 and use_fused_bn(model) == 20
    arena = arena_of(model)
    n = arena.n
    mask = ops.mask_topk(ops.fill_normal(n, 5, 0.0, 1e-3) * (1.0 + ops.fill_uniform(n, 6, 0.0, 0.5)), [n // 2],
                         check=True)[0]
    opt = FusedMaskedSGD(arena, 0.013, momentum=0.9, weight_decay=5e-4)
    opt.set_mask(mask)
    crit = nn.CrossEntropyLoss()
    g = torch.Generator().manual_seed(3)
    model.train()
    losses = []
    launched = 0
    for _ in range(4):
        x = torch.rand(64, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (64,), generator=g).to(dev)
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        if opt._reducer is not None:
            launched += sum(opt._reducer.launched)  # slices that went out DURING backward
        opt.step()
        losses.append(float(loss.detach()))
    # the two flat collectives of Phase A on the same group
    v = torch.arange(1000, device=dev, dtype=torch.float32)
    sdist.all_reduce_sum_(v)
    w = torch.full((1000,), 3.0, device=dev)
    sdist.all_reduce_mean_(w)
    torch.cuda.synchronize()
    digest = hashlib.sha256(arena.params.cpu().numpy().tobytes()).hexdigest()
    out = {"digest": digest, "losses": losses, "collectives": bool(sdist.collectives_on()),
           "backend": torch.distributed.get_backend() if sdist.is_dist() else None, "world": ws,
           "rccl_ranks": sdist.counted_ranks(), "buckets_launched_in_backward": launched,
           "sum_ok": bool(torch.equal(v, torch.arange(1000, device=dev, dtype=torch.float32))),
           "avg_ok": bool((w == 3.0).all())}
    print(json.dumps(out), flush=True)
    if sdist.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
