"""Host-side profile of the bench step (cProfile) + wall vs device time, to find CPU-bound stretches."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import bench
from unlearn_saliency_amd.flat import arena_of
from unlearn_saliency_amd.optim import FusedMaskedSGD
from unlearn_saliency_amd import ops
from unlearn_saliency_amd.conv import use_salun_convs

dev = torch.device("cuda")
model, fl, rl = bench.build_workload(dev, 0, 1, 256)
torch.backends.cudnn.deterministic = False
torch.backends.cudnn.benchmark = True
if "--lib" not in sys.argv:
    use_salun_convs(model)
    from unlearn_saliency_amd.norm import use_fused_bn
    use_fused_bn(model)
arena = arena_of(model)
opt = FusedMaskedSGD(arena, 0.013, 0.9, 5e-4)
opt.set_mask(ops.mask_topk(ops.fill_normal(arena.n, 5, 0, 1e-3), [arena.n // 2])[0])
crit = nn.CrossEntropyLoss()
model.train()
stream = iter(bench.StepStream(fl, rl))

def step():
    x, y, _ = next(stream)
    loss = crit(model(x), y)
    opt.zero_grad()
    loss.backward()
    opt.step()

for _ in range(25):
    step()
torch.cuda.synchronize()
# (1) host-only time: how long does it take to *enqueue* 20 steps
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue 20 steps: {1e3*(t1-t0)/20:.2f} ms/step host; drained after {1e3*(t2-t0)/20:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
