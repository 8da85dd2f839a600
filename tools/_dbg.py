import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fixtures import fill_params, sd_tiny_config
from unlearn_saliency_amd import ops
from unlearn_saliency_amd.SD import train_scripts as TS
from unlearn_saliency_amd.SD.ldm_lite import LatentDiffusionLite
from unlearn_saliency_amd.conv_bf16 import use_salun_convs_bf16, use_salun_linears_bf16
from unlearn_saliency_amd.optim import FusedMaskedAdam
cfg = dict(sd_tiny_config(), model_channels=64, num_heads=2, context_dim=64, use_checkpoint=True)
model = LatentDiffusionLite(cfg, bf16=True)
fill_params(model.model.diffusion_model, 9000)
model = model.cuda().train()
arena = TS._unet_arena(model)
print("convs", use_salun_convs_bf16(model), "linears", use_salun_linears_bf16(model), "params", len(arena.names))
opt = FusedMaskedAdam(arena, lr=1e-5)
mk = lambda *s: torch.randn(*s, device="cuda")
def step():
    opt.zero_grad()
    remain = model.shared_step({"z": mk(2, 4, 8, 8), "c": mk(2, 7, 64)})[0]
    z = mk(2, 4, 8, 8); t = torch.randint(0, 1000, (2,), device="cuda"); noise = torch.randn_like(z)
    zn = model.q_sample(z, t, noise)
    fo, po = TS.forget_and_target(model, zn, t, mk(2, 7, 64), mk(2, 7, 64))
    loss = ops.mse_loss(po, fo) + 0.1 * remain
    loss.backward()
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::add", "aten::add_", "aten::copy_", "aten::to", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::sum", "aten::mul")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(e.key, e.count, e.input_shapes[:3])
