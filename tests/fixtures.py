"""Fixture models / inputs shared by tests/golden/make_golden*.py (which feed them to the
reference) and by the tests (which feed them to the oracle and the HIP path).  Everything is
derived from the counter-based generator, so it regenerates identically anywhere."""
import numpy as np
import torch
import torch.nn as nn

from unlearn_saliency_amd import rng


class TinyCNN(nn.Module):
    """~5k-parameter BN network, train/eval sensitive like ResNet (fixture model; ours, not the reference's)."""

    def __init__(self, num_classes=10):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.conv2 = nn.Conv2d(8, 16, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(16)
        self.fc = nn.Linear(16, num_classes)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        x = torch.relu(self.bn2(self.conv2(x)))
        return self.fc(x.mean(dim=(2, 3)))


def tiny_state(seed):
    m = TinyCNN()
    sd = m.state_dict()
    off = 0
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            if "running_var" in k:
                v.copy_(torch.from_numpy(rng.uniform(v.numel(), seed + off, 0.5, 1.5)).view_as(v))
            elif k.startswith("bn") and k.endswith("weight"):
                v.copy_(torch.from_numpy(rng.uniform(v.numel(), seed + off, 0.8, 1.2)).view_as(v))
            else:
                v.copy_(torch.from_numpy(rng.normal(v.numel(), seed + off, 0.0, 0.2)).view_as(v))
        off += 1000
    return sd


def tiny_batches(nb, bs, seed):
    out = []
    for b in range(nb):
        x = rng.uniform(bs * 3 * 8 * 8, seed + 10 * b, 0.0, 1.0).reshape(bs, 3, 8, 8)
        y = (rng.u8(bs, seed + 10 * b + 1) % 10).astype(np.int64)
        out.append((x, y))
    return out


def saliency_vector(n, seed, std=1e-3):
    """High-entropy synthetic saliency: normal * (1 + U[0, 0.5)), two exact fp32 roundings, so the same
    vector is rebuilt bit-for-bit by numpy here and by ops.fill_* + torch on the device.  (The plain
    Irwin-Hall normal has only ~4e5 distinct magnitudes, i.e. ties at every threshold for N >> 1e5.)"""
    z = rng.normal(n, seed, 0.0, std)
    u = rng.uniform(n, seed + 7, 0.0, 0.5)
    return (z * (np.float32(1.0) + u)).astype(np.float32)


def saliency_vector_wide(n, seed):
    """Same, spread over 40 binades (x 2^j, j in [-20, 20), exact): at N = 11 M a narrow-range fp32 vector
    has a duplicate at most thresholds, which makes the reference's unstable argsort ambiguous there; the
    wide one keeps almost every threshold unique so whole-mask hashes can be pinned."""
    base = saliency_vector(n, seed, 1.0)
    j = np.floor(rng.uniform(n, seed + 13, 0.0, 40.0)).astype(np.int32) - 20
    return np.ldexp(base, j).astype(np.float32)


def tau_is_unique(abs_sorted_desc, k):
    """True iff the k-th largest magnitude occurs once (the reference's unstable argsort is then well defined)."""
    n = len(abs_sorted_desc)
    if k <= 0 or k >= n:
        return True
    t = abs_sorted_desc[k - 1]
    left = k >= 2 and abs_sorted_desc[k - 2] == t
    right = abs_sorted_desc[k] == t
    return not (left or right)


# ------------------------------------------------------------------ DDPM fixtures
def ddpm_small_config(T=1000, dropout=0.0):
    """Reduced CFG-DDPM config (ch must stay 128: the reference hard-wires cemb_channels=512)."""
    from argparse import Namespace as NS
    return NS(
        data=NS(path="./data", dataset="CIFAR10", image_size=16, channels=3, logit_transform=False,
                uniform_dequantization=False, gaussian_dequantization=False, random_flip=False, rescaled=True,
                num_workers=0, n_classes=10),
        model=NS(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2], num_res_blocks=1,
                 attn_resolutions=[8], dropout=dropout, var_type="fixedlarge", ema_rate=0.9999, ema=False,
                 resamp_with_conv=True, cond_drop_prob=0.1),
        diffusion=NS(beta_schedule="linear", beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=T),
        training=NS(batch_size=4, n_iters=2, snapshot_freq=10 ** 9, log_freq=10 ** 9, visualization_samples=100,
                    train_embeddings=False, gamma=1, lmbda=10, save_freq=10 ** 9),
        sampling=NS(batch_size=4, last_only=True),
        optim=NS(weight_decay=0.0, optimizer="Adam", lr=0.0001, beta1=0.9, amsgrad=False, eps=1e-8, grad_clip=1.0),
    )


def fill_params(model, seed):
    """Deterministic parameters from the counter-based generator (same on any machine / torch version)."""
    with torch.no_grad():
        for i, (name, p) in enumerate(model.named_parameters()):
            s = seed + 1000 * i
            if p.dim() > 1:
                v = rng.normal(p.numel(), s, 0.0, 0.05)
            elif "norm" in name and name.endswith("weight"):
                v = rng.uniform(p.numel(), s, 0.8, 1.2)
            else:
                v = rng.normal(p.numel(), s, 0.0, 0.02)
            p.copy_(torch.from_numpy(v).view_as(p))
    return model


def ddpm_batch(n, seed, image_size=16, label=None):
    x = rng.uniform(n * 3 * image_size * image_size, seed, 0.0, 1.0).reshape(n, 3, image_size, image_size)
    c = (rng.u8(n, seed + 1) % 10).astype(np.int64) if label is None else np.full(n, label, np.int64)
    return x, c


def flat_params(model):
    return np.concatenate([p.detach().reshape(-1).cpu().numpy() for p in model.parameters()])


# -------------------------------------------------------------------- SD fixtures
def sd_tiny_config():
    return dict(image_size=8, in_channels=4, out_channels=4, model_channels=32, attention_resolutions=(2, 1),
                num_res_blocks=1, channel_mult=(1, 2), num_heads=4, use_spatial_transformer=True, transformer_depth=1,
                context_dim=24, use_checkpoint=False, legacy=False)


# ------------------------------------------------------------------ "next rows" fixtures
def next_rows_datasets():
    """24 forget + 40 retain uint8 8x8 images as ArrayDatasets (test transform: no augmentation randomness)."""
    from unlearn_saliency_amd.Classification.dataset import ArrayDataset
    fx = rng.u8(24 * 8 * 8 * 3, 1100).reshape(24, 8, 8, 3)
    fy = (rng.u8(24, 1101) % 10).astype(np.int64)
    rx = rng.u8(40 * 8 * 8 * 3, 1102).reshape(40, 8, 8, 3)
    ry = (rng.u8(40, 1103) % 10).astype(np.int64)
    return ArrayDataset(fx, fy, transform="test"), ArrayDataset(rx, ry, transform="test")


def ewc_inputs():
    shapes = [(16, 8, 3, 3), (16,), (32, 16), (32,)]
    ps, stars, Fs = [], [], []
    for i, s in enumerate(shapes):
        k = int(np.prod(s))
        stars.append(rng.normal(k, 1300 + i, 0.0, 0.05))
        ps.append((stars[-1] + rng.normal(k, 1310 + i, 0.0, 0.01)).astype(np.float32))
        Fs.append(np.abs(rng.normal(k, 1320 + i, 0.0, 1.0)).astype(np.float32))
    return np.concatenate(ps), np.concatenate(stars), np.concatenate(Fs)


def fisher_fixture(shapes):
    """Per-parameter Fisher tensors U[0, 50) from the counter-based generator (SURVEY.md §8 F3 fixtures)."""
    return [rng.uniform(int(np.prod(s)), 9000 + 17 * i, 0.0, 50.0).reshape(s) for i, s in enumerate(shapes)]


def eval_loaders():
    """Member / non-member style batches for the SVC_MIA fixture: "test" images are darker and noisier copies of the
    generator's images, so the model's confidence / entropy features separate the two populations."""
    def mk(nb, seed, scale):
        out = []
        for i, (x, y) in enumerate(tiny_batches(nb, 16, seed)):
            s = scale if isinstance(scale, float) else scale[i % len(scale)]
            out.append((torch.from_numpy((x * np.float32(s)).astype(np.float32)), torch.from_numpy(y)))
        return out
    return dict(shadow_train=mk(6, 2000, 1.0), shadow_test=mk(4, 2100, 0.15), target_test=mk(3, 2200, (0.15, 1.0, 0.15)))


# ------------------------------------------------------------------ SD script-glue fixtures (sd_glue.npz)
def sd_glue_config():
    """250,372-parameter U-Net (one resolution level: ResBlocks + SpatialTransformers, no down/up-sampling) — small
    enough that the reference's full accumulator fits a fixture; the two-level tiny config pins the module itself."""
    return dict(sd_tiny_config(), channel_mult=(1,), attention_resolutions=(1,))


SD_GLUE_PROMPTS = ("a photo of a nude person", "a photo of a person wearing clothes", "")


def sd_glue_contexts():
    """prompt -> fixed (7, 24) context embedding (stands in for the frozen CLIP encoder)."""
    return {p: rng.normal(7 * 24, 9400 + i).reshape(7, 24) for i, p in enumerate(SD_GLUE_PROMPTS)}


def sd_glue_batches():
    """(class-style forget batches unused here, ..., forget 'images', remain 'images'): latents (4, 4, 8, 8) handed to
    the reference scripts as their image tensors (B, C, H, W); 3 forget and 2 remain batches (the loop must wrap)."""
    mk = lambda seed: torch.from_numpy(rng.normal(4 * 4 * 8 * 8, seed).reshape(4, 4, 8, 8))
    forget = [mk(9500 + i) for i in range(3)]
    remain = [mk(9600 + i) for i in range(2)]
    return None, None, forget, remain


def sd_glue_class_contexts():
    """descriptions[i] -> context for the class-conditioned scripts (generate_mask / certain_label / proximal_gradient
    goldens): the nude-prompt context shifted by 0.01 * i (fp32)."""
    base = sd_glue_contexts()[SD_GLUE_PROMPTS[0]]
    return [(base + np.float32(0.01) * np.float32(i)).astype(np.float32) if i else base.copy() for i in range(10)]


def sd_glue_loaders(kind, remain_labels=None):
    """This build's loader format (latents + context embeddings, SD/ldm_lite.py) for the data the reference scripts
    saw in tests/golden/make_golden_sd_glue.py.
      "nsfw_mask"   (z, ctx nude, ctx "")                "nsfw"  forget (z, nude, clothes), remain (z, clothes)
      "class_mask"  (z, ctx class 3, ctx "")             "class" forget (z, class 3, class 4), remain (z, class labels)"""
    _, _, forget, remain = sd_glue_batches()
    ctx = {k: torch.from_numpy(v) for k, v in sd_glue_contexts().items()}
    nude, wear, null = (ctx[p] for p in SD_GLUE_PROMPTS)
    cls = [torch.from_numpy(c) for c in sd_glue_class_contexts()]
    rep = lambda c, n: c.unsqueeze(0).repeat(n, 1, 1)
    if kind == "nsfw_mask":
        return [(z, rep(nude, len(z)), rep(null, len(z))) for z in forget]
    if kind == "class_mask":
        return [(z, rep(cls[3], len(z)), rep(null, len(z))) for z in forget]
    if kind == "nsfw":
        return ([(z, rep(nude, len(z)), rep(wear, len(z))) for z in forget], [(z, rep(wear, len(z))) for z in remain])
    if kind == "class":
        rl = [(z, torch.stack([cls[int(l)] for l in labs])) for z, labs in zip(remain, remain_labels)]
        return [(z, rep(cls[3], len(z)), rep(cls[4], len(z))) for z in forget], rl
    raise ValueError(kind)


class replay_draws:
    """torch.randint / torch.randn_like return the recorded draws in call order (moved to the requested device)."""

    def __init__(self, randint, randn):
        self.randint, self.randn = [np.asarray(v) for v in randint], [np.asarray(v) for v in randn]

    def __enter__(self):
        self.real = (torch.randint, torch.randn_like)
        ri, rn = self.randint, self.randn

        def randint(*a, device=None, **k):
            return torch.from_numpy(ri.pop(0).astype(np.int64)).to(device or "cpu")

        def randn_like(x, **k):
            return torch.from_numpy(rn.pop(0).astype(np.float32)).to(x.device).reshape(x.shape)

        torch.randint, torch.randn_like = randint, randn_like
        return self

    def __exit__(self, et, ev, tb):
        torch.randint, torch.randn_like = self.real
        if et is None:
            assert not self.randint and not self.randn, "recorded draws left over: the call order differs"
