/*
 * salun_oracle.c — CPU restatement of the reference's hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under unlearn_saliency_amd/ may import, link
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * Plain scalar C, one loop per function, in the operation order the reference's
 * PyTorch calls perform (citations are <file>:<line> in OPTML-Group/Unlearn-Saliency).
 * Compiled with -ffp-contract=off so that a*b+c is two roundings unless written
 * as fmaf() — the same contract as the HIP kernels, which is what makes the
 * element-wise comparisons bit-exact.
 *
 * Parity pinning: the reference has no tests or golden vectors of its own
 * (SURVEY.md §4).  This oracle is pinned against outputs of the reference's own
 * functions imported in the build container — tests/golden/make_golden.py writes
 * them, tests/test_oracle_vs_golden.py checks them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ helpers -- */
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }

/* torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6),
 * clamped to 1.0 (DDPM/runners/diffusion.py:582-587,985-990). */
ORACLE_EXPORT float oracle_clip_coef(float sqnorm, float max_norm) {
  float total = sqrtf(sqnorm);
  float c = max_norm / (total + 1e-6f);
  return c > 1.0f ? 1.0f : c;
}

/* -------------------------------------------------------------------- K1 ----
 * gradients[name] += param.grad.data   — Classification/generate_mask.py:41-44;
 * with the clip coefficient folded in for DDPM/runners/diffusion.py:985-996
 * (clip_grad_norm_ scales grad in place, then `gradients[name] += gradient`). */
ORACLE_EXPORT void oracle_saliency_accumulate(float *acc, const float *g, float scale, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    float t = g[i] * scale;
    acc[i] = acc[i] + t;
  }
}

/* -------------------------------------------------------------------- K2 ----
 * Classification/generate_mask.py:46-80 (twins: DDPM/runners/diffusion.py:998-1037,
 * SD/train-scripts/generate_mask.py:71-106):
 *   all_elements = -cat(abs(g));  positions = argsort(all_elements);
 *   ranks = argsort(positions);   mask = ranks < threshold_index
 * restated literally with a STABLE argsort (ties keep flat-index order; NaN sorts
 * last, as torch.sort does).  Sorting key: ascending (descending-|x| code, index). */
static void radix_sort_u64(uint64_t *a, uint64_t *tmp, int64_t n) {
  /* the input is in flat-index order and every pass is stable, so only the code
   * (bits 32..63) needs sorting for ties to keep index order */
  for (int pass = 2; pass < 4; ++pass) {
    const int shift = 16 * pass;
    int64_t *cnt = (int64_t *)calloc(65537, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) cnt[((a[i] >> shift) & 0xFFFF) + 1]++;
    for (int j = 0; j < 65536; ++j) cnt[j + 1] += cnt[j];
    for (int64_t i = 0; i < n; ++i) tmp[cnt[(a[i] >> shift) & 0xFFFF]++] = a[i];
    memcpy(a, tmp, (size_t)n * sizeof(uint64_t));
    free(cnt);
  }
}

/* returns 0 on success, -1 if n does not fit the 32-bit index packing / no memory */
ORACLE_EXPORT int oracle_mask_topk(const float *acc, int64_t n, const int64_t *ks, int nk, uint8_t *const *masks_out) {
  if (n < 0 || n >= ((int64_t)1 << 32)) return -1;
  if (n == 0) return 0;
  uint64_t *a = (uint64_t *)malloc((size_t)n * 8), *tmp = (uint64_t *)malloc((size_t)n * 8);
  uint32_t *ranks = (uint32_t *)malloc((size_t)n * 4);
  if (!a || !tmp || !ranks) { free(a); free(tmp); free(ranks); return -1; }
  for (int64_t i = 0; i < n; ++i) {
    /* order of -|x| ascending == |x| descending; NaN last */
    uint32_t b = f2u(acc[i]) & 0x7FFFFFFFu;
    uint32_t code = (b > 0x7F800000u) ? 0xFFFFFFFFu : (0x7F800000u - b);
    a[i] = ((uint64_t)code << 32) | (uint64_t)(uint32_t)i;
  }
  radix_sort_u64(a, tmp, n);                              /* positions = argsort(all_elements) */
  for (int64_t r = 0; r < n; ++r) ranks[(uint32_t)a[r]] = (uint32_t)r; /* ranks = argsort(positions) */
  for (int j = 0; j < nk; ++j) {
    const int64_t k = ks[j];
    uint8_t *m = masks_out[j];
    for (int64_t i = 0; i < n; ++i) m[i] = ((int64_t)ranks[i] < k) ? 1 : 0; /* ranks < threshold_index */
  }
  free(a); free(tmp); free(ranks);
  return 0;
}

/* ----------------------------------------------------------------- K3+K4 ----
 * _apply_mask_to_grads (Classification/unlearn/RL.py:11-14)
 *   -> torch.optim.SGD.step, momentum mu, weight_decay wd, dampening 0, no nesterov
 *      (unlearn/impl.py:68-73)
 *   -> _restore_masked_params (RL.py:17-34): p = p*m + theta0*(1-m); buf *= m.
 * The reference sequence on one element, with theta0 passed explicitly.  (The HIP
 * kernel never reads theta0: where m==0 the restore makes p == theta0 before and
 * after every step, so it leaves p untouched.) */
ORACLE_EXPORT void oracle_masked_sgd_step_reference(float *p, const float *g, float *buf, const uint8_t *m,
                                                    const float *theta0, double lr, double mu, double wd,
                                                    int first_step, int64_t n) {
  const float flr = (float)(-lr), fmu = (float)mu, fwd = (float)wd;
  for (int64_t i = 0; i < n; ++i) {
    const float mf = m ? (m[i] ? 1.0f : 0.0f) : 1.0f;
    float gi = g[i] * mf;                                 /* param.grad *= mask[name] */
    float d = (wd != 0.0) ? fmaf(fwd, p[i], gi) : gi;     /* grad.add(param, alpha=wd) */
    float nb = d;
    if (mu != 0.0) {
      nb = first_step ? d : (fmu * buf[i]) + d;           /* buf.mul_(mu).add_(grad) */
      buf[i] = nb;
    }
    float pn = fmaf(flr, nb, p[i]);                       /* param.add_(buf, alpha=-lr) */
    if (m) {                                              /* restore */
      const float inv = 1.0f - mf;
      pn = (pn * mf) + (theta0[i] * inv);
      if (mu != 0.0) buf[i] = buf[i] * mf;
    }
    p[i] = pn;
  }
}

/* The fused form the HIP kernel implements (include/salun.h, salun_masked_sgd_step). */
ORACLE_EXPORT void oracle_masked_sgd_step(float *p, const float *g, float *buf, const uint8_t *m, double lr,
                                          double mu, double wd, int first_step, int64_t n) {
  const float flr = (float)(-lr), fmu = (float)mu, fwd = (float)wd;
  for (int64_t i = 0; i < n; ++i) {
    if (!m || m[i]) {
      float d = (wd != 0.0) ? fmaf(fwd, p[i], g[i]) : g[i];
      float nb = d;
      if (mu != 0.0) {
        nb = first_step ? d : (fmu * buf[i]) + d;
        buf[i] = nb;
      }
      p[i] = fmaf(flr, nb, p[i]);
    } else if (mu != 0.0) {
      buf[i] = 0.0f;
    }
  }
}

/* -------------------------------------------------------------------- K5 ----
 * Squared global gradient norm (the quantity under clip_grad_norm_'s sqrt). */
ORACLE_EXPORT float oracle_grad_sqnorm(const float *g, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += (double)g[i] * (double)g[i];
  return (float)s;
}

/* clip_grad_norm_ -> `param.grad *= mask[name]` -> torch.optim.Adam.step (amsgrad off)
 * DDPM/runners/diffusion.py:582-593 + DDPM/functions/__init__.py:9-18;
 * SD/train-scripts/random_label.py:129-139.  Order of torch's _single_tensor_adam
 * (torch 2.0.x, the version the reference pins): exp_avg.mul_(b1).add_(g, alpha=1-b1);
 * exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2); denom = sqrt(v)/sqrt(bc2) + eps;
 * param.addcdiv_(exp_avg, denom, value=-lr/bc1). */
ORACLE_EXPORT void oracle_masked_adam_step(float *p, const float *g, float *m1, float *v, const uint8_t *mask,
                                           double gscale, double lr, double b1, double b2, double eps, double wd,
                                           int step, int64_t n) {
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  const float s = (float)gscale, fb1 = (float)b1, fb2 = (float)b2, omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
  const float feps = (float)eps, fwd = (float)wd, bc2s = (float)sqrt(bc2), nss = (float)(-(lr / bc1));
  for (int64_t i = 0; i < n; ++i) {
    const float mf = mask ? (mask[i] ? 1.0f : 0.0f) : 1.0f;
    float ge = (g[i] * s) * mf;
    if (wd != 0.0) ge = fmaf(fwd, p[i], ge);
    m1[i] = (fb1 * m1[i]) + (omb1 * ge);
    v[i] = (fb2 * v[i]) + ((omb2 * ge) * ge);
    const float den = (sqrtf(v[i]) / bc2s) + feps;
    p[i] = p[i] + (nss * (m1[i] / den));
  }
}

/* -------------------------------------------------------------------- K6 ----
 * x = x0 * a.sqrt() + e * (1.0 - a).sqrt()   — DDPM/functions/losses.py:31-32. */
ORACLE_EXPORT void oracle_qsample(const float *x0, const float *e, const float *sqrt_ab, const float *sqrt_1mab,
                                  const int64_t *t, int64_t T, float *xt, int64_t B, int64_t chw) {
  for (int64_t b = 0; b < B; ++b) {
    int64_t tb = t[b] < 0 ? 0 : (t[b] >= T ? T - 1 : t[b]);
    const float a = sqrt_ab[tb], c = sqrt_1mab[tb];
    for (int64_t j = 0; j < chw; ++j) xt[b * chw + j] = (x0[b * chw + j] * a) + (e[b * chw + j] * c);
  }
}

/* (e - output).square().sum(dim=(1,2,3)).mean(dim=0)  — DDPM/functions/losses.py:34-37
 * [coef = 1/B]; nn.MSELoss(pseudo, output) — runners/diffusion.py:570 [coef = 1/(B*chw)];
 * dloss_db = d loss / d b = -2*coef*(a-b). */
ORACLE_EXPORT void oracle_sqerr_loss(const float *a, const float *b, int64_t B, int64_t chw, double coef, float *loss,
                                     float *per_sample, float *dloss_db) {
  const float n2c = (float)(-2.0 * coef);
  double tot = 0.0;
  for (int64_t s = 0; s < B; ++s) {
    double acc = 0.0;
    for (int64_t j = 0; j < chw; ++j) {
      const float d = a[s * chw + j] - b[s * chw + j];
      acc += (double)(d * d);
      if (dloss_db) dloss_db[s * chw + j] = n2c * d;
    }
    if (per_sample) per_sample[s] = (float)acc;
    tot += acc;
  }
  *loss = (float)(tot * coef);
}

/* -------------------------------------------------------------------- K7 ----
 * fisher_dict[name] += tmp**2 / len(dataset); tmp = 0  — DDPM/runners/diffusion.py:176-183. */
ORACLE_EXPORT void oracle_fim_square_accumulate(float *F, float *tmp, double n_data, int64_t n) {
  const float nd = (float)n_data;
  for (int64_t i = 0; i < n; ++i) {
    F[i] = F[i] + ((tmp[i] * tmp[i]) / nd);
    tmp[i] = 0.0f;
  }
}

/* -------------------------------------------------------------------- K0 ----
 * torchvision RandomCrop(32, padding=4) + RandomHorizontalFlip + ToTensor
 * (Classification/dataset.py:549-555) with the random draws passed in. */
ORACLE_EXPORT void oracle_image_batch(const uint8_t *data, const int64_t *idx, const int32_t *crop,
                                      const uint8_t *flip, float *out, int64_t B, int H, int W, int C, int pad) {
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          int dy = crop ? crop[2 * b] : pad, dx = crop ? crop[2 * b + 1] : pad;
          int xs = (flip && flip[b]) ? (W - 1 - x) : x;
          int sy = y + dy - pad, sx = xs + dx - pad;
          float v = 0.0f;
          if (sy >= 0 && sy < H && sx >= 0 && sx < W)
            v = (float)data[((idx[b] * H + sy) * (int64_t)W + sx) * C + c] / 255.0f;
          out[((b * C + c) * H + y) * (int64_t)W + x] = v;
        }
}

/* ------------------------------------------------- counter-based generators ---
 * Identical to unlearn_saliency_amd/csrc/salun_common.h (integer arithmetic only). */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
ORACLE_EXPORT void oracle_fill_uniform(float *out, int64_t n, uint64_t seed, double lo, double hi) {
  const float flo = (float)lo, span = (float)(hi - lo);
  for (int64_t i = 0; i < n; ++i)
    out[i] = flo + (span * ((float)(splitmix64(seed + (uint64_t)i) >> 40) * (1.0f / 16777216.0f)));
}
ORACLE_EXPORT void oracle_fill_normal(float *out, int64_t n, uint64_t seed, double mean, double std) {
  const float fm = (float)mean, fs = (float)std;
  for (int64_t i = 0; i < n; ++i) {
    int32_t s = 0;
    for (int j = 0; j < 3; ++j) {
      uint64_t r = splitmix64(seed + 3ull * (uint64_t)i + (uint64_t)j);
      s += (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) + (int32_t)((r >> 32) & 0xFFFF) + (int32_t)((r >> 48) & 0xFFFF);
    }
    out[i] = fm + (fs * ((float)(s - 393210) * (1.0f / 65536.0f)));
  }
}
ORACLE_EXPORT void oracle_fill_u8(uint8_t *out, int64_t n, uint64_t seed) {
  for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)(splitmix64(seed + ((uint64_t)i >> 3)) >> (8 * (i & 7)));
}

/* Counter-based dropout — the CPU statement of csrc/salun_loss.hip::k_dropout (which replaces nn.Dropout of the
 * reference's ResnetBlock, DDPM/models/diffusion.py:108,124, under per-GPU sharding of the batch):
 * keep(gi) = 24 bits of splitmix64(key + (gi >> 1)) >= round(p * 2^24), gi = (sample_offset + s) * chw + j. */
ORACLE_EXPORT void oracle_dropout(const float *x, float *y, int64_t n_samples, int64_t chw, int64_t sample_offset,
                                  double p, uint64_t key) {
  double t = p * 16777216.0 + 0.5;
  if (t < 0.0) t = 0.0;
  if (t > 16777216.0) t = 16777216.0;
  const uint32_t thr = (uint32_t)t;
  const float scale = (float)(1.0 / (1.0 - p));
  const uint64_t base = (uint64_t)sample_offset * (uint64_t)chw;
  for (int64_t i = 0; i < n_samples * chw; ++i) {
    const uint64_t gi = base + (uint64_t)i;
    const uint64_t r = splitmix64(key + (gi >> 1));
    const uint32_t bits = (gi & 1ull) ? (uint32_t)((r >> 8) & 0xFFFFFFull) : (uint32_t)(r >> 40);
    y[i] = bits >= thr ? x[i] * scale : 0.0f;
  }
}
