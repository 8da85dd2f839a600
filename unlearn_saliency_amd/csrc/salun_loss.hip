// salun_loss.hip — K6 (q-sample, squared-error loss + gradient), K0 (device-resident
// image batch assembly) and the counter-based synthetic-input generators.
// gfx950 / CDNA4, compiled with -ffp-contract=off.
//
// K6 tensors are tiny next to the parameter arena (128 x 3 x 32 x 32 = 393,216
// floats for DDPM, 8 x 4 x 64 x 64 = 131,072 for SD latents): the kernels are
// launch/latency bound, so the point is one pass + a fixed reduction tree
// (LDS-staged partial sums, wavefront __shfl reductions) instead of ~8 ATen launches.
#include "salun_common.h"

namespace {

constexpr int SEG = 1024;  // floats per (sample, segment) work item: 256 lanes x float4

// ------------------------------------------------------------------ q-sample
__global__ __launch_bounds__(SALUN_BLOCK) void k_qsample(const float *__restrict__ x0, const float *__restrict__ e,
                                                         const float *__restrict__ sa, const float *__restrict__ sb,
                                                         const long long *__restrict__ t, int64_t T,
                                                         float *__restrict__ xt, int64_t B, int64_t chw, int vec) {
  const int64_t segs = (chw + SEG - 1) / SEG;
  for (int64_t w = blockIdx.x; w < B * segs; w += gridDim.x) {
    const int64_t b = w / segs, s = w - b * segs;
    long long tb = t[b];
    if (tb < 0) tb = 0;
    if (tb >= T) tb = T - 1;
    const float a = sa[tb], c = sb[tb];
    const int64_t off = b * chw, j0 = s * SEG + (int64_t)threadIdx.x * 4;
    if (vec && j0 + 3 < chw) {
      const float4 xv = *reinterpret_cast<const float4 *>(x0 + off + j0);
      const float4 ev = *reinterpret_cast<const float4 *>(e + off + j0);
      float4 r;
      r.x = (xv.x * a) + (ev.x * c);
      r.y = (xv.y * a) + (ev.y * c);
      r.z = (xv.z * a) + (ev.z * c);
      r.w = (xv.w * a) + (ev.w * c);
      *reinterpret_cast<float4 *>(xt + off + j0) = r;
    } else {
      for (int q = 0; q < 4; ++q)
        if (j0 + q < chw) xt[off + j0 + q] = (x0[off + j0 + q] * a) + (e[off + j0 + q] * c);
    }
  }
}

// ------------------------------------------------- squared error: partial sums
// Work item = (sample b, segment s): sum over <= 1024 elements of (a-b)^2 in fp32
// per lane (4 lanes-worth), folded in fp64 across the workgroup -> partial[b*segs+s].
__global__ __launch_bounds__(SALUN_BLOCK) void k_sqerr_partial(const float *__restrict__ a, const float *__restrict__ b,
                                                               int64_t B, int64_t chw, float neg2coef,
                                                               double *__restrict__ partial,
                                                               float *__restrict__ dloss, int vec) {
  __shared__ double lds[4];
  const int64_t segs = (chw + SEG - 1) / SEG;
  for (int64_t w = blockIdx.x; w < B * segs; w += gridDim.x) {
    const int64_t bi = w / segs, s = w - bi * segs;
    const int64_t off = bi * chw, j0 = s * SEG + (int64_t)threadIdx.x * 4;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec && j0 + 3 < chw) {
      const float4 av = *reinterpret_cast<const float4 *>(a + off + j0);
      const float4 bv = *reinterpret_cast<const float4 *>(b + off + j0);
      d[0] = av.x - bv.x; d[1] = av.y - bv.y; d[2] = av.z - bv.z; d[3] = av.w - bv.w;
      if (dloss) {
        float4 g;
        g.x = neg2coef * d[0]; g.y = neg2coef * d[1]; g.z = neg2coef * d[2]; g.w = neg2coef * d[3];
        *reinterpret_cast<float4 *>(dloss + off + j0) = g;
      }
    } else {
      for (int q = 0; q < 4; ++q)
        if (j0 + q < chw) {
          d[q] = a[off + j0 + q] - b[off + j0 + q];
          if (dloss) dloss[off + j0 + q] = neg2coef * d[q];
        }
    }
    const double sq = ((double)(d[0] * d[0]) + (double)(d[1] * d[1])) + ((double)(d[2] * d[2]) + (double)(d[3] * d[3]));
    const double tot = salun_block_sum(sq, lds);
    if (threadIdx.x == 0) partial[w] = tot;
  }
}

// One workgroup: per-sample sums (segments folded in index order), then the batch
// sum (samples folded in a fixed tree), times coef.
__global__ __launch_bounds__(SALUN_BLOCK) void k_sqerr_final(const double *__restrict__ partial, int64_t B, int64_t segs,
                                                             double coef, float *__restrict__ loss,
                                                             float *__restrict__ per_sample) {
  __shared__ double lds[4];
  double acc = 0.0;
  for (int64_t b = threadIdx.x; b < B; b += SALUN_BLOCK) {
    double s = 0.0;
    for (int64_t j = 0; j < segs; ++j) s += partial[b * segs + j];
    if (per_sample) per_sample[b] = (float)s;
    acc += s;
  }
  const double tot = salun_block_sum(acc, lds);
  if (threadIdx.x == 0) *loss = (float)(tot * coef);
}

// ------------------------------------------------------------- image batches
// One lane per output pixel (b, y, x): reads the 3 (C) interleaved source bytes of
// the cropped/flipped source pixel, writes C planar fp32 values (coalesced per plane).
__global__ __launch_bounds__(SALUN_BLOCK) void k_image_batch(const uint8_t *__restrict__ data,
                                                             const long long *__restrict__ idx,
                                                             const int *__restrict__ crop,
                                                             const uint8_t *__restrict__ flip,
                                                             float *__restrict__ out, int64_t B, int H, int W, int C,
                                                             int pad) {
  const int64_t hw = (int64_t)H * W;
  const int64_t total = B * hw;
  for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * SALUN_BLOCK) {
    const int64_t b = i / hw;
    const int r = (int)(i - b * hw);
    const int y = r / W, x = r - y * W;
    int dy = pad, dx = pad;
    if (crop) { dy = crop[2 * b]; dx = crop[2 * b + 1]; }
    const int xs = (flip && flip[b]) ? (W - 1 - x) : x;
    const int sy = y + dy - pad, sx = xs + dx - pad;
    const bool inside = (sy >= 0) && (sy < H) && (sx >= 0) && (sx < W);
    const uint8_t *src = data + ((int64_t)idx[b] * hw + (int64_t)sy * W + sx) * C;
    float *dst = out + b * C * hw + r;
    for (int c = 0; c < C; ++c) dst[c * hw] = inside ? ((float)src[c] / 255.0f) : 0.0f;
  }
}

// ------------------------------------------------------------ generators
__global__ __launch_bounds__(SALUN_BLOCK) void k_fill_uniform(float *out, int64_t n, uint64_t seed, float lo, float span) {
  for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
    out[i] = lo + (span * salun_u01(seed, (uint64_t)i));
}
__global__ __launch_bounds__(SALUN_BLOCK) void k_fill_normal(float *out, int64_t n, uint64_t seed, float mean, float std) {
  for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
    out[i] = mean + (std * salun_ih12(seed, (uint64_t)i));
}
__global__ __launch_bounds__(SALUN_BLOCK) void k_fill_u8(uint8_t *out, int64_t n, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
    out[i] = (uint8_t)(salun_splitmix64(seed + ((uint64_t)i >> 3)) >> (8 * (i & 7)));
}

// ------------------------------------------------------------ counter-based dropout
// keep(gi) for the GLOBAL element index gi = (sample_offset + s) * chw + j of the batch: 24 bits of
// splitmix64(key + (gi >> 1)) — the high 24 for even gi, bits [8, 32) for odd gi — compared with thr = round(p * 2^24).
// A function of (key, global sample index, position) only: a rank holding samples [lo, hi) of a global batch produces
// exactly rows lo..hi-1 of the single-process result, no mask is stored (backward re-derives it from the same key),
// and the comparison is integer, so the CPU oracle agrees bit for bit.
__device__ __forceinline__ uint32_t drop_bits(uint64_t key, uint64_t gi) {
  const uint64_t r = salun_splitmix64(key + (gi >> 1));
  return (gi & 1ull) ? (uint32_t)((r >> 8) & 0xFFFFFFull) : (uint32_t)(r >> 40);
}
__global__ __launch_bounds__(SALUN_BLOCK) void k_dropout(const float *__restrict__ x, float *__restrict__ y, int64_t total4,
                                                         int64_t total, uint64_t base, uint64_t seed,
                                                         const uint64_t *__restrict__ seed_dev, uint32_t thr,
                                                         float scale, int vec) {
  const uint64_t key = seed + (seed_dev ? *seed_dev : 0ull);
  if (vec) {  // total % 4 == 0, base even, 16-byte aligned: one splitmix64 per pair of elements
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < total4; i += (int64_t)gridDim.x * SALUN_BLOCK) {
      const uint64_t gi = base + 4ull * (uint64_t)i;
      const uint64_t r0 = salun_splitmix64(key + (gi >> 1)), r1 = salun_splitmix64(key + (gi >> 1) + 1ull);
      const float4 v = *reinterpret_cast<const float4 *>(x + 4 * i);
      float4 o;
      o.x = ((uint32_t)(r0 >> 40) >= thr) ? v.x * scale : 0.0f;
      o.y = ((uint32_t)((r0 >> 8) & 0xFFFFFFull) >= thr) ? v.y * scale : 0.0f;
      o.z = ((uint32_t)(r1 >> 40) >= thr) ? v.z * scale : 0.0f;
      o.w = ((uint32_t)((r1 >> 8) & 0xFFFFFFull) >= thr) ? v.w * scale : 0.0f;
      *reinterpret_cast<float4 *>(y + 4 * i) = o;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * SALUN_BLOCK)
      y[i] = (drop_bits(key, base + (uint64_t)i) >= thr) ? x[i] * scale : 0.0f;
  }
}
__global__ void k_u64_add(uint64_t *p, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc;
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT int salun_qsample(const float *x0, const float *e, const float *sqrt_ab, const float *sqrt_1mab,
                               const int64_t *t, int64_t T, float *xt, int64_t B, int64_t chw,
                               salun_stream_t stream) {
  if (B < 0 || chw < 0 || T < 1) return SALUN_EINVAL;
  if (B == 0 || chw == 0) return SALUN_OK;
  if (!x0 || !e || !sqrt_ab || !sqrt_1mab || !t || !xt) return SALUN_EINVAL;
  const int64_t items = B * ((chw + SEG - 1) / SEG);
  const int vec = salun_aligned16(x0) && salun_aligned16(e) && salun_aligned16(xt) && (chw % 4 == 0);
  hipLaunchKernelGGL(k_qsample, dim3(salun_grid_for(items, 1)), dim3(SALUN_BLOCK), 0, salun_hip_stream(stream), x0, e,
                     sqrt_ab, sqrt_1mab, reinterpret_cast<const long long *>(t), T, xt, B, chw, vec);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT size_t salun_sqerr_workspace_bytes(int64_t B, int64_t chw) {
  if (B < 0 || chw < 0) return 0;
  return sizeof(double) * (size_t)(B * ((chw + SEG - 1) / SEG) + 1);
}

SALUN_EXPORT int salun_sqerr_loss(const float *a, const float *b, int64_t B, int64_t chw, double coef, float *loss,
                                  float *per_sample, float *dloss_db, void *ws, size_t ws_bytes,
                                  salun_stream_t stream) {
  if (B < 1 || chw < 1 || !a || !b || !loss || !ws) return SALUN_EINVAL;
  if (ws_bytes < salun_sqerr_workspace_bytes(B, chw)) return SALUN_ENOSPC;
  hipStream_t st = salun_hip_stream(stream);
  const int64_t segs = (chw + SEG - 1) / SEG;
  const int vec = salun_aligned16(a) && salun_aligned16(b) && (!dloss_db || salun_aligned16(dloss_db)) && (chw % 4 == 0);
  double *partial = static_cast<double *>(ws);
  hipLaunchKernelGGL(k_sqerr_partial, dim3(salun_grid_for(B * segs, 1)), dim3(SALUN_BLOCK), 0, st, a, b, B, chw,
                     (float)(-2.0 * coef), partial, dloss_db, vec);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sqerr_final, dim3(1), dim3(SALUN_BLOCK), 0, st, partial, B, segs, coef, loss, per_sample);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_image_batch(const uint8_t *data, const int64_t *idx, const int32_t *crop, const uint8_t *flip,
                                   float *out, int64_t B, int H, int W, int C, int pad, salun_stream_t stream) {
  if (B < 0 || H < 1 || W < 1 || C < 1 || pad < 0) return SALUN_EINVAL;
  if (B == 0) return SALUN_OK;
  if (!data || !idx || !out) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_image_batch, dim3(salun_grid_for(B * H * W, SALUN_BLOCK)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), data, reinterpret_cast<const long long *>(idx), crop, flip, out, B, H,
                     W, C, pad);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_fill_uniform(float *out, int64_t n, uint64_t seed, double lo, double hi, salun_stream_t stream) {
  if (n < 0 || (n > 0 && !out)) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipLaunchKernelGGL(k_fill_uniform, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), out, n, seed, (float)lo, (float)(hi - lo));
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
SALUN_EXPORT int salun_fill_normal(float *out, int64_t n, uint64_t seed, double mean, double std, salun_stream_t stream) {
  if (n < 0 || (n > 0 && !out)) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipLaunchKernelGGL(k_fill_normal, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), out, n, seed, (float)mean, (float)std);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
SALUN_EXPORT int salun_fill_u8(uint8_t *out, int64_t n, uint64_t seed, salun_stream_t stream) {
  if (n < 0 || (n > 0 && !out)) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipLaunchKernelGGL(k_fill_u8, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), out, n, seed);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_dropout(const float *x, float *y, int64_t n_samples, int64_t chw, int64_t sample_offset, double p,
                               uint64_t seed, const uint64_t *seed_dev, salun_stream_t stream) {
  if (n_samples < 0 || chw < 0 || sample_offset < 0 || !(p >= 0.0) || !(p < 1.0)) return SALUN_EINVAL;
  if (n_samples == 0 || chw == 0) return SALUN_OK;
  if (!x || !y) return SALUN_EINVAL;
  const int64_t total = n_samples * chw;
  const uint64_t base = (uint64_t)sample_offset * (uint64_t)chw;
  const uint32_t thr = salun_dropout_threshold(p);
  const float scale = (float)(1.0 / (1.0 - p));
  const int vec = salun_aligned16(x) && salun_aligned16(y) && (total % 4 == 0) && ((base & 1ull) == 0);
  const int64_t items = vec ? total / 4 : total;
  hipLaunchKernelGGL(k_dropout, dim3(salun_grid_for(items, SALUN_BLOCK * 2)), dim3(SALUN_BLOCK), 0, salun_hip_stream(stream),
                     x, y, total / 4, total, base, seed, seed_dev, thr, scale, vec);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_u64_add(uint64_t *value, uint64_t inc, salun_stream_t stream) {
  if (!value) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_u64_add, dim3(1), dim3(64), 0, salun_hip_stream(stream), value, inc);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ---------------------------------------------------------------- identification
SALUN_EXPORT int salun_version(void) { return 100; }  // 0.1.0
SALUN_EXPORT const char *salun_arch(void) { return "gfx950"; }
SALUN_EXPORT const char *salun_strerror(int code) {
  switch (code) {
    case SALUN_OK: return "ok";
    case SALUN_EINVAL: return "invalid argument";
    case SALUN_ENOSPC: return "workspace too small";
    case SALUN_EIO: return "HIP launch/runtime error";
    default: return "unknown salun error";
  }
}
