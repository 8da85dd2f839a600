// salun_update.hip — HBM-bound element-wise kernels over the flat parameter arena:
//   K1 saliency accumulate, K3+K4 masked SGD-momentum, K5 grad sq-norm + masked Adam,
//   K7 Fisher square-accumulate.   gfx950 / CDNA4, compiled with -ffp-contract=off.
//
// Streaming shape shared by all of them: 256-thread workgroups (one wave per SIMD),
// each workgroup walks "tiles" of UNROLL x 256 float4 (= 4096 floats for UNROLL 4)
// grid-stride; inside a tile lane l of sub-vector u touches float4 index
// tile*UNROLL*256 + u*256 + l, so every global_load_dwordx4 of a wave covers one
// contiguous 1 KiB and all UNROLL loads of every stream are issued before the first
// use (12-16 x 16 B in flight per lane).  The u8 mask travels as one dword per float4.
#include "salun_common.h"

namespace {

constexpr int UNROLL = 4;
constexpr int TILE_VEC = UNROLL * SALUN_BLOCK;  // float4 per tile
constexpr int TILE_ELEMS = TILE_VEC * 4;        // floats per tile

__device__ __forceinline__ float4 ld4(const float *p, int64_t v) {
  return reinterpret_cast<const float4 *>(p)[v];
}
// Streamed-once operand (the gradient): bypass-friendly non-temporal load.
__device__ __forceinline__ float4 ld4_nt(const float *p, int64_t v) {
  const float4 *q = reinterpret_cast<const float4 *>(p) + v;
  float4 r;
  r.x = __builtin_nontemporal_load(&q->x);
  r.y = __builtin_nontemporal_load(&q->y);
  r.z = __builtin_nontemporal_load(&q->z);
  r.w = __builtin_nontemporal_load(&q->w);
  return r;
}
__device__ __forceinline__ void st4(float *p, int64_t v, float4 x) {
  reinterpret_cast<float4 *>(p)[v] = x;
}
__device__ __forceinline__ uint32_t ldm(const uint8_t *m, int64_t v) {
  return reinterpret_cast<const uint32_t *>(m)[v];
}

// ------------------------------------------------------------------------ K1 ----
struct AccumArgs {
  float *acc;
  const float *g;
  const float *sqnorm;
  float scale;
  float max_norm;
  int64_t n;
};

__device__ __forceinline__ float accum_elem(float a, float g, float s) { return a + (g * s); }

template <bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_saliency_accumulate(AccumArgs a) {
  const float s = a.sqnorm ? salun_clip_coef(*a.sqnorm, a.max_norm) : a.scale;
  if (VEC) {
    const int64_t nvec = a.n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 av[UNROLL], gv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          av[u] = ld4(a.acc, v);
          gv[u] = ld4_nt(a.g, v);
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          float4 r;
          r.x = accum_elem(av[u].x, gv[u].x, s);
          r.y = accum_elem(av[u].y, gv[u].y, s);
          r.z = accum_elem(av[u].z, gv[u].z, s);
          r.w = accum_elem(av[u].w, gv[u].w, s);
          st4(a.acc, v, r);
        }
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < a.n) a.acc[i] = accum_elem(a.acc[i], a.g[i], s);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < a.n;
         i += (int64_t)gridDim.x * SALUN_BLOCK)
      a.acc[i] = accum_elem(a.acc[i], a.g[i], s);
  }
}

// --------------------------------------------------------------------- K3+K4 ----
struct SgdArgs {
  float *p;
  const float *g;
  float *buf;
  const uint8_t *m;
  float neg_lr, mu, wd;
  int first_step;
  int64_t n;
};

template <bool HAS_WD, bool HAS_MOM>
__device__ __forceinline__ void sgd_elem(float &p, float g, float &b, bool on, const SgdArgs &a) {
  if (on) {
    const float d = HAS_WD ? __builtin_fmaf(a.wd, p, g) : g;
    float nb = d;
    if (HAS_MOM) {
      nb = a.first_step ? d : (a.mu * b) + d;
      b = nb;
    }
    p = __builtin_fmaf(a.neg_lr, nb, p);
  } else if (HAS_MOM) {
    b = 0.0f;
  }
}

template <bool HAS_MASK, bool HAS_WD, bool HAS_MOM>
__global__ __launch_bounds__(SALUN_BLOCK) void k_masked_sgd_vec(SgdArgs a) {
  const int64_t nvec = a.n >> 2;
  const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
  for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int64_t base = t * TILE_VEC + threadIdx.x;
    float4 pv[UNROLL], gv[UNROLL], bv[UNROLL];
    uint32_t mv[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t v = base + u * SALUN_BLOCK;
      if (v < nvec) {
        mv[u] = HAS_MASK ? ldm(a.m, v) : 0x01010101u;
        pv[u] = ld4(a.p, v);
        gv[u] = ld4_nt(a.g, v);
        if (HAS_MOM) bv[u] = ld4(a.buf, v);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int64_t v = base + u * SALUN_BLOCK;
      if (v < nvec) {
        const uint32_t mm = mv[u];
        sgd_elem<HAS_WD, HAS_MOM>(pv[u].x, gv[u].x, bv[u].x, (mm & 0x000000FFu) != 0, a);
        sgd_elem<HAS_WD, HAS_MOM>(pv[u].y, gv[u].y, bv[u].y, (mm & 0x0000FF00u) != 0, a);
        sgd_elem<HAS_WD, HAS_MOM>(pv[u].z, gv[u].z, bv[u].z, (mm & 0x00FF0000u) != 0, a);
        sgd_elem<HAS_WD, HAS_MOM>(pv[u].w, gv[u].w, bv[u].w, (mm & 0xFF000000u) != 0, a);
        st4(a.p, v, pv[u]);
        if (HAS_MOM) st4(a.buf, v, bv[u]);
      }
    }
  }
  if (blockIdx.x == 0) {  // n % 4 tail
    const int64_t i = (nvec << 2) + threadIdx.x;
    if (i < a.n) {
      float p = a.p[i], b = HAS_MOM ? a.buf[i] : 0.0f;
      sgd_elem<HAS_WD, HAS_MOM>(p, a.g[i], b, HAS_MASK ? a.m[i] != 0 : true, a);
      a.p[i] = p;
      if (HAS_MOM) a.buf[i] = b;
    }
  }
}

template <bool HAS_MASK, bool HAS_WD, bool HAS_MOM>
__global__ __launch_bounds__(SALUN_BLOCK) void k_masked_sgd_scalar(SgdArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < a.n;
       i += (int64_t)gridDim.x * SALUN_BLOCK) {
    float p = a.p[i], b = HAS_MOM ? a.buf[i] : 0.0f;
    sgd_elem<HAS_WD, HAS_MOM>(p, a.g[i], b, HAS_MASK ? a.m[i] != 0 : true, a);
    a.p[i] = p;
    if (HAS_MOM) a.buf[i] = b;
  }
}

template <bool HAS_MASK, bool HAS_WD, bool HAS_MOM>
int launch_sgd(const SgdArgs &a, bool vec, hipStream_t st) {
  if (vec) {
    const int grid = salun_grid_for(a.n, TILE_ELEMS);
    hipLaunchKernelGGL((k_masked_sgd_vec<HAS_MASK, HAS_WD, HAS_MOM>), dim3(grid), dim3(SALUN_BLOCK), 0, st, a);
  } else {
    const int grid = salun_grid_for(a.n, SALUN_BLOCK * 4);
    hipLaunchKernelGGL((k_masked_sgd_scalar<HAS_MASK, HAS_WD, HAS_MOM>), dim3(grid), dim3(SALUN_BLOCK), 0, st, a);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ------------------------------------------------------------------------ K5 ----
constexpr int REDUCE_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(SALUN_BLOCK) void k_sqnorm_partial(const float *__restrict__ g, int64_t n,
                                                                double *__restrict__ partial, int vec) {
  __shared__ double lds[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (vec) {
    const int64_t nvec = n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 gv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        gv[u] = (v < nvec) ? ld4(g, v) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        s0 = __builtin_fmaf(gv[u].x, gv[u].x, s0);
        s1 = __builtin_fmaf(gv[u].y, gv[u].y, s1);
        s2 = __builtin_fmaf(gv[u].z, gv[u].z, s2);
        s3 = __builtin_fmaf(gv[u].w, gv[u].w, s3);
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < n) s0 = __builtin_fmaf(g[i], g[i], s0);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * SALUN_BLOCK)
      s0 = __builtin_fmaf(g[i], g[i], s0);
  }
  const double tot = salun_block_sum(((double)s0 + (double)s1) + ((double)s2 + (double)s3), lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// One workgroup folds <= REDUCE_MAX_BLOCKS partials in a fixed order.
__global__ __launch_bounds__(SALUN_BLOCK) void k_sum_partials_f32(const double *__restrict__ partial, int count,
                                                                   float *__restrict__ out) {
  __shared__ double lds[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += SALUN_BLOCK) s += partial[i];
  const double tot = salun_block_sum(s, lds);
  if (threadIdx.x == 0) *out = (float)tot;
}

struct AdamArgs {
  float *p;
  const float *g;
  float *m1;
  float *v;
  const uint8_t *mask;
  const float *sqnorm;
  float max_norm, gscale;
  float b1, omb1, b2, omb2, eps, wd;
  float bc2_sqrt, neg_step_size;
  const float *coef;  // optional device pair {sqrt(1 - b2^t), -lr / (1 - b1^t)}: replaces the two host values above
  int64_t n;
};

template <bool HAS_WD>
__device__ __forceinline__ void adam_elem(float &p, float g, float &m1, float &v, float mf, float s,
                                          const AdamArgs &a) {
  float ge = (g * s) * mf;
  if (HAS_WD) ge = __builtin_fmaf(a.wd, p, ge);
  m1 = (a.b1 * m1) + (a.omb1 * ge);
  v = (a.b2 * v) + ((a.omb2 * ge) * ge);
  const float den = (sqrtf(v) / a.bc2_sqrt) + a.eps;
  p = p + (a.neg_step_size * (m1 / den));
}

// step-dependent scalars of Adam computed ON THE DEVICE from a device-resident step counter (a captured HIP graph
// replays the same kernel arguments every step: the host cannot pass t).  Same double arithmetic as the host path.
__global__ void k_adam_coefficients(long long *step, double lr, double b1, double b2, float *coef) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long t = *step + 1;
    *step = t;
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    coef[0] = (float)sqrt(bc2);
    coef[1] = (float)(-(lr / bc1));
  }
}

template <bool HAS_MASK, bool HAS_WD, bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_masked_adam(AdamArgs a) {
  const float s = a.sqnorm ? salun_clip_coef(*a.sqnorm, a.max_norm) : a.gscale;
  if (a.coef) { a.bc2_sqrt = a.coef[0]; a.neg_step_size = a.coef[1]; }
  if (VEC) {
    const int64_t nvec = a.n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 pv[UNROLL], gv[UNROLL], mv1[UNROLL], vv[UNROLL];
      uint32_t mk[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          mk[u] = HAS_MASK ? ldm(a.mask, v) : 0x01010101u;
          pv[u] = ld4(a.p, v);
          gv[u] = ld4_nt(a.g, v);
          mv1[u] = ld4(a.m1, v);
          vv[u] = ld4(a.v, v);
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          const uint32_t mm = mk[u];
          adam_elem<HAS_WD>(pv[u].x, gv[u].x, mv1[u].x, vv[u].x, (mm & 0x000000FFu) ? 1.f : 0.f, s, a);
          adam_elem<HAS_WD>(pv[u].y, gv[u].y, mv1[u].y, vv[u].y, (mm & 0x0000FF00u) ? 1.f : 0.f, s, a);
          adam_elem<HAS_WD>(pv[u].z, gv[u].z, mv1[u].z, vv[u].z, (mm & 0x00FF0000u) ? 1.f : 0.f, s, a);
          adam_elem<HAS_WD>(pv[u].w, gv[u].w, mv1[u].w, vv[u].w, (mm & 0xFF000000u) ? 1.f : 0.f, s, a);
          st4(a.p, v, pv[u]);
          st4(a.m1, v, mv1[u]);
          st4(a.v, v, vv[u]);
        }
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < a.n) {
        float p = a.p[i], m1 = a.m1[i], v = a.v[i];
        adam_elem<HAS_WD>(p, a.g[i], m1, v, HAS_MASK ? (a.mask[i] ? 1.f : 0.f) : 1.f, s, a);
        a.p[i] = p;
        a.m1[i] = m1;
        a.v[i] = v;
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < a.n;
         i += (int64_t)gridDim.x * SALUN_BLOCK) {
      float p = a.p[i], m1 = a.m1[i], v = a.v[i];
      adam_elem<HAS_WD>(p, a.g[i], m1, v, HAS_MASK ? (a.mask[i] ? 1.f : 0.f) : 1.f, s, a);
      a.p[i] = p;
      a.m1[i] = m1;
      a.v[i] = v;
    }
  }
}

template <bool HAS_MASK, bool HAS_WD>
int launch_adam(const AdamArgs &a, bool vec, hipStream_t st) {
  if (vec) {
    const int grid = salun_grid_for(a.n, TILE_ELEMS);
    hipLaunchKernelGGL((k_masked_adam<HAS_MASK, HAS_WD, true>), dim3(grid), dim3(SALUN_BLOCK), 0, st, a);
  } else {
    const int grid = salun_grid_for(a.n, SALUN_BLOCK * 4);
    hipLaunchKernelGGL((k_masked_adam<HAS_MASK, HAS_WD, false>), dim3(grid), dim3(SALUN_BLOCK), 0, st, a);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ------------------------------------------------------------------------ K7 ----
template <bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_fim_square_accumulate(float *__restrict__ F, float *__restrict__ tmp,
                                                                        float n_data, int64_t n) {
  if (VEC) {
    const int64_t nvec = n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 fv[UNROLL], tv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          fv[u] = ld4(F, v);
          tv[u] = ld4(tmp, v);
        }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          fv[u].x = fv[u].x + ((tv[u].x * tv[u].x) / n_data);
          fv[u].y = fv[u].y + ((tv[u].y * tv[u].y) / n_data);
          fv[u].z = fv[u].z + ((tv[u].z * tv[u].z) / n_data);
          fv[u].w = fv[u].w + ((tv[u].w * tv[u].w) / n_data);
          st4(F, v, fv[u]);
          st4(tmp, v, zero);
        }
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < n) {
        const float t = tmp[i];
        F[i] = F[i] + ((t * t) / n_data);
        tmp[i] = 0.f;
      }
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * SALUN_BLOCK) {
      const float t = tmp[i];
      F[i] = F[i] + ((t * t) / n_data);
      tmp[i] = 0.f;
    }
  }
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT int salun_saliency_accumulate(float *acc, const float *g, double scale, const float *sqnorm,
                                           double max_norm, int64_t n, salun_stream_t stream) {
  if (n < 0 || (n > 0 && (!acc || !g))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  AccumArgs a{acc, g, sqnorm, (float)scale, (float)max_norm, n};
  hipStream_t st = salun_hip_stream(stream);
  if (salun_aligned16(acc) && salun_aligned16(g)) {
    hipLaunchKernelGGL(k_saliency_accumulate<true>, dim3(salun_grid_for(n, TILE_ELEMS)), dim3(SALUN_BLOCK), 0, st, a);
  } else {
    hipLaunchKernelGGL(k_saliency_accumulate<false>, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0, st, a);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_masked_sgd_step(float *p, const float *g, float *buf, const uint8_t *m, double lr,
                                       double mu, double wd, int first_step, int64_t n,
                                       salun_stream_t stream) {
  if (n < 0 || (n > 0 && (!p || !g))) return SALUN_EINVAL;
  const bool has_mom = (mu != 0.0);
  if (n > 0 && has_mom && !buf) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  SgdArgs a{p, g, buf, m, (float)(-lr), (float)mu, (float)wd, first_step, n};
  const bool has_wd = (wd != 0.0), has_mask = (m != nullptr);
  const bool vec = salun_aligned16(p) && salun_aligned16(g) && (!has_mom || salun_aligned16(buf)) &&
                   (!has_mask || salun_aligned4(m));
  hipStream_t st = salun_hip_stream(stream);
#define SALUN_SGD_CASE(M, W, O) \
  if (has_mask == M && has_wd == W && has_mom == O) return launch_sgd<M, W, O>(a, vec, st);
  SALUN_SGD_CASE(true, true, true)
  SALUN_SGD_CASE(true, true, false)
  SALUN_SGD_CASE(true, false, true)
  SALUN_SGD_CASE(true, false, false)
  SALUN_SGD_CASE(false, true, true)
  SALUN_SGD_CASE(false, true, false)
  SALUN_SGD_CASE(false, false, true)
  SALUN_SGD_CASE(false, false, false)
#undef SALUN_SGD_CASE
  return SALUN_EINVAL;
}

SALUN_EXPORT size_t salun_reduce_workspace_bytes(int64_t n) {
  (void)n;
  return sizeof(double) * REDUCE_MAX_BLOCKS * 4;  // partial sums (+ slack for sqerr per-sample staging)
}

SALUN_EXPORT int salun_grad_sqnorm(const float *g, int64_t n, float *out, void *ws, size_t ws_bytes,
                                   salun_stream_t stream) {
  if (n < 0 || !out || !ws || (n > 0 && !g)) return SALUN_EINVAL;
  if (ws_bytes < sizeof(double) * REDUCE_MAX_BLOCKS) return SALUN_ENOSPC;
  hipStream_t st = salun_hip_stream(stream);
  double *partial = static_cast<double *>(ws);
  const int vec = salun_aligned16(g) ? 1 : 0;
  int grid = salun_grid_for(n, TILE_ELEMS);
  if (grid > REDUCE_MAX_BLOCKS) grid = REDUCE_MAX_BLOCKS;
  hipLaunchKernelGGL(k_sqnorm_partial, dim3(grid), dim3(SALUN_BLOCK), 0, st, g, n, partial, vec);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials_f32, dim3(1), dim3(SALUN_BLOCK), 0, st, partial, grid, out);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

static int masked_adam_impl(float *p, const float *g, float *m1, float *v, const uint8_t *mask, const float *sqnorm,
                            double max_norm, double gscale, double lr, double b1, double b2, double eps, double wd,
                            int step, const float *coef, int64_t n, salun_stream_t stream);

SALUN_EXPORT int salun_masked_adam_step(float *p, const float *g, float *m1, float *v, const uint8_t *mask,
                                        const float *sqnorm, double max_norm, double gscale, double lr,
                                        double b1, double b2, double eps, double wd, int step, int64_t n,
                                        salun_stream_t stream) {
  if (step < 1) return SALUN_EINVAL;
  return masked_adam_impl(p, g, m1, v, mask, sqnorm, max_norm, gscale, lr, b1, b2, eps, wd, step, nullptr, n, stream);
}

SALUN_EXPORT int salun_adam_coefficients(int64_t *step, double lr, double b1, double b2, float *coef,
                                         salun_stream_t stream) {
  if (!step || !coef) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_adam_coefficients, dim3(1), dim3(64), 0, salun_hip_stream(stream),
                     reinterpret_cast<long long *>(step), lr, b1, b2, coef);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_masked_adam_step_coef(float *p, const float *g, float *m1, float *v, const uint8_t *mask,
                                             const float *sqnorm, double max_norm, double gscale, const float *coef,
                                             double b1, double b2, double eps, double wd, int64_t n,
                                             salun_stream_t stream) {
  if (!coef) return SALUN_EINVAL;
  return masked_adam_impl(p, g, m1, v, mask, sqnorm, max_norm, gscale, 0.0, b1, b2, eps, wd, 1, coef, n, stream);
}

static int masked_adam_impl(float *p, const float *g, float *m1, float *v, const uint8_t *mask, const float *sqnorm,
                            double max_norm, double gscale, double lr, double b1, double b2, double eps, double wd,
                            int step, const float *coef, int64_t n, salun_stream_t stream) {
  if (n < 0 || step < 1 || (n > 0 && (!p || !g || !m1 || !v))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  // Python-scalar arithmetic of torch.optim.adam._single_tensor_adam, in double.
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  AdamArgs a;
  a.coef = coef;
  a.p = p; a.g = g; a.m1 = m1; a.v = v; a.mask = mask; a.sqnorm = sqnorm;
  a.max_norm = (float)max_norm; a.gscale = (float)gscale;
  a.b1 = (float)b1; a.omb1 = (float)(1.0 - b1); a.b2 = (float)b2; a.omb2 = (float)(1.0 - b2);
  a.eps = (float)eps; a.wd = (float)wd;
  a.bc2_sqrt = (float)sqrt(bc2);
  a.neg_step_size = (float)(-(lr / bc1));
  a.n = n;
  const bool has_mask = mask != nullptr, has_wd = wd != 0.0;
  const bool vec = salun_aligned16(p) && salun_aligned16(g) && salun_aligned16(m1) && salun_aligned16(v) &&
                   (!has_mask || salun_aligned4(mask));
  hipStream_t st = salun_hip_stream(stream);
  if (has_mask && has_wd) return launch_adam<true, true>(a, vec, st);
  if (has_mask && !has_wd) return launch_adam<true, false>(a, vec, st);
  if (!has_mask && has_wd) return launch_adam<false, true>(a, vec, st);
  return launch_adam<false, false>(a, vec, st);
}

SALUN_EXPORT int salun_fim_square_accumulate(float *F, float *tmp, double n_data, int64_t n,
                                             salun_stream_t stream) {
  if (n < 0 || n_data == 0.0 || (n > 0 && (!F || !tmp))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipStream_t st = salun_hip_stream(stream);
  if (salun_aligned16(F) && salun_aligned16(tmp)) {
    hipLaunchKernelGGL(k_fim_square_accumulate<true>, dim3(salun_grid_for(n, TILE_ELEMS)), dim3(SALUN_BLOCK), 0, st,
                       F, tmp, (float)n_data, n);
  } else {
    hipLaunchKernelGGL(k_fim_square_accumulate<false>, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                       st, F, tmp, (float)n_data, n);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}


// ---------------------------------------------------------------------------------------------------
// Shader-clock probe (measurement tooling: tools/clock_probe.py).  One wave reads the shader-cycle counter (s_memtime)
// and the 100 MHz constant counter (s_memrealtime) `spins` sleeps apart: out[2i] = shader cycles, out[2i+1] = 100 MHz
// ticks of sample i — their ratio x 100 MHz is the clock the chip actually ran at while whatever else was resident ran.
namespace {
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *__restrict__ out, int samples, int spins) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < samples; ++i) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < spins; ++k) __builtin_amdgcn_s_sleep(127);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[2 * i] = t1 - t0;
    out[2 * i + 1] = r1 - r0;
  }
}
}  // namespace

SALUN_EXPORT int salun_clock_probe(unsigned long long *out, int samples, int spins, salun_stream_t stream) {
  if (!out || samples < 1 || spins < 1) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, salun_hip_stream(stream), out, samples, spins);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
