// salun_prox.hip — the "next" rows that reuse the flat-vector machinery:
//   K9   proximal / soft-threshold step of RL_proximal (Classification/unlearn/RL_pro.py:52-60):
//          d = p - p0 ; tau = ratio-th smallest |d| (radix select of salun_topk.hip on d) ;
//          p <- d >  tau ? p - tau : d < -tau ? p + tau : p0
//   K10  EWC / Selective-Amnesia penalty of DDPM train_forget (DDPM/runners/diffusion.py:343-350):
//          loss += lambda * sum F (p - p*)^2 ;  g += (lambda F) * (2 (p - p*))
// All three are streaming kernels in the shape of salun_update.hip: float4 tiles of 4 x 256 per workgroup step,
// grid-stride, scalar tail by workgroup 0; -ffp-contract=off (every rounding is the one written).
#include "salun_common.h"

namespace {

constexpr int UNROLL = 4;
constexpr int TILE_VEC = UNROLL * SALUN_BLOCK;  // float4 per tile
constexpr int TILE_ELEMS = TILE_VEC * 4;        // floats per tile
constexpr int EWC_MAX_BLOCKS = 2048;

__device__ __forceinline__ float4 ld4(const float *p, int64_t v) { return reinterpret_cast<const float4 *>(p)[v]; }
__device__ __forceinline__ void st4(float *p, int64_t v, float4 x) { reinterpret_cast<float4 *>(p)[v] = x; }

// A NaN threshold is the select's failure signal (salun_mask_topk_thresholds exports NaN when the full scan's grid
// barrier timed out, or when the ranked element itself is NaN): the weights are poisoned with it so that the next
// loss is NaN — never a silent "every weight reset to p0", which is what the comparisons below would give.
__device__ __forceinline__ float soft(float p, float p0, float tau) {
  const float d = p - p0;
  if (tau != tau) return tau;
  return d > tau ? p - tau : (d < -tau ? p + tau : p0);
}

// out = p - p0
template <bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_param_diff(const float *__restrict__ p, const float *__restrict__ p0,
                                                            float *__restrict__ out, int64_t n) {
  if (VEC) {
    const int64_t nvec = n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 a[UNROLL], b[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) { a[u] = ld4(p, v); b[u] = ld4(p0, v); }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) st4(out, v, make_float4(a[u].x - b[u].x, a[u].y - b[u].y, a[u].z - b[u].z, a[u].w - b[u].w));
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < n) out[i] = p[i] - p0[i];
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
      out[i] = p[i] - p0[i];
  }
}

// p <- soft-threshold towards p0 with the device-resident threshold *tau
template <bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_soft_threshold(float *__restrict__ p, const float *__restrict__ p0,
                                                                const float *__restrict__ tau_ptr, int64_t n) {
  const float tau = *tau_ptr;
  if (VEC) {
    const int64_t nvec = n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 a[UNROLL], b[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) { a[u] = ld4(p, v); b[u] = ld4(p0, v); }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec)
          st4(p, v, make_float4(soft(a[u].x, b[u].x, tau), soft(a[u].y, b[u].y, tau), soft(a[u].z, b[u].z, tau),
                                soft(a[u].w, b[u].w, tau)));
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < n) p[i] = soft(p[i], p0[i], tau);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
      p[i] = soft(p[i], p0[i], tau);
  }
}

// g += (lam*F) * (2*(p - p*)) ;  partial[block] = sum F*(p-p*)^2 (fp64)
template <bool VEC>
__global__ __launch_bounds__(SALUN_BLOCK) void k_ewc(const float *__restrict__ p, const float *__restrict__ pstar,
                                                     const float *__restrict__ F, float *__restrict__ g, float lam,
                                                     int64_t n, double *__restrict__ partial) {
  __shared__ double lds[4];
  double acc = 0.0;
  auto elem = [&](float pv, float sv, float fv, float gv, double &a) {
    const float d = pv - sv;
    a += (double)(fv * (d * d));
    return gv + ((lam * fv) * (2.0f * d));
  };
  if (VEC) {
    const int64_t nvec = n >> 2;
    const int64_t ntile = (nvec + TILE_VEC - 1) / TILE_VEC;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
      const int64_t base = t * TILE_VEC + threadIdx.x;
      float4 a[UNROLL], b[UNROLL], f[UNROLL], gg[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) { a[u] = ld4(p, v); b[u] = ld4(pstar, v); f[u] = ld4(F, v); gg[u] = ld4(g, v); }
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t v = base + u * SALUN_BLOCK;
        if (v < nvec) {
          float4 o;
          o.x = elem(a[u].x, b[u].x, f[u].x, gg[u].x, acc);
          o.y = elem(a[u].y, b[u].y, f[u].y, gg[u].y, acc);
          o.z = elem(a[u].z, b[u].z, f[u].z, gg[u].z, acc);
          o.w = elem(a[u].w, b[u].w, f[u].w, gg[u].w, acc);
          st4(g, v, o);
        }
      }
    }
    if (blockIdx.x == 0) {
      const int64_t i = (nvec << 2) + threadIdx.x;
      if (i < n) g[i] = elem(p[i], pstar[i], F[i], g[i], acc);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * SALUN_BLOCK)
      g[i] = elem(p[i], pstar[i], F[i], g[i], acc);
  }
  const double t = salun_block_sum(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// loss_out[0] = (float)(lam * sum partial) ; loss_out[1] = (float)sum partial   (fixed order)
__global__ __launch_bounds__(SALUN_BLOCK) void k_ewc_final(const double *__restrict__ partial, int nblocks, float lam,
                                                           float *__restrict__ loss_out) {
  __shared__ double lds[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += SALUN_BLOCK) s += partial[i];
  const double t = salun_block_sum(s, lds);
  if (threadIdx.x == 0) {
    loss_out[0] = (float)((double)lam * t);
    loss_out[1] = (float)t;
  }
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT int salun_param_diff(const float *p, const float *p0, float *out, int64_t n, salun_stream_t stream) {
  if (n < 0 || (n > 0 && (!p || !p0 || !out))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipStream_t st = salun_hip_stream(stream);
  if (salun_aligned16(p) && salun_aligned16(p0) && salun_aligned16(out))
    hipLaunchKernelGGL(k_param_diff<true>, dim3(salun_grid_for(n, TILE_ELEMS)), dim3(SALUN_BLOCK), 0, st, p, p0, out, n);
  else
    hipLaunchKernelGGL(k_param_diff<false>, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0, st, p, p0,
                       out, n);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_soft_threshold_step(float *p, const float *p0, const float *tau, int64_t n,
                                           salun_stream_t stream) {
  if (n < 0 || !tau || (n > 0 && (!p || !p0))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  hipStream_t st = salun_hip_stream(stream);
  if (salun_aligned16(p) && salun_aligned16(p0))
    hipLaunchKernelGGL(k_soft_threshold<true>, dim3(salun_grid_for(n, TILE_ELEMS)), dim3(SALUN_BLOCK), 0, st, p, p0,
                       tau, n);
  else
    hipLaunchKernelGGL(k_soft_threshold<false>, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0, st, p,
                       p0, tau, n);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT size_t salun_ewc_workspace_bytes(int64_t n) {
  (void)n;
  return sizeof(double) * EWC_MAX_BLOCKS;
}

SALUN_EXPORT int salun_ewc_penalty_grad(const float *p, const float *p_star, const float *F, float *g, double lambda,
                                        float *loss_out, int64_t n, void *ws, size_t ws_bytes,
                                        salun_stream_t stream) {
  if (n < 0 || !loss_out || !ws || (n > 0 && (!p || !p_star || !F || !g))) return SALUN_EINVAL;
  if (ws_bytes < sizeof(double) * EWC_MAX_BLOCKS) return SALUN_ENOSPC;
  hipStream_t st = salun_hip_stream(stream);
  double *partial = static_cast<double *>(ws);
  const bool vec = salun_aligned16(p) && salun_aligned16(p_star) && salun_aligned16(F) && salun_aligned16(g);
  int grid = n > 0 ? salun_grid_for(n, vec ? TILE_ELEMS : SALUN_BLOCK * 4) : 1;
  if (grid > EWC_MAX_BLOCKS) grid = EWC_MAX_BLOCKS;
  if (vec)
    hipLaunchKernelGGL(k_ewc<true>, dim3(grid), dim3(SALUN_BLOCK), 0, st, p, p_star, F, g, (float)lambda, n, partial);
  else
    hipLaunchKernelGGL(k_ewc<false>, dim3(grid), dim3(SALUN_BLOCK), 0, st, p, p_star, F, g, (float)lambda, n, partial);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ewc_final, dim3(1), dim3(SALUN_BLOCK), 0, st, partial, grid, (float)lambda, loss_out);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
