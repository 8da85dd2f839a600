// salun_norm_bf16.hip — K12: GroupNorm (+ SiLU) forward / backward on bf16 NHWC activations, fp32 statistics.
//
// The normalisation layers of the Stable-Diffusion U-Net in its bf16 configuration (reference:
// SD/ldm/modules/diffusionmodules/util.py:215-217 GroupNorm32 evaluates in fp32 and casts back; used by
// openaimodel.py:192-196,214-221 ResBlock, :716-718 output head, attention.py:228 SpatialTransformer.norm).  Under
// autocast that is x.float() -> group_norm -> .type(bf16) -> silu: five passes over the activation, three of them fp32.
// Here: one read for the statistics, one read + one bf16 write for normalise(+SiLU); backward one pass for the
// per-channel sums and one pass for dX.  HBM-bound by construction; every reduction runs in a fixed order.
//
// Layout: x[N][HW][C] bf16, C % 8 == 0, group g owns channels [g*cpg, (g+1)*cpg).  A thread owns 8 consecutive
// channels (one 16-byte load per pixel) and strides over pixels, so loads are full rows and per-channel sums stay in
// registers; group sums are formed afterwards from the per-channel partials (a group's channels are only 20..160 bytes
// of a pixel's row — reading by group would waste most of every cache line).
#include "salun_common.h"

namespace {

__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }  // v_cvt_pk_bf16_f32: RNE
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((uint16_t)(w[j] & 0xffffu)); f[2 * j + 1] = bf2f((uint16_t)(w[j] >> 16)); }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2(f[2 * j], f[2 * j + 1]);
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Workspace layout (floats): part[N][chunks][C][2] | mr[N][G][2] (mean, rstd) lives in its own output | ab[N][C][2]
constexpr int GN_CHUNKS_MAX = 64;

// ---- per-channel partial sums over one pixel chunk: block = 32 channel octets x 8 pixel lanes
//   MODE 0: {sum x, sum x^2}        MODE 1 (backward): {sum dz, sum dz*xhat}
template <int MODE, bool SILU>
__global__ __launch_bounds__(256) void k_gn16_partial(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                      const float *__restrict__ ab /*[N][C][2]*/,
                                                      const float *__restrict__ mr /*[N][G][2]*/, float2 *__restrict__ part,
                                                      int HW, int C, int cpg, int rows_per_chunk) {
  __shared__ float2 s[8][32][9];
  const int gx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + gx) * 8;
  const int n = blockIdx.z, chunk = blockIdx.y;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = min(HW, r0 + rows_per_chunk);
  float a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a0[j] = a1[j] = 0.f;
  if (c < C) {
    float ca[8], cb[8], cm[8], cr[8];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ca[j] = ab[((size_t)n * C + c + j) * 2];
        cb[j] = ab[((size_t)n * C + c + j) * 2 + 1];
        const int g = (c + j) / cpg;
        cm[j] = mr[((size_t)n * (C / cpg) + g) * 2];
        cr[j] = mr[((size_t)n * (C / cpg) + g) * 2 + 1];
      }
    }
    const uint16_t *xp = x + ((size_t)n * HW) * C + c;
    const uint16_t *dp = MODE == 1 ? dy + ((size_t)n * HW) * C + c : nullptr;
    for (int r = r0 + ry; r < r1; r += 8) {
      float xv[8];
      unpack8(*reinterpret_cast<const uint4 *>(xp + (size_t)r * C), xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { a0[j] += xv[j]; a1[j] += xv[j] * xv[j]; }
      } else {
        float dv[8];
        unpack8(*reinterpret_cast<const uint4 *>(dp + (size_t)r * C), dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float dz = dv[j];
          if (SILU) {
            const float z = ca[j] * xv[j] + cb[j];
            const float sg = 1.f / (1.f + __expf(-z));
            dz *= sg * (1.f + z * (1.f - sg));
          }
          a0[j] += dz;
          a1[j] += dz * ((xv[j] - cm[j]) * cr[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s[ry][gx][j] = make_float2(a0[j], a1[j]);
  __syncthreads();
  if (ry == 0 && c < C) {
    float2 *dst = part + ((size_t)(n * gridDim.y + chunk) * C + c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float2 t = s[0][gx][j];
#pragma unroll
      for (int q = 1; q < 8; ++q) { t.x += s[q][gx][j].x; t.y += s[q][gx][j].y; }
      dst[j] = t;
    }
  }
}

// ---- forward finalize: one wave per (n, g): mean / rstd in fp64, then a = gamma*rstd, b = beta - mean*a per channel
__global__ __launch_bounds__(64) void k_gn16_finalize(const float2 *__restrict__ part, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ mr,
                                                      float *__restrict__ ab, int C, int cpg, int chunks, int HW, float eps) {
  const int g = blockIdx.x, n = blockIdx.y, G = gridDim.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  const int items = chunks * cpg;
  for (int i = lane; i < items; i += 64) {
    const int ch = i / cpg, cc = i - ch * cpg;
    const float2 v = part[(size_t)(n * chunks + ch) * C + g * cpg + cc];
    s0 += (double)v.x; s1 += (double)v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
  const double m = (double)HW * cpg;
  const double mean = s0 / m;
  double var = s1 / m - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (lane == 0) { mr[((size_t)n * G + g) * 2] = (float)mean; mr[((size_t)n * G + g) * 2 + 1] = rstd; }
  for (int cc = lane; cc < cpg; cc += 64) {
    const int c = g * cpg + cc;
    const float a = gamma[c] * rstd;
    ab[((size_t)n * C + c) * 2] = a;
    ab[((size_t)n * C + c) * 2 + 1] = beta[c] - (float)mean * a;
  }
}

// ---- forward apply: y = [silu](a*x + b)
template <bool SILU>
__global__ __launch_bounds__(256) void k_gn16_apply(const uint16_t *__restrict__ x, const float *__restrict__ ab,
                                                    uint16_t *__restrict__ y, int64_t octets, int hw_c8, int c8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < octets; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / hw_c8);
    const int c = (int)(i % c8) * 8;
    float v[8];
    unpack8(*reinterpret_cast<const uint4 *>(x + i * 8), v);
    const float4 *p = reinterpret_cast<const float4 *>(ab + ((size_t)n * c8 * 8 + c) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = p[q];  // a0 b0 a1 b1
      float z0 = t.x * v[2 * q] + t.y, z1 = t.z * v[2 * q + 1] + t.w;
      if (SILU) { z0 = z0 / (1.f + __expf(-z0)); z1 = z1 / (1.f + __expf(-z1)); }
      v[2 * q] = z0; v[2 * q + 1] = z1;
    }
    *reinterpret_cast<uint4 *>(y + i * 8) = pack8(v);
  }
}

// ---- backward finalize per (n, g): s1 = sum_c gamma*A, s2 = sum_c gamma*B over chunks and the group's channels
__global__ __launch_bounds__(64) void k_gn16_bwd_group(const float2 *__restrict__ part, const float *__restrict__ gamma,
                                                       float *__restrict__ gs /*[N][G][2]*/, int C, int cpg, int chunks) {
  const int g = blockIdx.x, n = blockIdx.y, G = gridDim.x, lane = threadIdx.x;
  double s0 = 0.0, s1 = 0.0;
  const int items = chunks * cpg;
  for (int i = lane; i < items; i += 64) {
    const int ch = i / cpg, cc = i - ch * cpg;
    const float2 v = part[(size_t)(n * chunks + ch) * C + g * cpg + cc];
    const double gm = (double)gamma[g * cpg + cc];
    s0 += gm * (double)v.x; s1 += gm * (double)v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
  if (lane == 0) { gs[((size_t)n * G + g) * 2] = (float)s0; gs[((size_t)n * G + g) * 2 + 1] = (float)s1; }
}

// ---- parameter gradients: dbeta[c] = sum_{n,chunk} A, dgamma[c] = sum_{n,chunk} B; block = 32 channels x 8 lanes over
// the (n, chunk) list, lane partials folded in a fixed order
__global__ __launch_bounds__(256) void k_gn16_bwd_params(const float2 *__restrict__ part, float *__restrict__ dgamma,
                                                         float *__restrict__ dbeta, int C, int nchunks_total, int accumulate) {
  __shared__ double sa[8][33], sb[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = ry; i < nchunks_total; i += 8) { const float2 v = part[(size_t)i * C + c]; a += (double)v.x; b += (double)v.y; }
  sa[ry][cx] = a; sb[ry][cx] = b;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int q = 1; q < 8; ++q) { a += sa[q][cx]; b += sb[q][cx]; }
    dbeta[c] = accumulate ? dbeta[c] + (float)a : (float)a;
    dgamma[c] = accumulate ? dgamma[c] + (float)b : (float)b;
  }
}

// ---- backward apply: dx = rstd * (dz*gamma - (s1 + xhat*s2)/m)
template <bool SILU>
__global__ __launch_bounds__(256) void k_gn16_bwd_apply(const uint16_t *__restrict__ x, const uint16_t *__restrict__ dy,
                                                        const float *__restrict__ ab, const float *__restrict__ mr,
                                                        const float *__restrict__ gs, const float *__restrict__ gamma,
                                                        uint16_t *__restrict__ dx, int64_t octets, int hw_c8, int c8, int cpg,
                                                        float inv_m) {
  const int G = c8 * 8 / cpg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < octets; i += (int64_t)gridDim.x * 256) {
    const int n = (int)(i / hw_c8);
    const int c = (int)(i % c8) * 8;
    float xv[8], dv[8], o[8];
    unpack8(*reinterpret_cast<const uint4 *>(x + i * 8), xv);
    unpack8(*reinterpret_cast<const uint4 *>(dy + i * 8), dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cj = c + j, g = cj / cpg;
      const float mean = mr[((size_t)n * G + g) * 2], rstd = mr[((size_t)n * G + g) * 2 + 1];
      const float s1 = gs[((size_t)n * G + g) * 2], s2 = gs[((size_t)n * G + g) * 2 + 1];
      float dz = dv[j];
      if (SILU) {
        const float z = ab[((size_t)n * c8 * 8 + cj) * 2] * xv[j] + ab[((size_t)n * c8 * 8 + cj) * 2 + 1];
        const float sg = 1.f / (1.f + __expf(-z));
        dz *= sg * (1.f + z * (1.f - sg));
      }
      const float xh = (xv[j] - mean) * rstd;
      o[j] = rstd * (dz * gamma[cj] - (s1 + xh * s2) * inv_m);
    }
    *reinterpret_cast<uint4 *>(dx + i * 8) = pack8(o);
  }
}

int gn_chunks(int HW) {
  int c = (HW + 31) / 32;  // at least 32 pixels per chunk (4 per pixel lane)
  if (c > GN_CHUNKS_MAX) c = GN_CHUNKS_MAX;
  return c < 1 ? 1 : c;
}

bool gn_ok(int N, int C, int HW, int G) { return N >= 1 && HW >= 1 && G >= 1 && C % 8 == 0 && C % G == 0 && C >= 8; }

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT size_t salun_gn_bf16_workspace_bytes(int N, int C, int HW, int G) {
  if (!gn_ok(N, C, HW, G)) return 0;
  return ((size_t)N * gn_chunks(HW) * C * 2 + (size_t)N * G * 2) * sizeof(float);
}

// y = [silu](GroupNorm_G(x)); outputs mr[N][G][2] = (mean, rstd) and ab[N][C][2] = (gamma*rstd, beta - mean*gamma*rstd),
// both consumed by the backward.
SALUN_EXPORT int salun_gn_bf16_forward(const uint16_t *x, const float *gamma, const float *beta, uint16_t *y, float *mr,
                                       float *ab, int N, int C, int HW, int G, double eps, int silu, void *ws,
                                       size_t ws_bytes, salun_stream_t stream) {
  if (!x || !gamma || !beta || !y || !mr || !ab || !ws || !gn_ok(N, C, HW, G)) return SALUN_EINVAL;
  if (ws_bytes < salun_gn_bf16_workspace_bytes(N, C, HW, G)) return SALUN_ENOSPC;
  if (!salun_aligned16(x) || !salun_aligned16(y) || !salun_aligned16(ab)) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int chunks = gn_chunks(HW), rpc = (HW + chunks - 1) / chunks, cpg = C / G;
  float2 *part = static_cast<float2 *>(ws);
  hipLaunchKernelGGL((k_gn16_partial<0, false>), dim3((C / 8 + 31) / 32, chunks, N), dim3(256), 0, st, x, nullptr, nullptr,
                     nullptr, part, HW, C, cpg, rpc);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn16_finalize, dim3(G, N), dim3(64), 0, st, part, gamma, beta, mr, ab, C, cpg, chunks, HW, (float)eps);
  SALUN_LAUNCH_CHECK();
  const int64_t octets = (int64_t)N * HW * (C / 8);
  const int grid = salun_grid_for(octets, 256);
  if (silu) hipLaunchKernelGGL(k_gn16_apply<true>, dim3(grid), dim3(256), 0, st, x, ab, y, octets, HW * (C / 8), C / 8);
  else hipLaunchKernelGGL(k_gn16_apply<false>, dim3(grid), dim3(256), 0, st, x, ab, y, octets, HW * (C / 8), C / 8);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// dx (bf16), dgamma / dbeta (fp32, = or +=) from dy, x and the forward's mr / ab.
SALUN_EXPORT int salun_gn_bf16_backward(const uint16_t *dy, const uint16_t *x, const float *gamma, const float *mr,
                                        const float *ab, uint16_t *dx, float *dgamma, float *dbeta, int N, int C, int HW,
                                        int G, int silu, int accumulate, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!dy || !x || !gamma || !mr || !ab || !dx || !dgamma || !dbeta || !ws || !gn_ok(N, C, HW, G)) return SALUN_EINVAL;
  if (ws_bytes < salun_gn_bf16_workspace_bytes(N, C, HW, G)) return SALUN_ENOSPC;
  if (!salun_aligned16(x) || !salun_aligned16(dy) || !salun_aligned16(dx)) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int chunks = gn_chunks(HW), rpc = (HW + chunks - 1) / chunks, cpg = C / G;
  float2 *part = static_cast<float2 *>(ws);
  float *gs = reinterpret_cast<float *>(part + (size_t)N * chunks * C);
  const dim3 pg((C / 8 + 31) / 32, chunks, N);
  if (silu) hipLaunchKernelGGL((k_gn16_partial<1, true>), pg, dim3(256), 0, st, x, dy, ab, mr, part, HW, C, cpg, rpc);
  else hipLaunchKernelGGL((k_gn16_partial<1, false>), pg, dim3(256), 0, st, x, dy, ab, mr, part, HW, C, cpg, rpc);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn16_bwd_group, dim3(G, N), dim3(64), 0, st, part, gamma, gs, C, cpg, chunks);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn16_bwd_params, dim3((C + 31) / 32), dim3(256), 0, st, part, dgamma, dbeta, C, N * chunks, accumulate);
  SALUN_LAUNCH_CHECK();
  const int64_t octets = (int64_t)N * HW * (C / 8);
  const int grid = salun_grid_for(octets, 256);
  const float inv_m = 1.0f / ((float)HW * (float)cpg);
  if (silu)
    hipLaunchKernelGGL(k_gn16_bwd_apply<true>, dim3(grid), dim3(256), 0, st, x, dy, ab, mr, gs, gamma, dx, octets, HW * (C / 8),
                       C / 8, cpg, inv_m);
  else
    hipLaunchKernelGGL(k_gn16_bwd_apply<false>, dim3(grid), dim3(256), 0, st, x, dy, ab, mr, gs, gamma, dx, octets, HW * (C / 8),
                       C / 8, cpg, inv_m);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
