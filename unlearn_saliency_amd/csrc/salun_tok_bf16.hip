// salun_tok_bf16.hip — K14: the token-wise element-wise layers of the SD transformer blocks on bf16 tokens:
// LayerNorm forward / backward and GEGLU forward / backward.
//
// Reference: SD/ldm/modules/attention.py:196-216 `BasicTransformerBlock` (three `nn.LayerNorm(dim)` ahead of attn1 /
// attn2 / ff) and :37-57 `GEGLU` / `FeedForward` (`x, gate = proj(x).chunk(2, dim=-1); x * F.gelu(gate)`).  Under bf16
// autocast a LayerNorm is cast-to-fp32 -> layer_norm -> (fp32 result, cast again by the next Linear) and its backward
// four more kernels; GEGLU is chunk -> gelu -> mul forward and five element-wise kernels backward.  Here each is ONE
// pass forward and one pass (+ a small deterministic column reduction for dgamma / dbeta) backward: bf16 in / out,
// fp32 arithmetic, HBM-bound.
//
// LayerNorm: one wave per token row (C = 320 / 640 / 1280 channels = 40 / 80 / 160 octets of 8 bf16); a lane owns the
// octets lane, lane+64, lane+128 (<= 3), so the row stays in registers between the statistics and the normalise step.
#include "salun_common.h"

namespace {

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bf2f((uint16_t)(w[j] & 0xffffu)); f[2 * j + 1] = bf2f((uint16_t)(w[j] >> 16)); }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int LN_MAXO = 4;        // octets per lane: C <= 2048
constexpr int LN_ROWS_PER_WG = 4; // one wave per row

// y = gamma * (x - mean) * rstd + beta ; stats[row] = (mean, rstd)
__global__ __launch_bounds__(256) void k_ln16_fwd(const uint16_t *__restrict__ x, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, uint16_t *__restrict__ y,
                                                  float2 *__restrict__ stats, int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * LN_ROWS_PER_WG + wave;
  if (row >= rows) return;
  const int no = C >> 3;
  float v[LN_MAXO][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXO; ++i) {
    const int o = lane + 64 * i;
    if (o < no) {
      unpack8(*reinterpret_cast<const uint4 *>(x + row * C + o * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXO; ++i)
    if (lane + 64 * i < no) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0 && stats) stats[row] = make_float2(mean, rstd);
#pragma unroll
  for (int i = 0; i < LN_MAXO; ++i) {
    const int o = lane + 64 * i;
    if (o < no) {
      float out[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) out[j] = (v[i][j] - mean) * rstd * gamma[o * 8 + j] + beta[o * 8 + j];  // slices of the flat arena: 4-byte aligned only
      *reinterpret_cast<uint4 *>(y + row * C + o * 8) = pack8(out);
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-workgroup partial sums of dy*xhat / dy for
// dgamma / dbeta: a workgroup walks `rows_per_wg` rows (4 at a time), each lane keeps the sums of its channels, the
// four waves are folded through LDS -> part[wg][C][2]
__global__ __launch_bounds__(256) void k_ln16_bwd(const uint16_t *__restrict__ dy, const uint16_t *__restrict__ x,
                                                  const float *__restrict__ gamma, const float2 *__restrict__ stats,
                                                  uint16_t *__restrict__ dx, float2 *__restrict__ part, int64_t rows, int C,
                                                  int rows_per_wg) {
  extern __shared__ float2 red[];  // [3 waves][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int no = C >> 3;
  float gg[LN_MAXO][8], sg[LN_MAXO][8], sb[LN_MAXO][8];
#pragma unroll
  for (int i = 0; i < LN_MAXO; ++i) {
    const int o = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) { gg[i][j] = (o < no) ? gamma[o * 8 + j] : 0.f; sg[i][j] = sb[i][j] = 0.f; }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  for (int64_t row = r0 + wave; row < r1; row += 4) {
    const float2 st = stats[row];
    float xh[LN_MAXO][8], g[LN_MAXO][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXO; ++i) {
      const int o = lane + 64 * i;
      if (o < no) {
        float xv[8], dv[8];
        unpack8(*reinterpret_cast<const uint4 *>(x + row * C + o * 8), xv);
        unpack8(*reinterpret_cast<const uint4 *>(dy + row * C + o * 8), dv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xh[i][j] = (xv[j] - st.x) * st.y;
          g[i][j] = dv[j] * gg[i][j];
          s1 += g[i][j];
          s2 += g[i][j] * xh[i][j];
          sg[i][j] += dv[j] * xh[i][j];
          sb[i][j] += dv[j];
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < LN_MAXO; ++i) {
      const int o = lane + 64 * i;
      if (o < no) {
        float out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = st.y * (g[i][j] - m1 - xh[i][j] * m2);
        *reinterpret_cast<uint4 *>(dx + row * C + o * 8) = pack8(out);
      }
    }
  }
  // fold the four waves' column sums in a fixed order
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < LN_MAXO; ++i) {
      const int o = lane + 64 * i;
      if (o < no)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(size_t)(wave - 1) * C + o * 8 + j] = make_float2(sg[i][j], sb[i][j]);
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < LN_MAXO; ++i) {
      const int o = lane + 64 * i;
      if (o < no)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = sg[i][j], b = sb[i][j];
          for (int w = 0; w < 3; ++w) { const float2 t = red[(size_t)w * C + o * 8 + j]; a += t.x; b += t.y; }
          part[(size_t)blockIdx.x * C + o * 8 + j] = make_float2(a, b);
        }
    }
  }
}

// dgamma[c] / dbeta[c] (= or +=) = sum over workgroup partials, 32 channels x 8 lanes per block, fixed order
__global__ __launch_bounds__(256) void k_ln16_params(const float2 *__restrict__ part, float *__restrict__ dgamma,
                                                     float *__restrict__ dbeta, int C, int nparts, int accumulate) {
  __shared__ double sa[8][33], sb[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int i = ry; i < nparts; i += 8) { const float2 v = part[(size_t)i * C + c]; a += (double)v.x; b += (double)v.y; }
  sa[ry][cx] = a; sb[ry][cx] = b;
  __syncthreads();
  if (ry == 0 && c < C) {
    for (int q = 1; q < 8; ++q) { a += sa[q][cx]; b += sb[q][cx]; }
    dgamma[c] = accumulate ? dgamma[c] + (float)a : (float)a;
    dbeta[c] = accumulate ? dbeta[c] + (float)b : (float)b;
  }
}

// ---- GEGLU on h[rows][2*F]: out[rows][F] = h[:, :F] * gelu(h[:, F:])   (erf form, as torch.nn.functional.gelu)
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v);
}
__global__ __launch_bounds__(256) void k_geglu16_fwd(const uint16_t *__restrict__ h, uint16_t *__restrict__ out, int64_t rows,
                                                     int F) {
  const int fo = F >> 3;
  const int64_t total = rows * fo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / fo;
    const int o = (int)(i - r * fo);
    float a[8], b[8], y[8];
    unpack8(*reinterpret_cast<const uint4 *>(h + r * 2 * F + o * 8), a);
    unpack8(*reinterpret_cast<const uint4 *>(h + r * 2 * F + F + o * 8), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = a[j] * gelu_f(b[j]);
    *reinterpret_cast<uint4 *>(out + r * F + o * 8) = pack8(y);
  }
}
// dh[:, :F] = dy * gelu(b) ; dh[:, F:] = dy * a * gelu'(b)
__global__ __launch_bounds__(256) void k_geglu16_bwd(const uint16_t *__restrict__ h, const uint16_t *__restrict__ dy,
                                                     uint16_t *__restrict__ dh, int64_t rows, int F) {
  const int fo = F >> 3;
  const int64_t total = rows * fo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / fo;
    const int o = (int)(i - r * fo);
    float a[8], b[8], d[8], da[8], db[8];
    unpack8(*reinterpret_cast<const uint4 *>(h + r * 2 * F + o * 8), a);
    unpack8(*reinterpret_cast<const uint4 *>(h + r * 2 * F + F + o * 8), b);
    unpack8(*reinterpret_cast<const uint4 *>(dy + r * F + o * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) { da[j] = d[j] * gelu_f(b[j]); db[j] = d[j] * a[j] * gelu_grad(b[j]); }
    *reinterpret_cast<uint4 *>(dh + r * 2 * F + o * 8) = pack8(da);
    *reinterpret_cast<uint4 *>(dh + r * 2 * F + F + o * 8) = pack8(db);
  }
}

int ln_parts(int64_t rows) {
  int64_t p = (rows + 63) / 64;  // >= 64 rows per workgroup
  if (p > 1024) p = 1024;
  return (int)(p < 1 ? 1 : p);
}
bool ln_ok(int64_t rows, int C) { return rows >= 1 && C >= 8 && C % 8 == 0 && C <= 8 * 64 * LN_MAXO; }

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT size_t salun_ln_bf16_workspace_bytes(int64_t rows, int C) {
  if (!ln_ok(rows, C)) return 0;
  return (size_t)ln_parts(rows) * C * 2 * sizeof(float);
}

SALUN_EXPORT int salun_ln_bf16_forward(const uint16_t *x, const float *gamma, const float *beta, uint16_t *y, float *stats,
                                       int64_t rows, int C, double eps, salun_stream_t stream) {
  if (!x || !gamma || !beta || !y || !ln_ok(rows, C)) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(y)) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_ln16_fwd, dim3((unsigned)((rows + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG)), dim3(256), 0,
                     salun_hip_stream(stream), x, gamma, beta, y, reinterpret_cast<float2 *>(stats), rows, C, (float)eps);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_ln_bf16_backward(const uint16_t *dy, const uint16_t *x, const float *gamma, const float *stats,
                                        uint16_t *dx, float *dgamma, float *dbeta, int64_t rows, int C, int accumulate,
                                        void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!dy || !x || !gamma || !stats || !dx || !dgamma || !dbeta || !ws || !ln_ok(rows, C)) return SALUN_EINVAL;
  if (ws_bytes < salun_ln_bf16_workspace_bytes(rows, C)) return SALUN_ENOSPC;
  if (!salun_aligned16(x) || !salun_aligned16(dy) || !salun_aligned16(dx)) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int parts = ln_parts(rows);
  const int rpw = (int)((rows + parts - 1) / parts);
  float2 *part = static_cast<float2 *>(ws);
  hipLaunchKernelGGL(k_ln16_bwd, dim3(parts), dim3(256), sizeof(float2) * 3 * (size_t)C, st, dy, x, gamma,
                     reinterpret_cast<const float2 *>(stats), dx, part, rows, C, rpw);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_ln16_params, dim3((C + 31) / 32), dim3(256), 0, st, part, dgamma, dbeta, C, parts, accumulate);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_geglu_bf16_forward(const uint16_t *h, uint16_t *out, int64_t rows, int F, salun_stream_t stream) {
  if (!h || !out || rows < 1 || F < 8 || F % 8 || !salun_aligned16(h) || !salun_aligned16(out)) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_geglu16_fwd, dim3(salun_grid_for(rows * (F / 8), 256)), dim3(256), 0, salun_hip_stream(stream), h, out,
                     rows, F);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_geglu_bf16_backward(const uint16_t *h, const uint16_t *dy, uint16_t *dh, int64_t rows, int F,
                                           salun_stream_t stream) {
  if (!h || !dy || !dh || rows < 1 || F < 8 || F % 8 || !salun_aligned16(h) || !salun_aligned16(dy) || !salun_aligned16(dh))
    return SALUN_EINVAL;
  hipLaunchKernelGGL(k_geglu16_bwd, dim3(salun_grid_for(rows * (F / 8), 256)), dim3(256), 0, salun_hip_stream(stream), h, dy, dh,
                     rows, F);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
