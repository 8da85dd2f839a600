// salun_norm.hip — fused BatchNorm2d (+ residual add) (+ ReLU), forward and backward, NCHW fp32.
//
// ResNet's BN / ReLU / residual-add chain is pure HBM traffic: as separate library launches it makes ~19 passes over
// every activation tensor per training step (profiles/r01_bench_kernel_stats.csv: 2.65 of 14.1 ms).  Fused:
//   forward   stats (1 read)                      -> mean, invstd (+ running stats update)
//             apply (1-2 reads, 1 write)          y = relu(gamma*(x-mean)*invstd + beta + residual)
//   backward  reduce (3 reads)                    dbeta = sum dz, dgamma = sum dz*xhat,  dz = dy * [y > 0]
//             apply (3 reads, 1-2 writes)         dx = gamma*invstd*(dz - (dbeta + xhat*dgamma)/M)   (train)
//                                                 dx = gamma*invstd*dz                               (eval)
//                                                 dres = dz
// Reductions: per (channel, slice-of-batch) workgroup partials in fp64, folded in a fixed order (deterministic).
// Semantics follow torch.nn.BatchNorm2d (biased variance for normalisation, unbiased for running_var, momentum).
#include "salun_common.h"

namespace {

constexpr int BN_MAX_SPLIT = 64;

// (Round 6, measured and NOT adopted: s_setprio 1 / 3 in these kernels.  Beside the backward-weight kernel they got faster
// — BatchNorm backward 128 -> 96 us — but the training step got 2.7 % slower: the side stream's backward-weight chain is
// the critical path of backward and loses what these kernels win.  Same sign as priorities on the matrix kernels.)

// Walk of the float4 items e = tid, tid + 256, ... of a (channel, batch slice): image n = n_lo + e / hw4, piece
// i = e % hw4 — advanced WITHOUT a division per item.  Round 6: these kernels run beside the backward-weight kernel of
// the side stream, whose MFMA stream leaves other waves of its SIMD few VALU issue slots (BatchNorm backward took
// 118 - 147 us there against 79 us alone); the integer division was the longest dependent VALU chain of every iteration.
struct BnWalk {
  int n, i, dn, di, hw4;
  __device__ __forceinline__ BnWalk(int n_lo, int hw4_) : hw4(hw4_) {
    const int t = threadIdx.x;
    n = n_lo + t / hw4_;
    i = t - (t / hw4_) * hw4_;
    dn = 256 / hw4_;
    di = 256 - dn * hw4_;
  }
  __device__ __forceinline__ void next() {
    n += dn;
    i += di;
    if (i >= hw4) { i -= hw4; ++n; }
  }
};

// partial[(c*nsplit + s)*2 + {0,1}]: sums over images n in slice s of channel c
__global__ __launch_bounds__(256) void k_bn_stats_partial(const float *__restrict__ x, int N, int C, int HW, int nsplit,
                                                          double *__restrict__ partial) {
  __shared__ double lds[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const int n_lo = (int)((int64_t)N * s / nsplit), n_hi = (int)((int64_t)N * (s + 1) / nsplit);
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  const int hw4 = HW >> 2;
  const int work = (n_hi - n_lo) * hw4;  // float4 items of this (channel, slice): all lanes busy even for 4x4 maps
  int it = 0;
  BnWalk wk(n_lo, hw4);
#pragma unroll 2
  for (int e = threadIdx.x; e < work; e += 256, ++it, wk.next()) {
    const int n = wk.n, i = wk.i;
    const float4 v = reinterpret_cast<const float4 *>(x + ((size_t)n * C + c) * HW)[i];
    s1 += (v.x + v.y) + (v.z + v.w);
    s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    if ((it & 7) == 7) { d1 += s1; d2 += s2; s1 = 0.f; s2 = 0.f; }  // bound fp32 accumulation length
  }
  d1 += s1; d2 += s2;
  const double t1 = salun_block_sum(d1, lds);
  const double t2 = salun_block_sum(d2, lds);
  if (threadIdx.x == 0) {
    partial[((size_t)c * nsplit + s) * 2 + 0] = t1;
    partial[((size_t)c * nsplit + s) * 2 + 1] = t2;
  }
}

// Per-channel statistics from the fp64 partials, computed redundantly by every workgroup of the channel (nsplit <= 64
// values: one wave, fixed order) — no separate finalise launch.  Slice 0 publishes them and updates the running stats.
__device__ __forceinline__ void bn_channel_stats(const double *__restrict__ partial, int c, int nsplit, double M,
                                                 float eps, float *s_out /*LDS: mean, invstd*/, bool publish,
                                                 float momentum, float *__restrict__ mean, float *__restrict__ invstd,
                                                 float *__restrict__ running_mean, float *__restrict__ running_var) {
  if (threadIdx.x < 64) {
    double s1 = 0.0, s2 = 0.0;
    if ((int)threadIdx.x < nsplit) {
      s1 = partial[((size_t)c * nsplit + threadIdx.x) * 2 + 0];
      s2 = partial[((size_t)c * nsplit + threadIdx.x) * 2 + 1];
    }
    s1 = salun_wave_sum(s1);
    s2 = salun_wave_sum(s2);
    if (threadIdx.x == 0) {
      const double mu = s1 / M;
      double var = s2 / M - mu * mu;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      s_out[0] = (float)mu;
      s_out[1] = is;
      if (publish) {
        mean[c] = (float)mu;
        invstd[c] = is;
        if (running_mean) {
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
          const double unbiased = (M > 1.0) ? var * M / (M - 1.0) : var;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
      }
    }
  }
  __syncthreads();
}

// y = [relu](gamma*(x-mean)*invstd + beta [+ res]) for channel blockIdx.x, batch slice blockIdx.y.
// TRAIN: mean/invstd come from the partial sums (see above); else from the running statistics.
template <bool RELU, bool RES, bool TRAIN>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, const float *__restrict__ res,
                                                  float *__restrict__ y, const double *__restrict__ partial,
                                                  float *__restrict__ mean, float *__restrict__ invstd,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float *__restrict__ running_mean, float *__restrict__ running_var,
                                                  long long *__restrict__ num_batches_tracked, int N, int C, int HW,
                                                  int nsplit, float eps, float momentum) {
  __shared__ float st[2];
  const int c = blockIdx.x, s = blockIdx.y;
  if (TRAIN) {
    if (num_batches_tracked && c == 0 && s == 0 && threadIdx.x == 0) *num_batches_tracked += 1;  // nn.BatchNorm2d's counter
    bn_channel_stats(partial, c, nsplit, (double)N * HW, eps, st, s == 0, momentum, mean, invstd, running_mean,
                     running_var);
  } else {
    if (threadIdx.x == 0) {
      st[0] = running_mean[c];
      st[1] = 1.0f / sqrtf(running_var[c] + eps);
      if (s == 0) { mean[c] = st[0]; invstd[c] = st[1]; }
    }
    __syncthreads();
  }
  const float a = st[1] * gamma[c];
  const float b = beta[c] - st[0] * a;
  const int n_lo = (int)((int64_t)N * s / nsplit), n_hi = (int)((int64_t)N * (s + 1) / nsplit);
  const int hw4 = HW >> 2;
  const int work = (n_hi - n_lo) * hw4;
  BnWalk wk(n_lo, hw4);
#pragma unroll 2
  for (int e = threadIdx.x; e < work; e += 256, wk.next()) {
    const int n = wk.n, i = wk.i;
    const size_t off = ((size_t)n * C + c) * HW;
    const float4 v = reinterpret_cast<const float4 *>(x + off)[i];
    float4 o;
    o.x = v.x * a + b; o.y = v.y * a + b; o.z = v.z * a + b; o.w = v.w * a + b;
    if (RES) {
      const float4 r = reinterpret_cast<const float4 *>(res + off)[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (RELU) {
      o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
      o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
    }
    reinterpret_cast<float4 *>(y + off)[i] = o;
  }
}

// partial[(c*nsplit + s)*2 + {0: sum dz, 1: sum dz*xhat}]
template <bool RELU>
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float *__restrict__ dy, const float *__restrict__ y,
                                                        const float *__restrict__ x, const float *__restrict__ mean,
                                                        const float *__restrict__ invstd, int N, int C, int HW,
                                                        int nsplit, double *__restrict__ partial) {
  __shared__ double lds[4];
  const int c = blockIdx.x, s = blockIdx.y;
  const int n_lo = (int)((int64_t)N * s / nsplit), n_hi = (int)((int64_t)N * (s + 1) / nsplit);
  const float mu = mean[c], is = invstd[c];
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  const int hw4 = HW >> 2;
  const int work = (n_hi - n_lo) * hw4;
  int it = 0;
  BnWalk wk(n_lo, hw4);
#pragma unroll 2
  for (int e = threadIdx.x; e < work; e += 256, ++it, wk.next()) {
    const int n = wk.n, i = wk.i;
    const size_t off = ((size_t)n * C + c) * HW;
    float4 g = reinterpret_cast<const float4 *>(dy + off)[i];
    const float4 xv = reinterpret_cast<const float4 *>(x + off)[i];
    if (RELU) {
      const float4 yv = reinterpret_cast<const float4 *>(y + off)[i];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    s1 += (g.x + g.y) + (g.z + g.w);
    s2 += (g.x * ((xv.x - mu) * is) + g.y * ((xv.y - mu) * is)) + (g.z * ((xv.z - mu) * is) + g.w * ((xv.w - mu) * is));
    if ((it & 7) == 7) { d1 += s1; d2 += s2; s1 = 0.f; s2 = 0.f; }
  }
  d1 += s1; d2 += s2;
  const double t1 = salun_block_sum(d1, lds);
  const double t2 = salun_block_sum(d2, lds);
  if (threadIdx.x == 0) {
    partial[((size_t)c * nsplit + s) * 2 + 0] = t1;
    partial[((size_t)c * nsplit + s) * 2 + 1] = t2;
  }
}

// dx (and dres) for channel blockIdx.x, batch slice blockIdx.y; dbeta / dgamma are folded from the partials by every
// workgroup of the channel (fixed order), slice 0 publishes them (and accumulates into the parameters' .grad).
template <bool RELU, bool TRAIN, bool DRES>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dy, const float *__restrict__ y,
                                                      const float *__restrict__ x, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd,
                                                      const float *__restrict__ gamma,
                                                      const double *__restrict__ partial, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta, float *__restrict__ gacc,
                                                      float *__restrict__ bacc, float *__restrict__ dx,
                                                      float *__restrict__ dres, int N, int C, int HW, int nsplit,
                                                      float inv_m) {
  __shared__ float sg[2];
  const int c = blockIdx.x, s = blockIdx.y;
  if (threadIdx.x < 64) {
    double s1 = 0.0, s2 = 0.0;
    if ((int)threadIdx.x < nsplit) {
      s1 = partial[((size_t)c * nsplit + threadIdx.x) * 2 + 0];
      s2 = partial[((size_t)c * nsplit + threadIdx.x) * 2 + 1];
    }
    s1 = salun_wave_sum(s1);
    s2 = salun_wave_sum(s2);
    if (threadIdx.x == 0) {
      sg[0] = (float)s1;  // dbeta
      sg[1] = (float)s2;  // dgamma
      if (s == 0) {
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        if (gacc) gacc[c] += (float)s2;  // optional: accumulate straight into the parameters' .grad storage
        if (bacc) bacc[c] += (float)s1;
      }
    }
  }
  __syncthreads();
  const float mu = mean[c], is = invstd[c];
  const float gi = gamma[c] * is;
  const float kb = sg[0] * inv_m, kg = sg[1] * inv_m;
  const int n_lo = (int)((int64_t)N * s / nsplit), n_hi = (int)((int64_t)N * (s + 1) / nsplit);
  const int hw4 = HW >> 2;
  const int work = (n_hi - n_lo) * hw4;
  BnWalk wk(n_lo, hw4);
#pragma unroll 2
  for (int e = threadIdx.x; e < work; e += 256, wk.next()) {
    const int n = wk.n, i = wk.i;
    const size_t off = ((size_t)n * C + c) * HW;
    float4 g = reinterpret_cast<const float4 *>(dy + off)[i];
    if (RELU) {
      const float4 yv = reinterpret_cast<const float4 *>(y + off)[i];
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if (DRES) reinterpret_cast<float4 *>(dres + off)[i] = g;
    float4 o;
    if (TRAIN) {
      const float4 xv = reinterpret_cast<const float4 *>(x + off)[i];
      o.x = gi * (g.x - (kb + ((xv.x - mu) * is) * kg));
      o.y = gi * (g.y - (kb + ((xv.y - mu) * is) * kg));
      o.z = gi * (g.z - (kb + ((xv.z - mu) * is) * kg));
      o.w = gi * (g.w - (kb + ((xv.w - mu) * is) * kg));
    } else {
      o.x = gi * g.x; o.y = gi * g.y; o.z = gi * g.z; o.w = gi * g.w;
    }
    reinterpret_cast<float4 *>(dx + off)[i] = o;
  }
}

inline int bn_nsplit(int N, int C) {
  int ns = (1024 + C - 1) / C;  // aim for >= ~1024 workgroups
  if (ns > N) ns = N;
  if (ns > BN_MAX_SPLIT) ns = BN_MAX_SPLIT;
  if (ns < 1) ns = 1;
  return ns;
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT size_t salun_bn_workspace_bytes(int C) {
  return sizeof(double) * 2 * (size_t)(C > 0 ? C : 0) * BN_MAX_SPLIT;
}

// training != 0: batch statistics -> mean/invstd (saved for backward) and running-stat update (if pointers given);
// training == 0: mean/invstd from the running statistics.  Then y = [relu](gamma*(x-mean)*invstd + beta [+ res]).
SALUN_EXPORT int salun_bn_forward(const float *x, const float *res, float *y, const float *gamma, const float *beta,
                                  float *running_mean, float *running_var, long long *num_batches_tracked,
                                  float *save_mean, float *save_invstd, int N, int C, int HW, int training,
                                  double momentum, double eps, int relu, void *ws, size_t ws_bytes,
                                  salun_stream_t stream) {
  if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || N < 1 || C < 1 || HW < 4 || (HW & 3))
    return SALUN_EINVAL;
  if (!training && (!running_mean || !running_var)) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(y) || (res && !salun_aligned16(res))) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int ns = bn_nsplit(N, C);
  double *partial = static_cast<double *>(ws);
  if (training) {
    if (!ws || ws_bytes < salun_bn_workspace_bytes(C)) return SALUN_ENOSPC;
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(C, ns), dim3(256), 0, st, x, N, C, HW, ns, partial);
    SALUN_LAUNCH_CHECK();
  }
#define SALUN_BN_APPLY(RELU_, RES_, TRAIN_)                                                                       \
  hipLaunchKernelGGL((k_bn_apply<RELU_, RES_, TRAIN_>), dim3(C, ns), dim3(256), 0, st, x, res, y, partial,         \
                     save_mean, save_invstd, gamma, beta, running_mean, running_var, num_batches_tracked, N, C,  \
                     HW, ns, (float)eps, (float)momentum)
  const bool r = relu != 0, a = res != nullptr, t = training != 0;
  if (r && a && t) SALUN_BN_APPLY(true, true, true);
  else if (r && a) SALUN_BN_APPLY(true, true, false);
  else if (r && t) SALUN_BN_APPLY(true, false, true);
  else if (r) SALUN_BN_APPLY(true, false, false);
  else if (a && t) SALUN_BN_APPLY(false, true, true);
  else if (a) SALUN_BN_APPLY(false, true, false);
  else if (t) SALUN_BN_APPLY(false, false, true);
  else SALUN_BN_APPLY(false, false, false);
#undef SALUN_BN_APPLY
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// dgamma, dbeta always (this call's values); grad_*_acc (optional) += them; dx; dres (= dz) if non-NULL.
SALUN_EXPORT int salun_bn_backward(const float *dy, const float *y, const float *x, const float *gamma,
                                   const float *save_mean, const float *save_invstd, float *dx, float *dres,
                                   float *dgamma, float *dbeta, float *grad_gamma_acc, float *grad_beta_acc, int N,
                                   int C, int HW, int training, int relu, void *ws, size_t ws_bytes,
                                   salun_stream_t stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || N < 1 || C < 1 || HW < 4 ||
      (HW & 3) || (relu && !y))
    return SALUN_EINVAL;
  if (!ws || ws_bytes < salun_bn_workspace_bytes(C)) return SALUN_ENOSPC;
  if (!salun_aligned16(dy) || !salun_aligned16(x) || !salun_aligned16(dx) || (y && !salun_aligned16(y)) ||
      (dres && !salun_aligned16(dres)))
    return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int ns = bn_nsplit(N, C);
  double *partial = static_cast<double *>(ws);
  if (relu) hipLaunchKernelGGL(k_bn_bwd_partial<true>, dim3(C, ns), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, N, C, HW, ns, partial);
  else hipLaunchKernelGGL(k_bn_bwd_partial<false>, dim3(C, ns), dim3(256), 0, st, dy, y, x, save_mean, save_invstd, N, C, HW, ns, partial);
  SALUN_LAUNCH_CHECK();
  const float inv_m = (float)(1.0 / ((double)N * HW));
#define SALUN_BN_BWD(RELU_, TRAIN_, DRES_)                                                                          \
  hipLaunchKernelGGL((k_bn_bwd_apply<RELU_, TRAIN_, DRES_>), dim3(C, ns), dim3(256), 0, st, dy, y, x, save_mean,   \
                     save_invstd, gamma, partial, dgamma, dbeta, grad_gamma_acc, grad_beta_acc, dx, dres, N, C, HW, \
                     ns, inv_m)
  const bool r = relu != 0, t = training != 0, d = dres != nullptr;
  if (r && t && d) SALUN_BN_BWD(true, true, true);
  else if (r && t) SALUN_BN_BWD(true, true, false);
  else if (r && d) SALUN_BN_BWD(true, false, true);
  else if (r) SALUN_BN_BWD(true, false, false);
  else if (t && d) SALUN_BN_BWD(false, true, true);
  else if (t) SALUN_BN_BWD(false, true, false);
  else if (d) SALUN_BN_BWD(false, false, true);
  else SALUN_BN_BWD(false, false, false);
#undef SALUN_BN_BWD
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ====================================================================================================
// Fused GroupNorm (+ SiLU), NCHW fp32 — the `swish(norm(x))` prologue of every DDPM / SD ResNet block
// (DDPM/models/diffusion.py:36-41,101-128; SD openaimodel.py ResBlock in_layers / out_layers).
// One workgroup per (image, group): the group's (C/G)*HW values are contiguous.  Forward reads them once (kept in
// registers when they fit: <= 64 values per lane), reduces sum / sum of squares, applies gamma/beta and SiLU, writes
// once.  Backward recomputes y = gn(x) and sigma(y) instead of saving them, reduces the per-channel sums needed for
// dgamma/dbeta (per-image partials, folded over the batch in a fixed order by a second small kernel) and the two
// group sums of the input gradient, and writes dx — x and dz are read once.
namespace {

constexpr int GN_ITEMS = 16;            // most float4 per lane held in registers (ITEMS = 1, 2, 4, 8, 16; 0 = re-read mode)
constexpr int GN_MAX_SEG = 1024;        // LDS slots for segment sums

__device__ __forceinline__ float gn_sigmoid(float y) { return 1.0f / (1.0f + expf(-y)); }

// sum over aligned groups of `r` lanes (r a power of two <= 64); every lane of the group gets the total
__device__ __forceinline__ float seg_sum(float v, int r) {
  for (int off = 1; off < r; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool SILU, int ITEMS>
__global__ __launch_bounds__(256) void k_gn_fwd(const float *__restrict__ x, float *__restrict__ y,
                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                float *__restrict__ mean, float *__restrict__ rstd, int C, int HW,
                                                int G, float eps) {
  constexpr bool CACHED = ITEMS > 0;
  __shared__ double lds[4];
  __shared__ float st[2];
  const int ng = blockIdx.x, n = ng / G, g = ng - n * G;
  const int cpg = C / G;
  const int L = cpg * HW, nvec = L >> 2;
  const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
  const float4 *xv = reinterpret_cast<const float4 *>(x + base);
  float4 v[CACHED ? ITEMS : 1];
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {  // all loads in flight before the first use
      const int e = threadIdx.x + i * 256;
      if (e < nvec) v[i] = xv[e];
    }
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < nvec) {
        s1 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        s2 += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
    }
    d1 = s1; d2 = s2;
  } else {
    int it = 0;
    for (int e = threadIdx.x; e < nvec; e += 256, ++it) {
      const float4 t = xv[e];
      s1 += (t.x + t.y) + (t.z + t.w);
      s2 += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
      if ((it & 15) == 15) { d1 += s1; d2 += s2; s1 = 0.f; s2 = 0.f; }
    }
    d1 += s1; d2 += s2;
  }
  const double t1 = salun_block_sum(d1, lds);
  const double t2 = salun_block_sum(d2, lds);
  if (threadIdx.x == 0) {
    const double mu = t1 / L;
    double var = t2 / L - mu * mu;
    if (var < 0.0) var = 0.0;
    st[0] = (float)mu;
    st[1] = (float)(1.0 / sqrt(var + (double)eps));
    mean[ng] = st[0];
    rstd[ng] = st[1];
  }
  __syncthreads();
  const float mu = st[0], rs = st[1];
  const int hw4 = HW >> 2;
  float4 *yv = reinterpret_cast<float4 *>(y + base);
  auto apply = [&](int e, float4 t) {
    const int c = g * cpg + e / hw4;
    const float a = rs * gamma[c];
    const float b = beta[c] - mu * a;
    float4 o;
    o.x = t.x * a + b; o.y = t.y * a + b; o.z = t.z * a + b; o.w = t.w * a + b;
    if (SILU) {
      o.x = o.x * gn_sigmoid(o.x); o.y = o.y * gn_sigmoid(o.y);
      o.z = o.z * gn_sigmoid(o.z); o.w = o.w * gn_sigmoid(o.w);
    }
    yv[e] = o;
  };
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < nvec) apply(e, v[i]);
    }
  } else {
    for (int e = threadIdx.x; e < nvec; e += 256) apply(e, xv[e]);
  }
}

// Backward.  seg: aligned run of r = min(HW/4, 64) lanes inside one channel; segment sums go to LDS, channel sums
// are folded from them by one lane per channel, group sums from the channel sums — all in fixed order.
// EXTRA (the diffusion ResnetBlock node, resblock.py): `addend` (same shape as dx, or null) is added to dx before the
// store — the skip branch's gradient, instead of a separate add pass — and `nk_sum` (N*C floats, or null) receives
// sum_hw of the dx written per (image, channel): the gradient of the per-image channel bias the producing convolution
// added (the time/class embedding projection) and, folded over n by k_gn_bwd_final, of that convolution's bias.
template <bool SILU, int ITEMS, bool EXTRA>
__global__ __launch_bounds__(256) void k_gn_bwd(const float *__restrict__ dz, const float *__restrict__ x,
                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                float *__restrict__ dx, float *__restrict__ part_dgamma,
                                                float *__restrict__ part_dbeta, int C, int HW, int G,
                                                const float *__restrict__ addend, float *__restrict__ nk_sum) {
  constexpr bool CACHED = ITEMS > 0;
  __shared__ float seg_g[GN_MAX_SEG], seg_b[GN_MAX_SEG];
  __shared__ float grp[2];
  const int ng = blockIdx.x, n = ng / G, g = ng - n * G;
  const int cpg = C / G;
  const int L = cpg * HW, nvec = L >> 2, hw4 = HW >> 2;
  const size_t base = ((size_t)n * C + (size_t)g * cpg) * HW;
  const float4 *xv = reinterpret_cast<const float4 *>(x + base);
  const float4 *gv = reinterpret_cast<const float4 *>(dz + base);
  const float mu = mean[ng], rs = rstd[ng];
  const int r = hw4 < 64 ? hw4 : 64;          // lanes per segment (power of two)
  const int segs_per_ch = hw4 / r;            // >= 1
  float4 xc[CACHED ? ITEMS : 1], dc[CACHED ? ITEMS : 1];
  float gac[CACHED ? ITEMS : 1], bec[CACHED ? ITEMS : 1];  // gamma / beta of each cached float4's channel

  // dy (gradient w.r.t. the normalised, affine output) of one float4, from x and dz
  auto dy_of = [&](float ga, float be, float4 xt, float4 gt, float4 &xh) {
    xh.x = (xt.x - mu) * rs; xh.y = (xt.y - mu) * rs; xh.z = (xt.z - mu) * rs; xh.w = (xt.w - mu) * rs;
    if (SILU) {
      const float y0 = xh.x * ga + be, y1 = xh.y * ga + be, y2 = xh.z * ga + be, y3 = xh.w * ga + be;
      const float s0 = gn_sigmoid(y0), s1 = gn_sigmoid(y1), s2 = gn_sigmoid(y2), s3 = gn_sigmoid(y3);
      gt.x = gt.x * (s0 * (1.0f + y0 * (1.0f - s0)));
      gt.y = gt.y * (s1 * (1.0f + y1 * (1.0f - s1)));
      gt.z = gt.z * (s2 * (1.0f + y2 * (1.0f - s2)));
      gt.w = gt.w * (s3 * (1.0f + y3 * (1.0f - s3)));
    }
    return gt;
  };
  auto seg_accumulate = [&](int e, float4 dyv, float4 xh, bool valid) {
    float sg = valid ? (dyv.x * xh.x + dyv.y * xh.y) + (dyv.z * xh.z + dyv.w * xh.w) : 0.f;
    float sb = valid ? (dyv.x + dyv.y) + (dyv.z + dyv.w) : 0.f;
    sg = seg_sum(sg, r);
    sb = seg_sum(sb, r);
    if (valid && (threadIdx.x & (r - 1)) == 0) {
      seg_g[e / r] = sg;
      seg_b[e / r] = sb;
    }
  };
  const int nround = (nvec + 255) / 256;  // every lane walks the same number of rounds (shuffles need all lanes)
  if (CACHED) {
    // every load of the group is issued before the first use (2 * ITEMS float4 in flight per lane): with the loads
    // inside the reduction rounds each round waited for its own two loads and the kernel ran at a third of HBM speed
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < nvec) {
        xc[i] = xv[e];
        dc[i] = gv[e];
        const int c = g * cpg + e / hw4;
        gac[i] = gamma[c];
        bec[i] = beta[c];
      }
    }
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {
      if (i < nround) {
        const int e = threadIdx.x + i * 256;
        const bool valid = e < nvec;
        float4 xh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) dc[i] = dy_of(gac[i], bec[i], xc[i], dc[i], xh);
        seg_accumulate(valid ? e : 0, valid ? dc[i] : xh, xh, valid);
      }
    }
  } else {
    for (int i = 0; i < nround; ++i) {
      const int e = threadIdx.x + i * 256;
      const bool valid = e < nvec;
      float4 xh = make_float4(0.f, 0.f, 0.f, 0.f), d = xh;
      if (valid) {
        const int c = g * cpg + e / hw4;
        d = dy_of(gamma[c], beta[c], xv[e], gv[e], xh);
      }
      seg_accumulate(valid ? e : 0, d, xh, valid);
    }
  }
  __syncthreads();
  // channel sums (one lane per channel), published as per-image partials
  float my_g = 0.f, my_b = 0.f;
  if ((int)threadIdx.x < cpg) {
    for (int s = 0; s < segs_per_ch; ++s) {
      my_g += seg_g[threadIdx.x * segs_per_ch + s];
      my_b += seg_b[threadIdx.x * segs_per_ch + s];
    }
    const int c = g * cpg + threadIdx.x;
    part_dgamma[(size_t)n * C + c] = my_g;
    part_dbeta[(size_t)n * C + c] = my_b;
  }
  __syncthreads();
  if ((int)threadIdx.x < cpg) {  // reuse the segment arrays for the gamma-weighted channel sums
    const float ga = gamma[g * cpg + threadIdx.x];
    seg_g[threadIdx.x] = my_g * ga;
    seg_b[threadIdx.x] = my_b * ga;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int c = 0; c < cpg; ++c) { b += seg_g[c]; a += seg_b[c]; }
    grp[0] = a / (float)L;  // mean over the group of dy*gamma
    grp[1] = b / (float)L;  // mean over the group of dy*gamma*xhat
  }
  __syncthreads();
  const float ma = grp[0], mb = grp[1];
  float4 *ov = reinterpret_cast<float4 *>(dx + base);
  const float4 *av = reinterpret_cast<const float4 *>(EXTRA && addend ? addend + base : nullptr);
  auto emit = [&](int e, float ga, float4 xt, float4 dyv) {
    float4 o;
    o.x = rs * (dyv.x * ga - ma - ((xt.x - mu) * rs) * mb);
    o.y = rs * (dyv.y * ga - ma - ((xt.y - mu) * rs) * mb);
    o.z = rs * (dyv.z * ga - ma - ((xt.z - mu) * rs) * mb);
    o.w = rs * (dyv.w * ga - ma - ((xt.w - mu) * rs) * mb);
    if (EXTRA && av) {
      const float4 ad = av[e];
      o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
    }
    ov[e] = o;
    return o;
  };
  const bool want_nk = EXTRA && nk_sum != nullptr;  // uniform
  auto nk_accumulate = [&](int e, float4 o, bool valid) {  // segment sums of the written dx -> seg_g (free again here)
    float sv = valid ? (o.x + o.y) + (o.z + o.w) : 0.f;
    sv = seg_sum(sv, r);
    if (valid && (threadIdx.x & (r - 1)) == 0) seg_g[e / r] = sv;
  };
  if (CACHED) {
#pragma unroll
    for (int i = 0; i < (CACHED ? ITEMS : 1); ++i) {
      const int e = threadIdx.x + i * 256;
      const bool valid = e < nvec;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) o = emit(e, gac[i], xc[i], dc[i]);
      if (EXTRA) {
        if (want_nk && i < nround) nk_accumulate(valid ? e : 0, o, valid);
      }
    }
  } else {
    for (int i = 0; i < nround; ++i) {
      const int e = threadIdx.x + i * 256;
      const bool valid = e < nvec;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        float4 xh;
        const float4 xt = xv[e];
        const int c = g * cpg + e / hw4;
        const float ga = gamma[c];
        o = emit(e, ga, xt, dy_of(ga, beta[c], xt, gv[e], xh));
      }
      if (EXTRA) {
        if (want_nk) nk_accumulate(valid ? e : 0, o, valid);
      }
    }
  }
  if (EXTRA) {
    if (want_nk) {
      __syncthreads();
      if ((int)threadIdx.x < cpg) {
        float t = 0.f;
        for (int sgi = 0; sgi < segs_per_ch; ++sgi) t += seg_g[threadIdx.x * segs_per_ch + sgi];
        nk_sum[(size_t)n * C + g * cpg + threadIdx.x] = t;
      }
    }
  }
}

// dgamma[c] = sum_n part[n][c] (fixed order), optionally also accumulated into the parameter's .grad storage.
// Block = 32 channels x 8 lanes over n (a single thread per channel walked N = 128 rows one dependent load at a time).
__global__ __launch_bounds__(256) void k_gn_bwd_final(const float *__restrict__ part_dgamma,
                                                      const float *__restrict__ part_dbeta, int N, int C,
                                                      float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                      float *__restrict__ gacc, float *__restrict__ bacc,
                                                      const float *__restrict__ part_nk /*[N][C] or null*/,
                                                      float *__restrict__ csum /*[C] or null: = sum_n part_nk*/,
                                                      float *__restrict__ csum_acc /*[C] or null: += the same*/) {
  __shared__ double s_g[8][33], s_b[8][33], s_k[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double sg = 0.0, sb = 0.0, sk = 0.0;
  if (c < C)
    for (int n = ry; n < N; n += 8) {
      sg += (double)part_dgamma[(size_t)n * C + c];
      sb += (double)part_dbeta[(size_t)n * C + c];
      if (part_nk) sk += (double)part_nk[(size_t)n * C + c];
    }
  s_g[ry][cx] = sg; s_b[ry][cx] = sb; s_k[ry][cx] = sk;
  __syncthreads();
  if (ry != 0 || c >= C) return;
  for (int q = 1; q < 8; ++q) { sg += s_g[q][cx]; sb += s_b[q][cx]; sk += s_k[q][cx]; }
  dgamma[c] = (float)sg;
  dbeta[c] = (float)sb;
  if (gacc) gacc[c] += (float)sg;
  if (bacc) bacc[c] += (float)sb;
  if (part_nk && csum) csum[c] = (float)sk;
  if (part_nk && csum_acc) csum_acc[c] += (float)sk;
}

// float4 per lane kept in registers: the smallest of 1, 2, 4, 8, 16 covering the group; 0 = too large, re-read mode
inline int gn_items(int C, int HW, int G) {
  const int64_t nvec = (int64_t)(C / G) * HW / 4;
  for (int it = 1; it <= GN_ITEMS; it <<= 1)
    if (nvec <= (int64_t)it * 256) return it;
  return 0;
}

inline bool gn_shape_ok(int N, int C, int HW, int G) {
  if (N < 1 || C < 1 || G < 1 || C % G != 0 || HW < 4 || (HW & (HW - 1)) != 0) return false;  // HW: power of two >= 4
  const int cpg = C / G;
  if (cpg > 256) return false;
  const int hw4 = HW >> 2, r = hw4 < 64 ? hw4 : 64;
  return (int64_t)cpg * hw4 / r <= GN_MAX_SEG && (int64_t)cpg * HW < (1 << 30);
}

}  // namespace

SALUN_EXPORT size_t salun_gn_workspace_bytes(int N, int C) {
  return sizeof(float) * 2 * (size_t)(N > 0 ? N : 0) * (size_t)(C > 0 ? C : 0);
}

// y = [silu](GroupNorm(x; G groups, gamma, beta, eps)); save_mean / save_rstd: N*G floats (outputs, for backward)
SALUN_EXPORT int salun_gn_forward(const float *x, float *y, const float *gamma, const float *beta, float *save_mean,
                                  float *save_rstd, int N, int C, int HW, int G, double eps, int silu,
                                  salun_stream_t stream) {
  if (!x || !y || !gamma || !beta || !save_mean || !save_rstd || !gn_shape_ok(N, C, HW, G)) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(y)) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  const int items = gn_items(C, HW, G);
#define SALUN_GN_FWD(S_, I_) \
  hipLaunchKernelGGL((k_gn_fwd<S_, I_>), dim3(N * G), dim3(256), 0, st, x, y, gamma, beta, save_mean, save_rstd, C, HW, G, (float)eps)
#define SALUN_GN_FWD_I(S_)                    \
  switch (items) {                            \
    case 1: SALUN_GN_FWD(S_, 1); break;       \
    case 2: SALUN_GN_FWD(S_, 2); break;       \
    case 4: SALUN_GN_FWD(S_, 4); break;       \
    case 8: SALUN_GN_FWD(S_, 8); break;       \
    case 16: SALUN_GN_FWD(S_, 16); break;     \
    default: SALUN_GN_FWD(S_, 0); break;      \
  }
  if (silu) { SALUN_GN_FWD_I(true) } else { SALUN_GN_FWD_I(false) }
#undef SALUN_GN_FWD_I
#undef SALUN_GN_FWD
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// dx, dgamma, dbeta of  z = [silu](GroupNorm(x))  given dz; y and sigma(y) are recomputed from x.
// Fused form: dx += addend (optional, [N,C,HW]); nk_sum (optional, N*C floats) = sum_hw dx per (image, channel);
// csum / csum_acc (optional, C floats, need nk_sum) = / += sum_n nk_sum.
SALUN_EXPORT int salun_gn_backward_fused(const float *dz, const float *x, const float *gamma, const float *beta,
                                         const float *save_mean, const float *save_rstd, const float *addend,
                                         float *dx, float *dgamma, float *dbeta, float *grad_gamma_acc,
                                         float *grad_beta_acc, float *nk_sum, float *csum, float *csum_acc, int N,
                                         int C, int HW, int G, int silu, void *ws, size_t ws_bytes,
                                         salun_stream_t stream) {
  if (!dz || !x || !gamma || !beta || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || !gn_shape_ok(N, C, HW, G))
    return SALUN_EINVAL;
  if ((csum || csum_acc) && !nk_sum) return SALUN_EINVAL;
  if (addend == dx) return SALUN_EINVAL;
  if (!ws || ws_bytes < salun_gn_workspace_bytes(N, C)) return SALUN_ENOSPC;
  if (!salun_aligned16(dz) || !salun_aligned16(x) || !salun_aligned16(dx) || (addend && !salun_aligned16(addend)))
    return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  float *pg = static_cast<float *>(ws), *pb = pg + (size_t)N * C;
  const int items = gn_items(C, HW, G);
  const bool extra = addend != nullptr || nk_sum != nullptr;
#define SALUN_GN_BWD(S_, I_, E_) \
  hipLaunchKernelGGL((k_gn_bwd<S_, I_, E_>), dim3(N * G), dim3(256), 0, st, dz, x, gamma, beta, save_mean, save_rstd, dx, pg, pb, C, HW, G, addend, nk_sum)
#define SALUN_GN_BWD_I(S_, E_)                    \
  switch (items) {                                \
    case 1: SALUN_GN_BWD(S_, 1, E_); break;       \
    case 2: SALUN_GN_BWD(S_, 2, E_); break;       \
    case 4: SALUN_GN_BWD(S_, 4, E_); break;       \
    case 8: SALUN_GN_BWD(S_, 8, E_); break;       \
    case 16: SALUN_GN_BWD(S_, 16, E_); break;     \
    default: SALUN_GN_BWD(S_, 0, E_); break;      \
  }
  if (silu) {
    if (extra) { SALUN_GN_BWD_I(true, true) } else { SALUN_GN_BWD_I(true, false) }
  } else {
    if (extra) { SALUN_GN_BWD_I(false, true) } else { SALUN_GN_BWD_I(false, false) }
  }
#undef SALUN_GN_BWD_I
#undef SALUN_GN_BWD
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gn_bwd_final, dim3((C + 31) / 32), dim3(256), 0, st, pg, pb, N, C, dgamma, dbeta,
                     grad_gamma_acc, grad_beta_acc, nk_sum, csum, csum_acc);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_gn_backward(const float *dz, const float *x, const float *gamma, const float *beta,
                                   const float *save_mean, const float *save_rstd, float *dx, float *dgamma,
                                   float *dbeta, float *grad_gamma_acc, float *grad_beta_acc, int N, int C, int HW,
                                   int G, int silu, void *ws, size_t ws_bytes, salun_stream_t stream) {
  return salun_gn_backward_fused(dz, x, gamma, beta, save_mean, save_rstd, nullptr, dx, dgamma, dbeta, grad_gamma_acc,
                                 grad_beta_acc, nullptr, nullptr, nullptr, N, C, HW, G, silu, ws, ws_bytes, stream);
}
