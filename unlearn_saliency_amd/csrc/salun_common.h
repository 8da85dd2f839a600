// salun_common.h — shared device/host helpers for the gfx950 SalUn kernels.
// CDNA4 only: 64-lane wavefronts, 256-thread workgroups (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/salun.h"

#define SALUN_WAVE 64
#define SALUN_BLOCK 256
// 256 CUs x 8 resident 256-thread blocks: the grid-stride cap for HBM-bound kernels.
#define SALUN_MAX_GRID 2048

#define SALUN_EXPORT extern "C" __attribute__((visibility("default")))

static inline hipStream_t salun_hip_stream(salun_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Every launch goes through this: a failed launch becomes SALUN_EIO, never an abort.
#define SALUN_LAUNCH_CHECK()                         \
  do {                                               \
    hipError_t _e = hipGetLastError();               \
    if (_e != hipSuccess) return SALUN_EIO;          \
  } while (0)

static inline int salun_grid_for(int64_t work_items, int per_block) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > SALUN_MAX_GRID) b = SALUN_MAX_GRID;
  return static_cast<int>(b);
}

// Bit index of the calling thread's current device.  hipFuncSetAttribute's dynamic-LDS opt-in is per DEVICE: "done
// once" flags are kept as one bit per device so a process that drives several GPUs configures each of them.
static inline int salun_device_bit() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev & 63;  // (one node holds 8 devices; ids >= 64 would alias — the flags below are "set an attribute once", benign)
}
// "Done once per device" flags (dynamic-LDS opt-ins): one bit per device, read and published atomically (one host
// thread per GPU makes a plain read-modify-write a data race).  The bit is published AFTER the work, so a racing thread
// either repeats the (idempotent) attribute call or sees it done — never launches ahead of it.
static inline bool salun_once_needed(const unsigned long long *done) {
  return ((__atomic_load_n(done, __ATOMIC_ACQUIRE) >> salun_device_bit()) & 1ull) == 0;
}
static inline void salun_once_mark(unsigned long long *done) {
  (void)__atomic_fetch_or(done, 1ull << salun_device_bit(), __ATOMIC_RELEASE);
}

static inline bool salun_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool salun_aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// ---- wave / block reductions (wave = 64 lanes) ---------------------------------
__device__ __forceinline__ double salun_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ unsigned long long salun_wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// Block sum for 256 threads; result valid in thread 0.  `lds` needs 4 doubles.
__device__ __forceinline__ double salun_block_sum(double v, double *lds) {
  v = salun_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) lds[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  __syncthreads();
  return r;
}

// ---- counter-based generator (identical in oracle/salun_oracle.c) -----------------
__host__ __device__ __forceinline__ uint64_t salun_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ float salun_u01(uint64_t seed, uint64_t i) {
  return static_cast<float>(salun_splitmix64(seed + i) >> 40) * (1.0f / 16777216.0f);
}
__host__ __device__ __forceinline__ float salun_ih12(uint64_t seed, uint64_t i) {
  int32_t s = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    uint64_t r = salun_splitmix64(seed + 3ull * i + static_cast<uint64_t>(j));
    s += static_cast<int32_t>(r & 0xFFFF) + static_cast<int32_t>((r >> 16) & 0xFFFF) +
         static_cast<int32_t>((r >> 32) & 0xFFFF) + static_cast<int32_t>((r >> 48) & 0xFFFF);
  }
  return static_cast<float>(s - 393210) * (1.0f / 65536.0f);
}

// clip_grad_norm_ coefficient from a squared norm (torch.nn.utils.clip_grad_norm_:
// clip_coef = max_norm / (total_norm + 1e-6), clamped to 1.0).
__host__ __device__ __forceinline__ float salun_clip_coef(float sqnorm, float max_norm) {
  float total = sqrtf(sqnorm);
  float c = max_norm / (total + 1e-6f);
  return c > 1.0f ? 1.0f : c;
}

// dropout: keep iff a 24-bit draw >= round(p * 2^24)  (integer comparison: identical on host, device and oracle)
static inline uint32_t salun_dropout_threshold(double p) {
  double t = p * 16777216.0 + 0.5;
  if (t < 0.0) t = 0.0;
  if (t > 16777216.0) t = 16777216.0;
  return (uint32_t)t;
}
