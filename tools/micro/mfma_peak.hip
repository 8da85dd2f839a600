// Ceiling probe for the fp32 matrix path on gfx950: v_mfma_f32_32x32x2_f32 issue rate with (a) register operands only,
// (b) the conv kernels' operand pattern (1 + 9 LDS reads per 9 MFMAs, one step ahead).  Prints TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float *out, int iters, int stride) {
  __shared__ float lds[64 * 257];
  for (int i = threadIdx.x; i < 64 * 257; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  const int lane = threadIdx.x & 63;
  const float *row = lds + (lane & 31) * 257 + (lane >> 5);
  float a = 1.0f + lane * 1e-3f, b[NACC];
  for (int t = 0; t < NACC; ++t) b[t] = 0.5f + t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float an = a, bn[NACC];
      if (LDS) {
        const float *p = row + ((j * 2 + it) & 63) * stride;
        an = p[128];
#pragma unroll
        for (int t = 0; t < NACC; ++t) bn[t] = p[(t / 3) * 34 + (t % 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (LDS) {
        a = an;
#pragma unroll
        for (int t = 0; t < NACC; ++t) b[t] = bn[t];
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int v = 0; v < 16; ++v) s += acc[t][v];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int grid, int iters) {
  float *out;
  hipMalloc(&out, sizeof(float) * grid * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(grid), dim3(256), 0, 0, out, iters, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)grid * 4 * iters * 32 * NACC * 4096.0;
  printf("%-34s grid %5d  %8.3f ms  %7.1f TFLOP/s\n", name, grid, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<9, false>("9 acc, register operands", 256, 2000);
  run<9, false>("9 acc, register operands", 512, 2000);
  run<4, false>("4 acc, register operands", 256, 4000);
  run<4, false>("4 acc, register operands", 1024, 1000);
  run<9, true>("9 acc, LDS operands 1 step ahead", 256, 2000);
  run<4, true>("4 acc, LDS operands 1 step ahead", 256, 4000);
  run<4, true>("4 acc, LDS operands 1 step ahead", 1024, 1000);
  return 0;
}
