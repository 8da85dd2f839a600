// Probe (gfx950): (1) lane mapping of ds_read_b64_tr_b16, (2) operand/result mapping of v_mfma_f32_32x32x16_bf16.
// hipcc --offload-arch=gfx950 -O2 -o bf16_probe bf16_probe.hip && ./bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_tr(uint16_t *out, int row_stride_elems) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int lane = threadIdx.x;
  // lane l of a 16-lane group addresses row (l & 15), 4 consecutive elements; groups take column blocks of 4
  const uint32_t addr = (uint32_t)(uintptr_t)(&lds[(lane & 15) * row_stride_elems + (lane >> 4) * 4]);
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

__global__ void k_mfma(const float *A /*32x16*/, const float *B /*16x32*/, float *D /*32x32*/) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    const int k = 8 * (lane >> 5) + i;
    a[i] = (__bf16)A[(lane & 31) * 16 + k];
    b[i] = (__bf16)B[k * 32 + (lane & 31)];
  }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
    D[row * 32 + col] = c[r];
  }
}

int main() {
  uint16_t *d_out; hipMalloc(&d_out, 64 * 4 * 2);
  for (int rs : {16, 32, 64}) {
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, d_out, rs);
    uint16_t h[256]; hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
    printf("tr_b16 row_stride=%d: lane -> 4 x (row,col) of the element received\n", rs);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (%2d,%2d)", h[l * 4 + j] / rs, h[l * 4 + j] % rs);
      printf("\n");
    }
  }
  std::vector<float> A(512), B(512), D(1024), R(1024, 0.f);
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 11 - 5); B[i] = (float)((i * 5 + 1) % 13 - 6); }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j]; R[i * 32 + j] = s; }
  float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 1024; ++i) bad += D[i] != R[i];
  printf("mfma_f32_32x32x16_bf16 with A row=lane&31,k=8*(lane>>5)+i; B col=lane&31: mismatches = %d\n", bad);
  return 0;
}
