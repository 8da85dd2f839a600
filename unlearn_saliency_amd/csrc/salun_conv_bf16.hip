// salun_conv_bf16.hip — K11: bf16 2-D convolution forward / backward-data / backward-weight as implicit GEMM on the
// CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16: 8 bf16 per lane per operand, fp32 accumulators, 2.5 PFLOP/s dense).
//
// This is the convolution of the Stable-Diffusion U-Net in its bf16 configuration (BASELINE.json configs[4]; reference
// modules SD/ldm/modules/diffusionmodules/openaimodel.py:80-140 Upsample/Downsample, :163-275 ResBlock,
// SD/ldm/modules/attention.py:218-260 proj_in/proj_out).  Round 1 ran that configuration on library kernels.
//
// Layout (MI355X-first, not the reference's): activations are NHWC bf16 — `[pixel][channel]`, the GEMM's natural
// operand: a pixel's channels are one contiguous run, so the im2col gather is a 16-byte copy per (pixel, tap, 8
// channels) and the transformer blocks' `b (h w) c` token view is the same memory.  Master weights stay fp32 OIHW in
// the flat arena (the unit the saliency ranking and the masked update see); `salun_conv2d_bf16_pack_weights` writes the
// bf16 image `[k][tap][c]` the kernels read (once per optimizer step).
//
//   forward        D[pixel][k] = sum_{tap,c} X[pixel+tap][c] * Wp[k][tap][c]       A = pixels (rows), B = weights
//   backward-data  D[pixel][c] = sum_{tap,k} dY[pixel-tap][k] * Wp[k][tap][c]      same kernel; the B tile is staged
//                  as [k][c] rows (contiguous in Wp) and read TRANSPOSED by ds_read_b64_tr_b16, so no second weight
//                  image exists; a stride-2 convolution walks the zero-upsampled dY (taps on odd positions read 0)
//   backward-weight dW[k][tap][c] = sum_pixel dY[pixel][k] * X[pixel+tap][c]       reduction over pixels: both operands
//                  are needed pixel-major per lane, i.e. transposed — both tiles sit in LDS as they are in memory
//                  ([pixel][32 channels], 64-byte rows = conflict-free for the transposing read) and every tap is an
//                  immediate offset into the same input patch: 9 accumulators per wave, one staged patch per 64 pixels.
//                  Pixel ranges are split over workgroups -> fp32 partials -> fixed-order reduce into OIHW (no atomics).
//
// Workgroup = 4 or 8 waves, each wave a 64x64 result tile (2x2 MFMA tiles, 64 fp32 accumulators); LDS double
// buffered, global loads of stage i+1 in flight under the MFMAs of stage i, one barrier per stage.
#include "salun_common.h"
#include <cstdlib>

#ifndef SALUN_BF16_TILE
#define SALUN_BF16_TILE 0  // lab builds (tools/_run_bf16_lab.sh) pin one tile shape; 0 = choose per problem
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

// round-to-nearest-even, NaN -> quiet NaN (what torch's float -> bfloat16 does)
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }  // v_cvt_pk_bf16_f32: RNE
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// 8 bf16 (one MFMA operand) = two transposing reads of 4: lane i of a 16-lane group receives column i of the
// [4 rows][16 columns] block its group addresses (probe: tools/micro/bf16_probe.hip); `second` is the byte distance
// of rows +4.
__device__ __forceinline__ bf16x8 tr_operand(const char *lds_base, uint32_t byte_off, int second) {
  lds_s16x4_ptr p0 = (lds_s16x4_ptr)(lds_base + byte_off);
  lds_s16x4_ptr p1 = (lds_s16x4_ptr)(lds_base + byte_off + second);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}

// ------------------------------------------------------------------------------------------------ pack
// fp32 OIHW [K][C][RS] -> bf16 [K][RS][C]; one thread writes 8 consecutive channels (16 B).
__global__ __launch_bounds__(256) void k_pack_w(const float *__restrict__ w, uint16_t *__restrict__ wp, int K, int C,
                                                int RS) {
  const int c8n = C >> 3;
  const int64_t total = (int64_t)K * RS * c8n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    const int64_t kr = i / c8n;
    const int rs = (int)(kr % RS);
    const int64_t k = kr / RS;
    const float *src = w + (k * C + (int64_t)c8 * 8) * RS + rs;
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack2(src[(2 * j) * RS], src[(2 * j + 1) * RS]);
    *reinterpret_cast<uint4 *>(wp + (kr * C + (int64_t)c8 * 8)) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// The same for a whole model in a few launches (round 6): after an optimizer step every weight image is stale, and one
// launch per layer was 256 launches of ~13 us per SD step, most of them too small to fill the chip.  blockIdx.y = job;
// `transposed` jobs (R = 1) write the [C][K] image the Linear layers' input-gradient GEMM reads (k_pack_bf16_t's tiles).
struct PackJobsB {
  struct J { const float *w; uint16_t *wp; int K, C, RS, transposed; } j[SALUN_BF16_PACK_MAX_JOBS];
};
__global__ __launch_bounds__(256) void k_pack_jobs(const PackJobsB jobs) {
  __shared__ float tile[32][33];
  const float *__restrict__ w = jobs.j[blockIdx.y].w;
  uint16_t *__restrict__ wp = jobs.j[blockIdx.y].wp;
  const int K = jobs.j[blockIdx.y].K, C = jobs.j[blockIdx.y].C, RS = jobs.j[blockIdx.y].RS;
  if (!jobs.j[blockIdx.y].transposed) {
    // one thread = 8 channels of one filter, all taps: 8 * RS CONSECUTIVE floats of the OIHW weights (k_pack_w reads
    // them as 8 * RS separate dwords at a stride of RS: a third of the streaming rate on the 3x3 layers), written as RS
    // 16-byte pieces of the image, each coalesced over the threads of a wave
    const int c8n = C >> 3;
    const int64_t total = (int64_t)K * c8n;
    const bool al16 = (reinterpret_cast<uintptr_t>(w) & 15) == 0;  // (a view at any 4-byte offset of the flat arena)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const int c8 = (int)(i % c8n);
      const int64_t k = i / c8n;
      const float *src = w + (k * C + (int64_t)c8 * 8) * RS;
      uint16_t *dst = wp + (k * RS) * C + (int64_t)c8 * 8;
      if (RS == 9) {
        float v[72];
        if (al16) {
#pragma unroll
          for (int q = 0; q < 18; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(src)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 72; ++q) v[q] = src[q];
        }
#pragma unroll
        for (int rs = 0; rs < 9; ++rs) {
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = pack2(v[(2 * q) * 9 + rs], v[(2 * q + 1) * 9 + rs]);
          *reinterpret_cast<uint4 *>(dst + (int64_t)rs * C) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      } else {  // RS == 1
        float v[8];
        if (al16) {
          const float4 a = reinterpret_cast<const float4 *>(src)[0], b = reinterpret_cast<const float4 *>(src)[1];
          v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = src[q];
        }
        *reinterpret_cast<uint4 *>(dst) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
      }
    }
    return;
  }
  // w [K][C] fp32 -> wp [C][K] bf16, 32 x 32 tiles through LDS
  const int tc = (C + 31) / 32, tk = (K + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.x; t < tc * tk; t += gridDim.x) {
    const int c0 = (t % tc) * 32, k0 = (t / tc) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + ty + 8 * r, c = c0 + tx;
      tile[ty + 8 * r][tx] = (k < K && c < C) ? w[(size_t)k * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = c0 + ty + 8 * r, k = k0 + tx;
      if (c < C && k < K) wp[(size_t)c * K + k] = f2bf(tile[tx][ty + 8 * r]);
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------- forward / backward-data
struct IgArgs {
  const uint16_t *x;       // [N][H][W][Cin] bf16 (forward: input; backward-data: dY)
  const uint16_t *wp;      // packed weights [Kf][RS][Cf] bf16 (Kf, Cf = the FORWARD convolution's out / in channels)
  const float *bias;       // [Kout] fp32 or null
  const float *nbias;      // [N][Kout] fp32 or null (per-image channel offset: the time-embedding term of a ResBlock)
  const uint16_t *addend;  // [M][Kout] bf16 or null (residual branch)
  uint16_t *y;             // [M][Kout] bf16
  int M;                   // output pixels N*OH*OW
  int H, W, Cin;           // extent and channels of x
  int OH, OW, Kout;        // extent and channels of y
  int R, stride, pad, up;  // walk: virtual row = oh*stride - pad + r; `up` = 2 reads the zero-upsampled x (stride-2 dgrad)
  int Kf, Cf;              // forward dims of the packed weights
  float *part;             // split reduction: fp32 partial tiles [split][M][Kout] (null when gridDim.z == 1)
  int stages_per_split;    // reduction stages (tap x channel chunk) per blockIdx.z
  uint32_t x_bytes, w_bytes;  // extents of x and wp (buffer descriptors: reads past them return zeros)
};

template <int WM, int WN, int BK, bool BTR>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu((WM * WN == 4 && BK == 32) ? 3 : 2, (WM * WN == 4 && BK == 32) ? 3 : 2)))
void conv_bf16_igemm(const IgArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int NTHR = 64 * WM * WN;           // one wave per 64x64 result tile
  constexpr int CPR = BK / 8;                 // 16-byte chunks per A row
  constexpr int ROWB = BK * 2 + 16;           // padded LDS row (conflict-free ds_read_b128 across 16 rows)
  constexpr int NA = BM * CPR / NTHR;          // A chunks per thread per stage
  constexpr int A_BYTES = BM * ROWB;
  constexpr int B_BYTES = BTR ? (BN / 32) * (BK * 64) : BN * ROWB;
  constexpr int NB = BTR ? (BK * BN / 8) / NTHR : BN * CPR / NTHR;
  static_assert((WM * WN == 4 || WM * WN == 8) && NA >= 1 && NB >= 1, "tile");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // buffer b: A at b*STAGE_BYTES, B behind it

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int RS = g.R * g.R;
  const int cchunks = g.Cin / BK;
  const int nstage_all = RS * cchunks;
  const int st_begin = blockIdx.z * g.stages_per_split;
  const int st_end = min(nstage_all, st_begin + g.stages_per_split);

  // ---- the A rows this thread stages (fixed for the kernel): CPR consecutive lanes share one row, so a quad of lanes
  // loads one 64-byte line (dealing eight consecutive lanes to eight ROWS instead makes the ds_write_b128 of a stage
  // conflict-free, but every lane quad then touches four lines: forward 545 -> 469 TFLOP/s over the SD layer table,
  // measured on one box, profiles/r05_convbench_bf16_ab.txt)
  constexpr int RPP = NTHR / CPR;             // rows per pass
  const int a_cc = tid % CPR;
  const int a_row0 = tid / CPR;
  int a_hb[NA], a_wb[NA], a_nb[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = a_row0 + RPP * i;
    const int m = m0 + row;
    if (m < g.M) {
      const int n = m / (g.OH * g.OW);
      const int rem = m - n * (g.OH * g.OW);
      const int oh = rem / g.OW, ow = rem - oh * g.OW;
      a_hb[i] = oh * g.stride - g.pad;
      a_wb[i] = ow * g.stride - g.pad;
      a_nb[i] = n * g.H * g.W;
    } else {
      a_hb[i] = -(1 << 20); a_wb[i] = 0; a_nb[i] = 0;  // never in range
    }
  }
  // Staging costs NO vector-ALU work per stage (round 5; it was ~50 instructions per stage and thread — 8 per MFMA —
  // and the kernel's waves spent as many cycles in the vector ALU as in the matrix pipe): both operands come through
  // buffer descriptors, address = descriptor base + per-lane VGPR offset + wave-uniform SGPR offset.  The per-lane part
  // of an A row (pixel of this TAP, or an offset past the end of the buffer when the tap falls outside the image — the
  // hardware then returns zeros: no masks, no branches) is recomputed once per tap; the channel chunk of a stage is the
  // SGPR part.  The per-lane part of a weight row is a kernel constant, tap and channel chunk are the SGPR part.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.x), 0, (int)g.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.wp), 0, (int)g.w_bytes, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;  // past the end of x (launch_igemm refuses x >= 2^31 bytes)
  uint32_t wvo[NB];  // per-thread constant byte offsets of the weight chunks
  if (!BTR) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = a_row0 + RPP * i;
      const int k = min(n0 + row, g.Kout - 1);  // rows past the last channel re-read it: their results are never stored
      wvo[i] = (uint32_t)(k * RS * g.Cin + a_cc * 8) * 2u;
    }
  } else {
    constexpr int CPB = BN / 8;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int id = tid + NTHR * i;
      const int row = id / CPB, cc = id - row * CPB;
      wvo[i] = (uint32_t)(row * RS * g.Cf + min(n0 + cc * 8, g.Kout - 8)) * 2u;
    }
  }
  uint32_t avo[NA];  // byte offsets of this thread's A chunks for the tap being loaded (OOB: outside the image)
  auto aim_tap = [&](int tap) {
    const int r = (g.R == 3) ? (tap * 11) >> 5 : 0, s = tap - r * g.R;  // tap / 3 for tap < 9
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int vh = a_hb[i] + r, vw = a_wb[i] + s;
      bool ok = true;
      if (g.up == 2) { ok = !((vh | vw) & 1); vh >>= 1; vw >>= 1; }
      ok = ok && vh >= 0 && vh < g.H && vw >= 0 && vw < g.W;
      const uint32_t pix = (uint32_t)(a_nb[i] + __mul24(vh, g.W) + vw);        // < 2^24 pixels
      avo[i] = ok ? (__umul24(pix, (uint32_t)g.Cin) + (uint32_t)(a_cc * 8)) * 2u : OOB;
    }
  };
  // wave-uniform state of the LOAD side: next stage to fetch, its tap and first channel
  const int last = st_end - 1;
  int ls = st_begin, ltap = st_begin / cchunks, lc0 = (st_begin - ltap * cchunks) * BK;
  bool lneed = true;
  uint32_t wu_prev = 0;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  auto ld128 = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t voff, uint32_t soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  uint4 r0[NA + NB], r1[NA + NB], r2[NA + NB];  // one register set = the A chunks followed by the B chunks of a stage
  // Loads stage `ls` into `rr`.  A stage index past the end (the loop takes stages three at a time) loads zeros for A
  // (out-of-range offsets) and re-reads the last stage's weights: idle multiplies instead of branches in the loop.
  auto issue = [&](uint4 (&rr)[NA + NB]) {
    const bool dead = ls > last;  // uniform
    if (!dead && lneed) { aim_tap(ltap); lneed = false; }
    const uint32_t sa = (uint32_t)lc0 * 2u;
#pragma unroll
    for (int i = 0; i < NA; ++i) rr[i] = ld128(rx, dead ? OOB : avo[i], dead ? 0u : sa);
    uint32_t wu = wu_prev;
    if (!dead) wu = (uint32_t)(BTR ? (lc0 * RS + (RS - 1 - ltap)) * g.Cf : ltap * g.Cin + lc0) * 2u;
    wu_prev = wu;
#pragma unroll
    for (int i = 0; i < NB; ++i) rr[NA + i] = ld128(rw, wvo[i], wu);
    ++ls;
    if (!dead) {
      lc0 += BK;
      if (lc0 == g.Cin) { lc0 = 0; ++ltap; lneed = true; }
    }
  };
  auto store_stage = [&](int buf, const uint4 (&rr)[NA + NB]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = a_row0 + RPP * i;
      *reinterpret_cast<uint4 *>(lds + buf * STAGE_BYTES + row * ROWB + a_cc * 16) = rr[i];
    }
    if (!BTR) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int row = a_row0 + RPP * i;
        *reinterpret_cast<uint4 *>(lds + buf * STAGE_BYTES + A_BYTES + row * ROWB + a_cc * 16) = rr[NA + i];
      }
    } else {
      constexpr int CPB = BN / 8;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int id = tid + NTHR * i;
        const int row = id / CPB, cc = id - row * CPB;
        *reinterpret_cast<uint4 *>(lds + buf * STAGE_BYTES + A_BYTES + (cc >> 2) * (BK * 64) + row * 64 + (cc & 3) * 16) = rr[NA + i];
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

  const int a_off = (wm * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const int b_off = (wn * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
  // transposing read: 16-lane group gq, lane s: reduction row 8*(gq>>1) + (s>>2), columns 16*(gq&1) + 4*(s&3)
  const int gq = lane >> 4, sl = lane & 15;
  const uint32_t btr_off = (uint32_t)((wn * 2) * (BK * 64) + (8 * (gq >> 1) + (sl >> 2)) * 64 + (16 * (gq & 1) + 4 * (sl & 3)) * 2);

  auto compute = [&](int buf) {
    const char *A = lds + buf * STAGE_BYTES, *B = A + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(A + a_off + kk * 32);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(A + a_off + 32 * ROWB + kk * 32);
      bf16x8 b0, b1;
      if (!BTR) {
        b0 = *reinterpret_cast<const bf16x8 *>(B + b_off + kk * 32);
        b1 = *reinterpret_cast<const bf16x8 *>(B + b_off + 32 * ROWB + kk * 32);
      } else {
        b0 = tr_operand(B, btr_off + kk * (16 * 64), 4 * 64);
        b1 = tr_operand(B, btr_off + BK * 64 + kk * (16 * 64), 4 * 64);
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  // Two-deep register ring over a double-buffered LDS tile: while stage s is multiplied, stage s+1 sits in registers
  // (written to the other LDS buffer after the MFMAs) and the loads of stage s+2 are issued into the set stage s left.
  // Stages are taken three at a time (static register naming, three register sets: the loads of stages s+1 .. s+3 are
  // in flight while stage s is multiplied).
  issue(r0);
  issue(r1);
  issue(r2);
  store_stage(0, r0);
  __syncthreads();
  int buf = 0;
  for (int st = st_begin; st < st_end; st += 3) {
    issue(r0);
    compute(buf);
    store_stage(buf ^ 1, r1);
    __syncthreads();
    issue(r1);
    compute(buf ^ 1);
    store_stage(buf, r2);
    __syncthreads();
    issue(r2);
    compute(buf);
    store_stage(buf ^ 1, r0);
    __syncthreads();
    buf ^= 1;
  }
  if (g.part) {  // split reduction: raw fp32 tile, finished by k_splitk_finish
    float *dst = g.part + (size_t)blockIdx.z * g.M * g.Kout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = n0 + wn * 64 + j * 32 + (lane & 31);
      if (k >= g.Kout) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int m = m0 + wm * 64 + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
          if (m < g.M) dst[(size_t)m * g.Kout + k] = acc[i][j][v];
        }
    }
    return;
  }

  // ---- epilogue: D[row = pixel][col = channel]; row = (v&3) + 8*(v>>2) + 4*(lane>>5), col = lane&31
  const int ohow = g.OH * g.OW;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = n0 + wn * 64 + j * 32 + (lane & 31);
    if (k >= g.Kout) continue;
    const float bk = g.bias ? g.bias[k] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // the epilogue terms of a 32x32 tile are read before its first store, unconditionally (pixels past M re-read the
      // last one): element by element each load waited alone behind the previous store
      const int mb = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
      float nv[16];
      uint16_t av[16];
      if (g.nbias) {
#pragma unroll
        for (int v = 0; v < 16; ++v) nv[v] = g.nbias[(size_t)(min(mb + (v & 3) + 8 * (v >> 2), g.M - 1) / ohow) * g.Kout + k];
      }
      if (g.addend) {
#pragma unroll
        for (int v = 0; v < 16; ++v) av[v] = g.addend[(size_t)min(mb + (v & 3) + 8 * (v >> 2), g.M - 1) * g.Kout + k];
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int m = mb + (v & 3) + 8 * (v >> 2);
        if (m < g.M) {
          float o = acc[i][j][v] + bk;
          if (g.nbias) o += nv[v];
          if (g.addend) o += bf2f(av[v]);
          g.y[(size_t)m * g.Kout + k] = f2bf(o);
        }
      }
    }
  }
}

// out[m][k] = bf16( sum_z part[z][m][k] + bias[k] + nbias[n][k] + addend[m][k] ), 8 channels per thread
__global__ __launch_bounds__(256) void k_splitk_finish(const float *__restrict__ part, int splits, const float *__restrict__ bias,
                                                       const float *__restrict__ nbias, const uint16_t *__restrict__ addend,
                                                       uint16_t *__restrict__ y, int M, int K, int ohow) {
  const int k8n = K >> 3;
  const int64_t total = (int64_t)M * k8n;
  const int64_t mk = (int64_t)M * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / k8n), k = (int)(i - (int64_t)m * k8n) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    for (int z = 0; z < splits; ++z) {
      const float4 a = *reinterpret_cast<const float4 *>(part + z * mk + (int64_t)m * K + k);
      const float4 b = *reinterpret_cast<const float4 *>(part + z * mk + (int64_t)m * K + k + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    if (bias)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += bias[k + j];
    if (nbias)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += nbias[(size_t)(m / ohow) * K + k + j];
    if (addend) {
      const uint4 a = *reinterpret_cast<const uint4 *>(addend + (int64_t)m * K + k);
      const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f((uint16_t)(w[j] & 0xffffu)); v[2 * j + 1] += bf2f((uint16_t)(w[j] >> 16)); }
    }
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2(v[2 * j], v[2 * j + 1]);
    *reinterpret_cast<uint4 *>(y + (int64_t)m * K + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ----------------------------------------------------------------------------------------------- backward-weight
struct WgArgs {
  const uint16_t *x;   // [N][H][W][C] bf16
  const uint16_t *dy;  // [N][OH][OW][K] bf16
  float *part;         // [split][RS][K][C] fp32 partial sums
  float *dw;           // direct: the OIHW gradient itself (one split: no partials, no reduce launch)
  int direct, accumulate;
  int N, H, W, C, OH, OW, K, pad;
  int tiles_h, tiles_w;       // 8x8 output-pixel chunks per image
  int chunks, per_split;      // total chunks, chunks per split
  uint32_t x_bytes, dy_bytes; // extents of x and dy (buffer descriptors: reads past them return zeros)
};

// R: filter size (1 or 3); ST: convolution stride (1 or 2).  Workgroup tile = 64 k x 64 c x all R*R taps; wave (wk, wc)
// owns 32 k x 32 c.  One stage = one 8x8 block of output pixels: dY rows [64 px][64 k] and the input patch
// [(7*ST+R)^2 px][64 c], both as two [px][32 ch] column blocks with 64-byte rows.
//
// Two shapes of the same loop.  The 3x3 stride-1 kernel keeps its 144 accumulators + ONE staging register set inside 256
// registers (233) and 42 KB of LDS, so TWO workgroups share a CU and each hides the other's loads: 2.40 -> 2.05 ms over
// the SD layer table against one workgroup per CU with a ring of three register sets (same box, alternated,
// profiles/r05_convbench_bf16_ab.txt).  The stride-2 kernel stages a 17x17 patch (90 KB double buffered): two
// workgroups do not fit a CU's LDS, the single set then loses 7 - 10 %, so it (and the 1x1 fallback) keeps the ring.
#ifndef SALUN_BF16_WGRAD_EXP
#define SALUN_BF16_WGRAD_EXP 0  // timing experiments only (profiles/r05_wgrad_phases.txt): 1 = no stores, 2 = no reduction
#endif
template <int R, int ST>
__global__ __launch_bounds__(256, (R == 3 && ST == 1) ? 2 : 1) void conv_bf16_wgrad(const WgArgs g) {
  constexpr bool PAIRED = R == 3 && ST == 1;   // two workgroups per CU, one register set
  constexpr int RS = R * R;
  constexpr int PW = 7 * ST + R;               // patch width = height
  constexpr int PPX = PW * PW;
  constexpr int DY_BLK = 64 * 64;              // bytes of one [64 px][32 k] block
  constexpr int X_BLK = PPX * 64;
  constexpr int STAGE = 2 * DY_BLK + 2 * X_BLK;
  constexpr int NDY = 2;                       // 512 chunks / 256 threads
  constexpr int NX = (PPX * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave & 1, wc = wave >> 1;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64, split = blockIdx.z;
  const int cb = split * g.per_split;
#if SALUN_BF16_WGRAD_EXP == 2
  const int ce = cb;  // timing experiment: no reduction at all, the epilogue alone
#else
  const int ce = min(g.chunks, cb + g.per_split);
#endif

  // Staging (round 5).  36 bf16 MFMAs per chunk are ~1,150 cycles — less than a trip to HBM — so ONE wave per SIMD with a
  // single chunk of prefetch left the matrix pipe waiting on loads.  The ring variant keeps three register sets over the
  // double-buffered LDS tile (the loads of chunks s+1 .. s+3 are in flight while chunk s is multiplied); the paired
  // variant has a second workgroup on the CU instead.  Every load goes through a buffer descriptor: a lane whose pixel / channel lies outside the
  // tensor reads at an offset past its end and gets zeros — no branch around a load (hipcc cannot count loads behind a
  // branch and would wait for ALL of them before the first LDS store), no masks.
  const __amdgpu_buffer_rsrc_t rdyb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.dy), 0, (int)g.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(g.x), 0, (int)g.x_bytes, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;  // past the end of either tensor (the launcher refuses >= 2^31 bytes)
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  auto ld128 = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t voff, uint32_t soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  // per-thread constants of the items it stages
  int dy_row[NDY], dy_col[NDY];
  uint32_t dy_vo[NDY];   // byte offset inside the 8x8 tile (OOB: channel past K)
#pragma unroll
  for (int i = 0; i < NDY; ++i) {
    const int id = tid + 256 * i;
    const int px = id >> 3, cc = id & 7;
    dy_row[i] = px >> 3; dy_col[i] = px & 7;
    const int k = k0 + cc * 8;
    dy_vo[i] = (k < g.K) ? (uint32_t)(((px >> 3) * g.OW + (px & 7)) * g.K + k) * 2u : OOB;
  }
  int x_row[NX], x_col[NX], x_vo[NX];  // patch position and byte offset relative to the patch origin; row < -2^20: never valid
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int id = tid + 256 * i;
    const int px = id >> 3, cc = id & 7;
    const int c = c0 + cc * 8;
    const bool live = px < PPX && c < g.C;
    x_row[i] = live ? px / PW : -(1 << 24);
    x_col[i] = px % PW;
    x_vo[i] = ((px / PW) * g.W + (px % PW)) * g.C * 2 + c * 2;
  }
  // wave-uniform state of the LOAD side: next chunk to fetch = (image ln, tile row lth, tile column ltw)
  int lch = cb;
  int ln, lth, ltw;
  {
    const int per_img = g.tiles_h * g.tiles_w;
    ln = cb / per_img;
    const int t = cb - ln * per_img;
    lth = t / g.tiles_w; ltw = t - lth * g.tiles_w;
  }
  uint4 s0[NDY + NX], s1[NDY + NX], s2[NDY + NX];
  auto issue = [&](uint4 (&rr)[NDY + NX]) {
    const bool dead = lch >= ce;  // uniform: nothing left to fetch (the register set is not used afterwards)
    if (!dead) {
      const int oh0 = lth * 8, ow0 = ltw * 8;
      const uint32_t dsoff = (uint32_t)(((ln * g.OH + oh0) * g.OW + ow0) * g.K) * 2u;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        const bool ok = (oh0 + dy_row[i] < g.OH) && (ow0 + dy_col[i] < g.OW);
        rr[i] = ld128(rdyb, ok ? dy_vo[i] : OOB, dsoff);
      }
      const int ih0 = oh0 * ST - g.pad, iw0 = ow0 * ST - g.pad;
      const int xbase = ((ln * g.H + ih0) * g.W + iw0) * g.C * 2;  // may be negative (first rows of the first image)
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int ih = ih0 + x_row[i], iw = iw0 + x_col[i];
        const bool ok = ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
        rr[NDY + i] = ld128(rxb, ok ? (uint32_t)(xbase + x_vo[i]) : OOB, 0u);
      }
      ++lch;
      if (++ltw == g.tiles_w) { ltw = 0; if (++lth == g.tiles_h) { lth = 0; ++ln; } }
    }
  };
  auto store_stage = [&](int buf, const uint4 (&rr)[NDY + NX]) {
    char *base = lds + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int id = tid + 256 * i;
      const int px = id >> 3, cc = id & 7;
      *reinterpret_cast<uint4 *>(base + (cc >> 2) * DY_BLK + px * 64 + (cc & 3) * 16) = rr[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int id = tid + 256 * i;
      const int px = id >> 3, cc = id & 7;
      if (px < PPX) *reinterpret_cast<uint4 *>(base + 2 * DY_BLK + (cc >> 2) * X_BLK + px * 64 + (cc & 3) * 16) = rr[NDY + i];
    }
  };

  f32x16 acc[RS];
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  // transposing reads: group gq, lane s -> reduction pixel 8*(gq>>1) + (s>>2) (+4), channels 16*(gq&1) + 4*(s&3)
  const int gq = lane >> 4, sl = lane & 15;
  const int chan_b = (16 * (gq & 1) + 4 * (sl & 3)) * 2;
  // k-step kk covers output pixels 16*kk .. 16*kk+15 = rows 2*kk, 2*kk+1 of the 8x8 block; this lane's pixel:
  // row 2*kk + (gq>>1), column (s>>2) and (s>>2)+4
  const uint32_t dy_off = (uint32_t)(wk * DY_BLK + (8 * (gq >> 1) + (sl >> 2)) * 64 + chan_b);
  const uint32_t x_off = (uint32_t)(2 * DY_BLK + wc * X_BLK + (((gq >> 1) * ST) * PW + (sl >> 2) * ST) * 64 + chan_b);
  auto compute = [&](int buf) {
    const char *base = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 a = tr_operand(base, dy_off + kk * (16 * 64), 4 * 64);
#pragma unroll
      for (int t = 0; t < RS; ++t) {
        const int r = t / R, s = t % R;
        const bf16x8 b = tr_operand(base, x_off + (kk * 2 * ST * PW + r * PW + s) * 64, 4 * ST * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  };

  if constexpr (PAIRED) {
    // one set: chunk ch + 1 is in flight while chunk ch is multiplied; the other workgroup on the CU covers the rest
    issue(s0);
    if (cb < ce) store_stage(0, s0);
    __syncthreads();
    for (int ch = cb; ch < ce; ++ch) {
      const int b1 = (ch - cb) & 1;
      issue(s0);                                        // chunk ch + 1
      compute(b1);                                      // chunk ch
      if (ch + 1 < ce) store_stage(b1 ^ 1, s0);
      __syncthreads();
    }
  } else {
    // At the top of a trip chunk ch sits in LDS[buf], set s1 holds chunk ch + 1, s2 holds ch + 2 and s0 is free; the
    // trip multiplies three chunks and refills the three sets (static register naming), the LDS buffers alternate.
    issue(s0);
    issue(s1);
    issue(s2);
    if (cb < ce) store_stage(0, s0);
    __syncthreads();
    int buf = 0;
    for (int ch = cb; ch < ce; ch += 3) {
      issue(s0);                                        // chunk ch + 3
      compute(buf);                                     // chunk ch
      if (ch + 1 < ce) store_stage(buf ^ 1, s1);
      __syncthreads();
      if (ch + 1 < ce) {
        issue(s1);                                      // chunk ch + 4
        compute(buf ^ 1);                               // chunk ch + 1
        if (ch + 2 < ce) store_stage(buf, s2);
      }
      __syncthreads();
      if (ch + 2 < ce) {
        issue(s2);                                      // chunk ch + 5
        compute(buf);                                   // chunk ch + 2
        if (ch + 3 < ce) store_stage(buf ^ 1, s0);
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- partials: part[split][tap][k][c]; D row = k, col = c
  const int c = c0 + wc * 32 + (lane & 31);
  const bool kw_live = k0 + wk * 32 < g.K;  // wave-uniform: K is a multiple of 32, so are the wave's 32 rows
  if (g.direct) {
    // one split: this workgroup holds the whole sum of its (k, c) tile — a lane owns all R*R taps of an element, i.e.
    // R*R consecutive floats of the OIHW gradient, and the 32 lanes of a row 32*R*R consecutive floats.  (Sending the
    // tile through LDS to store whole rows was measured in round 5: 8x8 layers 67 -> 64 us, 16x16 layers +1 .. +6 us,
    // the stride-2 kernel 92 -> 115 us, the SD step +1 % — the L2 merges the nine partial stores of a line; not kept.)
    // `+=` (the product path: gradients accumulate into the flat arena) reads the old values of FOUR k rows, 4 * RS
    // loads in flight, before the first store: written as `dst[t] = accumulate ? dst[t] + a : a` the compiler emitted
    // load, s_waitcnt vmcnt(0), store for each of the 144 elements — 144 serial trips to memory, 47 of the 69 us of an
    // 8x8 1280 -> 1280 layer (profiles/r05_wgrad_phases.txt).
    if (c < g.C && kw_live) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float *dst = g.dw + ((size_t)(k0 + wk * 32 + 8 * j + 4 * (lane >> 5)) * g.C + c) * RS;
        const size_t row = (size_t)g.C * RS;
        if (g.accumulate) {
          float old[4][RS];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < RS; ++t) old[q][t] = dst[q * row + t];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < RS; ++t) {
#if SALUN_BF16_WGRAD_EXP == 1
              if (acc[t][4 * j + q] != 12345.678f) continue;
#endif
              dst[q * row + t] = old[q][t] + acc[t][4 * j + q];
            }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < RS; ++t) {
#if SALUN_BF16_WGRAD_EXP == 1
              if (acc[t][4 * j + q] != 12345.678f) continue;
#endif
              dst[q * row + t] = acc[t][4 * j + q];
            }
        }
      }
    }
    return;
  }
  if (c < g.C) {
#pragma unroll
    for (int t = 0; t < RS; ++t) {
      float *dst = g.part + (((size_t)split * RS + t) * g.K) * g.C;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = k0 + wk * 32 + (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5);
#if SALUN_BF16_WGRAD_EXP == 1
        if (acc[t][v] != 12345.678f) continue;
#endif
        if (k < g.K) dst[(size_t)k * g.C + c] = acc[t][v];
      }
    }
  }
}

// part[split][RS][K][C] -> dw[K][C][RS] (OIHW), summed over splits in index order.  Block = 64 consecutive (k, c) pairs:
// thread (q, j) sums taps q, q + 4, q + 8 of pair j (reads coalesced over j, four times the workgroups of a thread per
// pair), and the 64 * RS sums leave through LDS as the 64 * RS CONSECUTIVE floats they are in dw.
__global__ __launch_bounds__(256) void k_wgrad_reduce_bf16(const float *__restrict__ part, float *__restrict__ dw, int K,
                                                           int C, int RS, int splits, int accumulate) {
  __shared__ float sm[64 * 9];
  const int64_t kc = (int64_t)K * C;
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * 64, i = i0 + j;
  if (i < kc)
    for (int t = q; t < RS; t += 4) {
      float s = 0.f;
      for (int sp = 0; sp < splits; ++sp) s += part[((int64_t)sp * RS + t) * kc + i];
      sm[j * RS + t] = s;
    }
  __syncthreads();
  const int64_t o0 = i0 * RS, total = kc * RS;
  float old[3];  // 64 * 9 / 256 -> at most three floats per thread; all old values are read before the first store
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int idx = threadIdx.x + 256 * u;
    old[u] = (accumulate && idx < 64 * RS && o0 + idx < total) ? dw[o0 + idx] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int idx = threadIdx.x + 256 * u;
    if (idx < 64 * RS && o0 + idx < total) dw[o0 + idx] = accumulate ? old[u] + sm[idx] : sm[idx];
  }
}

// per-channel sum over pixels of an NHWC bf16 tensor (the bias gradient): [M][K] -> fp32 [K], two deterministic stages.
// Stage 1: block = 32 channel groups (8 channels each, one 16-byte load) x 8 row lanes over one row chunk.
constexpr int COLSUM_CHUNKS = 128;
// With `cpi` chunks per image (rows_per_image = OH * OW) no chunk straddles two images, and the finish kernel can also give
// the per-image sums dnb[n][k] — the gradient of the per-image channel offset `nbias` of the forward epilogue (a ResBlock's
// time-embedding term), which was a bf16 -> fp32 copy of dy plus a library reduction per ResBlock convolution before round 6.
// Without dnb the launcher passes rows_per_image = M, cpi = chunks: the walk of rounds 2 - 5.
__global__ __launch_bounds__(256) void k_colsum_partial(const uint16_t *__restrict__ dy, float *__restrict__ part, int64_t M,
                                                        int K, int64_t rows_per_chunk, int64_t rows_per_image, int cpi) {
  __shared__ float s[8][32][9];
  const int gx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int k = (blockIdx.x * 32 + gx) * 8;
  const int img = blockIdx.y / cpi;
  const int64_t i1 = (int64_t)(img + 1) * rows_per_image < M ? (int64_t)(img + 1) * rows_per_image : M;
  const int64_t r0 = (int64_t)img * rows_per_image + (int64_t)(blockIdx.y - img * cpi) * rows_per_chunk;
  const int64_t r1 = r0 + rows_per_chunk < i1 ? r0 + rows_per_chunk : i1;
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 0.f;
  if (k < K)
    for (int64_t m = r0 + ry; m < r1; m += 8) {
      const uint4 v = *reinterpret_cast<const uint4 *>(dy + m * K + k);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[2 * j] += bf2f((uint16_t)(w[j] & 0xffffu)); a[2 * j + 1] += bf2f((uint16_t)(w[j] >> 16)); }
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) s[ry][gx][j] = a[j];
  __syncthreads();
  if (ry == 0 && k < K) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = ((s[0][gx][j] + s[1][gx][j]) + (s[2][gx][j] + s[3][gx][j])) + ((s[4][gx][j] + s[5][gx][j]) + (s[6][gx][j] + s[7][gx][j]));
      part[(size_t)blockIdx.y * K + k + j] = t;
    }
  }
}
// out[k] (=, += or not at all) and, with dnb, dnb[n][k] = the sum of image n's chunks (chunks = images * cpi then)
__global__ __launch_bounds__(256) void k_colsum_finish(const float *__restrict__ part, float *__restrict__ out, int K, int chunks,
                                                       int accumulate, float *__restrict__ dnb, int cpi) {
  __shared__ float s[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + cx;
  float t = 0.f;
  if (k < K) {
    if (dnb) {
      for (int n = ry; n * cpi < chunks; n += 8) {
        float u = 0.f;
        for (int c = 0; c < cpi; ++c) u += part[(size_t)(n * cpi + c) * K + k];
        dnb[(size_t)n * K + k] = u;
        t += u;
      }
    } else {
      for (int c = ry; c < chunks; c += 8) t += part[(size_t)c * K + k];
    }
  }
  s[ry][cx] = t;
  __syncthreads();
  if (ry == 0 && k < K && out) {
    t = ((s[0][cx] + s[1][cx]) + (s[2][cx] + s[3][cx])) + ((s[4][cx] + s[5][cx]) + (s[6][cx] + s[7][cx]));
    out[k] = accumulate ? out[k] + t : t;
  }
}

// the two launches of a bias gradient (and, with dnb, of the per-image sums); `cpart`: COLSUM_CHUNKS * K floats
int launch_colsum(const uint16_t *dy, float *cpart, float *db, float *dnb, int64_t M, int K, int images, int accumulate,
                  hipStream_t st) {
  int chunks, cpi;
  int64_t rpc, rpi;
  if (dnb) {
    if (images < 1 || images > COLSUM_CHUNKS || M % images) return SALUN_EINVAL;
    rpi = M / images;
    cpi = COLSUM_CHUNKS / images;
    if ((int64_t)cpi > (rpi + 63) / 64) cpi = (int)((rpi + 63) / 64);
    if (cpi < 1) cpi = 1;
    rpc = (rpi + cpi - 1) / cpi;
    chunks = images * cpi;
  } else {
    chunks = (int)((M + 63) / 64);
    if (chunks > COLSUM_CHUNKS) chunks = COLSUM_CHUNKS;
    rpc = (M + chunks - 1) / chunks;
    rpi = M;
    cpi = chunks;
  }
  hipLaunchKernelGGL(k_colsum_partial, dim3((K / 8 + 31) / 32, chunks), dim3(256), 0, st, dy, cpart, M, K, rpc, rpi, cpi);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colsum_finish, dim3((K + 31) / 32), dim3(256), 0, st, cpart, db, K, chunks, accumulate, dnb, cpi);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

#ifndef SALUN_BF16_WGRAD_TARGET
#define SALUN_BF16_WGRAD_TARGET 384  // lab builds: other split targets (profiles/r05_wgrad_phases.txt, table 2)
#endif
int wgrad_splits(int tiles, int chunks) {
  // about 1.5 workgroups per CU, never more splits than chunks
  // measured on the SD-v1 step (round 3, one box, four builds): 1536 / 768 / 384 / 256 target workgroups ->
  // 219.3 / 209.3 / 205.7 / 206.6 ms; again with the paired 3x3 kernel (round 5, layer table): 512 / 384 / 256 ->
  // 2.39 / 1.99 / 2.08 ms — every split adds a K*C*R*R fp32 partial to write and re-read (59 MB for a
  // 1280x1280 3x3 layer), which costs more than the second resident round of workgroups gives back
  int s = (SALUN_BF16_WGRAD_TARGET + tiles - 1) / tiles;
  if (s > chunks) s = chunks;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}

// (The kernels lean on this domain: aim_tap's `tap / 3` as (tap * 11) >> 5 is exact for tap < 9 only, i.e. R in {1, 3}, and
// the backward-weight epilogue guards whole waves with K % 32 == 0.  Widening it means revisiting both — ADVICE r5.)
bool supported(int C, int K, int R, int stride, int pad) {
  return (R == 1 || R == 3) && (stride == 1 || stride == 2) && pad >= 0 && pad <= R - 1 && C % 32 == 0 && K % 32 == 0 &&
         C >= 32 && K >= 32;
}

// Reduction split of the data kernels: a convolution with few output tiles (the 16x16 and 8x8 levels of the U-Net)
// leaves most CUs without a workgroup, and a lone workgroup pays the full memory latency on every stage; splitting
// the (tap, channel) reduction over blockIdx.z fills the chip (target: 3 workgroups per CU) at the price of one fp32
// round trip of the output tile.
#ifndef SALUN_BF16_SPLIT_TARGET
#define SALUN_BF16_SPLIT_TARGET 768  // lab builds: other targets (profiles/r06_sd_split_targets.txt)
#endif
struct SplitPlan { int splits, per; };
SplitPlan plan_split(int tiles, int nstage) {
  int s = SALUN_BF16_SPLIT_TARGET / (tiles < 1 ? 1 : tiles);
  if (s > nstage / 6) s = nstage / 6;  // at least 6 stages per workgroup
  if (s > 16) s = 16;
  if (s < 1) s = 1;
  const int per = (nstage + s - 1) / s;
  s = (nstage + per - 1) / per;
  return {s, per};
}

template <int WM, int WN, int BK, bool BTR>
int launch_igemm(IgArgs a, void *ws, size_t ws_bytes, hipStream_t st) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  constexpr int ROWB = BK * 2 + 16;
  constexpr int A_BYTES = BM * ROWB;
  constexpr int B_BYTES = BTR ? (BN / 32) * (BK * 64) : BN * ROWB;
  const size_t lds = 2 * (size_t)(A_BYTES + B_BYTES);
  static unsigned long long attr_done = 0;  // one bit per device
  if (salun_once_needed(&attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_igemm<WM, WN, BK, BTR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SALUN_EIO;
    salun_once_mark(&attr_done);
  }
  const int mt = (a.M + BM - 1) / BM, nt = (a.Kout + BN - 1) / BN;
  const int nstage = a.R * a.R * (a.Cin / BK);
  SplitPlan sp = plan_split(mt * nt, nstage);
  if (sp.splits > 1 && (!ws || ws_bytes < (size_t)sp.splits * a.M * a.Kout * sizeof(float))) sp = {1, nstage};
  a.stages_per_split = sp.per;
  a.part = sp.splits > 1 ? static_cast<float *>(ws) : nullptr;
  {  // buffer descriptors: an out-of-image tap reads at byte offset 2^31, which must lie past the end of x
    const size_t xb = (size_t)(a.M / (a.OH * a.OW)) * a.H * a.W * a.Cin * 2;  // N images of H x W x Cin bf16
    const size_t wb = (size_t)a.Kf * a.R * a.R * a.Cf * 2;
    if (xb >= ((size_t)1 << 31) || wb >= ((size_t)1 << 32)) return SALUN_EINVAL;
    a.x_bytes = (uint32_t)xb;
    a.w_bytes = (uint32_t)wb;
  }
  hipLaunchKernelGGL((conv_bf16_igemm<WM, WN, BK, BTR>), dim3(mt, nt, sp.splits), dim3(64 * WM * WN), lds, st, a);
  SALUN_LAUNCH_CHECK();
  if (sp.splits > 1) {
    const int64_t total = (int64_t)a.M * (a.Kout / 8);
    hipLaunchKernelGGL(k_splitk_finish, dim3(salun_grid_for(total, 256)), dim3(256), 0, st, a.part, sp.splits, a.bias, a.nbias,
                       a.addend, a.y, a.M, a.Kout, a.OH * a.OW);
    SALUN_LAUNCH_CHECK();
  }
  return SALUN_OK;
}

template <bool BTR>
int dispatch_igemm(const IgArgs &a, void *ws, size_t ws_bytes, hipStream_t st) {
  const int bk64 = (a.Cin % 64 == 0);
  if (SALUN_BF16_TILE == 1 && bk64) return launch_igemm<2, 2, 64, BTR>(a, ws, ws_bytes, st);
  if (SALUN_BF16_TILE == 2) return launch_igemm<2, 2, 32, BTR>(a, ws, ws_bytes, st);
  if (SALUN_BF16_TILE == 3 && bk64) return launch_igemm<4, 1, 64, BTR>(a, ws, ws_bytes, st);
  if (SALUN_BF16_TILE == 4) return launch_igemm<4, 1, 32, BTR>(a, ws, ws_bytes, st);
  if (SALUN_BF16_TILE == 5) return launch_igemm<4, 2, 32, BTR>(a, ws, ws_bytes, st);
  if (SALUN_BF16_TILE == 6) return launch_igemm<2, 4, 32, BTR>(a, ws, ws_bytes, st);
  // 128 x 256 (eight waves) where the channel count tiles it: a third less operand traffic per MFMA, +5..13 % on the
  // 1280-channel levels; 128 x 128 otherwise (K = 320 / 640 would pad a 256-wide tile by 37 % / 17 %)
  if (a.Kout % 256 == 0) return launch_igemm<2, 4, 32, BTR>(a, ws, ws_bytes, st);
  return launch_igemm<2, 2, 32, BTR>(a, ws, ws_bytes, st);
}

template <int R, int ST>
int launch_wgrad(const WgArgs &a, int splits, hipStream_t st) {
  constexpr int PW = 7 * ST + R;
  const size_t lds = 2 * (size_t)(2 * 64 * 64 + 2 * PW * PW * 64);
  static unsigned long long attr_done = 0;  // one bit per device
  if (salun_once_needed(&attr_done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_bf16_wgrad<R, ST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SALUN_EIO;
    salun_once_mark(&attr_done);
  }
  WgArgs b = a;
  {  // buffer descriptors: an out-of-image lane reads at byte offset 2^31, which must lie past the end of both tensors
    const size_t xb = (size_t)a.N * a.H * a.W * a.C * 2, db = (size_t)a.N * a.OH * a.OW * a.K * 2;
    if (xb >= ((size_t)1 << 31) || db >= ((size_t)1 << 31)) return SALUN_EINVAL;
    b.x_bytes = (uint32_t)xb;
    b.dy_bytes = (uint32_t)db;
  }
  dim3 grid((a.K + 63) / 64, (a.C + 63) / 64, splits);
  hipLaunchKernelGGL((conv_bf16_wgrad<R, ST>), grid, dim3(256), lds, st, b);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT int salun_conv2d_bf16_supported(int C, int K, int R, int stride, int pad) {
  return supported(C, K, R, stride, pad) ? 1 : 0;
}

SALUN_EXPORT int salun_conv2d_bf16_pack_weights(const float *w, uint16_t *wp, int K, int C, int R, salun_stream_t stream) {
  if (!w || !wp || K < 1 || C < 8 || C % 8 || (R != 1 && R != 3)) return SALUN_EINVAL;
  if (!salun_aligned16(wp)) return SALUN_EINVAL;
  const int64_t total = (int64_t)K * R * R * (C / 8);
  hipLaunchKernelGGL(k_pack_w, dim3(salun_grid_for(total, 256)), dim3(256), 0, salun_hip_stream(stream), w, wp, K, C, R * R);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_bf16_pack_weights_batch(const salun_bf16_pack_job_t *jobs, int n, salun_stream_t stream) {
  if (!jobs || n < 1 || n > SALUN_BF16_PACK_MAX_JOBS) return SALUN_EINVAL;
  PackJobsB pj;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const salun_bf16_pack_job_t &q = jobs[i];
    if (!q.w || !q.wp || q.K < 1 || q.C < 8 || (q.R != 1 && q.R != 3) || (q.transposed && q.R != 1)) return SALUN_EINVAL;
    if (!q.transposed && (q.C % 8 || !salun_aligned16(q.wp))) return SALUN_EINVAL;
    pj.j[i] = {q.w, q.wp, q.K, q.C, q.R * q.R, q.transposed ? 1 : 0};
    const int64_t blocks = q.transposed ? (int64_t)((q.C + 31) / 32) * ((q.K + 31) / 32)
                                        : ((int64_t)q.K * (q.C / 8) + 255) / 256;
    if (blocks > most) most = blocks;
  }
  // grid.x: enough workgroups for the largest job to spread over the chip; the jobs loop over what is left
  const unsigned gx = (unsigned)(most < 1 ? 1 : (most > 256 ? 256 : most));
  hipLaunchKernelGGL(k_pack_jobs, dim3(gx, (unsigned)n), dim3(256), 0, salun_hip_stream(stream), pj);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// Scratch for the reduction split of forward / backward-data: 16 fp32 copies of the larger of the two outputs is the
// most any plan uses; a smaller (or null) workspace only disables the split.
SALUN_EXPORT size_t salun_conv2d_bf16_data_workspace_bytes(int N, int H, int W, int C, int K, int R, int stride, int pad) {
  if (!supported(C, K, R, stride, pad) || N < 1 || H < 1 || W < 1) return 0;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  const int BK = 32;
  // tile of the variant dispatch_igemm picks (lab builds pin others)
  const int BMt = (SALUN_BF16_TILE == 3 || SALUN_BF16_TILE == 4 || SALUN_BF16_TILE == 5) ? 256 : 128;
  const int BNt = (SALUN_BF16_TILE == 3 || SALUN_BF16_TILE == 4) ? 64 : (SALUN_BF16_TILE == 6 ? 256 : 128);
  const int BNf = (SALUN_BF16_TILE == 0 && K % 256 == 0) ? 256 : BNt;  // forward: output channels K
  const int BNd = (SALUN_BF16_TILE == 0 && C % 256 == 0) ? 256 : BNt;  // backward-data: output channels C
  size_t need = 0;
  {  // forward
    const int M = N * OH * OW, tiles = ((M + BMt - 1) / BMt) * ((K + BNf - 1) / BNf);
    const SplitPlan sp = plan_split(tiles, R * R * (C / BK));
    if (sp.splits > 1) need = (size_t)sp.splits * M * K * sizeof(float);
  }
  {  // backward-data
    const int M = N * H * W, tiles = ((M + BMt - 1) / BMt) * ((C + BNd - 1) / BNd);
    const SplitPlan sp = plan_split(tiles, R * R * (K / BK));
    const size_t b = sp.splits > 1 ? (size_t)sp.splits * M * C * sizeof(float) : 0;
    if (b > need) need = b;
  }
  return need;
}

extern "C" int salun_conv2d_bf16_forward_ring(const uint16_t *x, const uint16_t *wp, const float *bias, const float *nbias,
                                              const uint16_t *addend, uint16_t *y, int N, int H, int W, int C, int K, int R,
                                              int stride, int pad, hipStream_t st);

SALUN_EXPORT int salun_conv2d_bf16_forward(const uint16_t *x, const uint16_t *wp, const float *bias, const float *nbias,
                                           const uint16_t *addend, uint16_t *y, int N, int H, int W, int C, int K, int R,
                                           int stride, int pad, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!x || !wp || !y || N < 1 || H < 1 || W < 1 || !supported(C, K, R, stride, pad)) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(wp)) return SALUN_EINVAL;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  if (OH < 1 || OW < 1 || (int64_t)N * H * W * C >= (int64_t(1) << 31) || (int64_t)N * OH * OW * K >= (int64_t(1) << 31))
    return SALUN_EINVAL;
  // the staging addresses use 24-bit multiplies: pixel counts and channel counts below 2^24
  if ((int64_t)N * H * W >= (1 << 24) || (int64_t)N * OH * OW >= (1 << 24) || C >= (1 << 24) || K >= (1 << 24)) return SALUN_EINVAL;
  {  // the LDS-DMA ring form (salun_gemm.hip) where the launch has enough tiles to do without a reduction split
    const int rc = salun_conv2d_bf16_forward_ring(x, wp, bias, nbias, addend, y, N, H, W, C, K, R, stride, pad,
                                                  salun_hip_stream(stream));
    if (rc != 0) return rc < 0 ? rc : SALUN_OK;
  }
  IgArgs a{x, wp, bias, nbias, addend, y, N * OH * OW, H, W, C, OH, OW, K, R, stride, pad, 1, K, C, nullptr, 0, 0u, 0u};
  return dispatch_igemm<false>(a, ws, ws_bytes, salun_hip_stream(stream));
}

// dx[N][H][W][C] from dy[N][OH][OW][K]; `addend` (same shape as dx, may be null) rides in the epilogue.
SALUN_EXPORT int salun_conv2d_bf16_backward_data(const uint16_t *dy, const uint16_t *wp, const uint16_t *addend,
                                                 uint16_t *dx, int N, int H, int W, int C, int K, int R, int stride,
                                                 int pad, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!dy || !wp || !dx || N < 1 || H < 1 || W < 1 || !supported(C, K, R, stride, pad)) return SALUN_EINVAL;
  if (!salun_aligned16(dy) || !salun_aligned16(wp)) return SALUN_EINVAL;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  if (OH < 1 || OW < 1 || (int64_t)N * H * W * C >= (int64_t(1) << 31) || (int64_t)N * OH * OW * K >= (int64_t(1) << 31))
    return SALUN_EINVAL;
  // the staging addresses use 24-bit multiplies: pixel counts and channel counts below 2^24
  if ((int64_t)N * H * W >= (1 << 24) || (int64_t)N * OH * OW >= (1 << 24) || C >= (1 << 24) || K >= (1 << 24)) return SALUN_EINVAL;
  // walk dX's pixels; the source is dY (zero-upsampled by `stride`), padding R-1-pad, taps flipped in the weight read
  IgArgs a{dy, wp, nullptr, nullptr, addend, dx, N * H * W, OH, OW, K, H, W, C, R, 1, R - 1 - pad, stride, K, C, nullptr, 0, 0u, 0u};
  return dispatch_igemm<true>(a, ws, ws_bytes, salun_hip_stream(stream));
}

// 1x1 stride-1 unpadded convolutions (and the Linear layers, which arrive as such) are plain dY^T . X products: K16's
// TN kernel (salun_gemm.hip) takes them.  SALUN_WGRAD_TN=0 keeps them on the tap kernel below (A/B measurements).
extern "C" int salun_gemm_bf16_tn_supported(int64_t M, int Na, int Nb);
extern "C" size_t salun_gemm_bf16_tn_workspace_bytes(int64_t M, int Na, int Nb, int variant);
extern "C" int salun_gemm_bf16_tn(const void *dy, const void *x, float *dw, int64_t M, int Na, int Nb, int accumulate, int variant,
                                  void *ws, size_t ws_bytes, salun_stream_t stream);
static bool tn_route(int64_t M, int C, int K, int R, int stride, int pad) {
  static const int on = [] { const char *e = getenv("SALUN_WGRAD_TN"); return e ? atoi(e) : 1; }();
  // short reductions (the 8 x 8 level: M = 512) keep the tap kernel: with one split it adds straight into the gradient,
  // the GEMM would need partials there (profiles/r04_gemmbench_bf16.txt: 11.6 vs 17.5 us, 51 vs 61 us)
  return on && R == 1 && stride == 1 && pad == 0 && M >= 1024 && salun_gemm_bf16_tn_supported(M, K, C);
}

SALUN_EXPORT size_t salun_conv2d_bf16_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int stride, int pad) {
  if (!supported(C, K, R, stride, pad) || N < 1) return 0;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  const int chunks = N * ((OH + 7) / 8) * ((OW + 7) / 8);
  const int tiles = ((K + 63) / 64) * ((C + 63) / 64);
  const size_t tap = (size_t)wgrad_splits(tiles, chunks) * R * R * K * C * sizeof(float) + (size_t)COLSUM_CHUNKS * K * sizeof(float);
  if (tn_route((int64_t)N * H * W, C, K, R, stride, pad)) {
    // never 0 ("unsupported"): the column-sum partials of the bias gradient follow the GEMM's partials.  The larger
    // of the two routes: a gradient pointer that is not 16-byte aligned (an odd-sized parameter ahead of it in the
    // flat arena) takes the tap kernels at launch time, and this query does not see the pointer.
    size_t b = salun_gemm_bf16_tn_workspace_bytes((int64_t)N * H * W, K, C, 0);
    b = (b + 255) & ~(size_t)255;
    b += (size_t)COLSUM_CHUNKS * K * sizeof(float);
    return b > tap ? b : tap;
  }
  return tap;
}

// dw fp32 OIHW [K][C][R][R] (+= when accumulate); db fp32 [K] or null (the bias gradient, += when accumulate); dnb fp32
// [N][K] or null (per-image channel sums of dy: the gradient of the forward's `nbias`, always overwritten)
SALUN_EXPORT int salun_conv2d_bf16_backward_weight_ex(const uint16_t *x, const uint16_t *dy, float *dw, float *db, float *dnb,
                                                      int N, int H, int W, int C, int K, int R, int stride, int pad,
                                                      int accumulate, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!x || !dy || !dw || !ws || N < 1 || !supported(C, K, R, stride, pad)) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(dy)) return SALUN_EINVAL;
  if (dnb && N > COLSUM_CHUNKS) return SALUN_EINVAL;
  const size_t need = salun_conv2d_bf16_wgrad_workspace_bytes(N, H, W, C, K, R, stride, pad);
  if (ws_bytes < need) return SALUN_ENOSPC;
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  if (tn_route((int64_t)N * H * W, C, K, R, stride, pad) && salun_aligned16(dw)) {  // else: the tap kernels take any dw
    const int64_t M = (int64_t)N * H * W;
    size_t gb = salun_gemm_bf16_tn_workspace_bytes(M, K, C, 0);
    const int rc = salun_gemm_bf16_tn(dy, x, dw, M, K, C, accumulate, 0, ws, gb, stream);
    if (rc != SALUN_OK) return rc;
    if (db || dnb) {
      gb = (gb + 255) & ~(size_t)255;
      float *cpart = reinterpret_cast<float *>(static_cast<char *>(ws) + gb);
      return launch_colsum(dy, cpart, db, dnb, M, K, N, accumulate, salun_hip_stream(stream));
    }
    return SALUN_OK;
  }
  WgArgs a;
  a.x = x; a.dy = dy; a.part = static_cast<float *>(ws);
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.K = K; a.pad = pad;
  a.tiles_h = (OH + 7) / 8; a.tiles_w = (OW + 7) / 8;
  a.chunks = N * a.tiles_h * a.tiles_w;
  const int tiles = ((K + 63) / 64) * ((C + 63) / 64);
  const int splits = wgrad_splits(tiles, a.chunks);
  a.per_split = (a.chunks + splits - 1) / splits;
  a.dw = dw; a.direct = splits == 1; a.accumulate = accumulate;
  hipStream_t st = salun_hip_stream(stream);
  int rc;
  if (R == 3 && stride == 1) rc = launch_wgrad<3, 1>(a, splits, st);
  else if (R == 3) rc = launch_wgrad<3, 2>(a, splits, st);
  else if (stride == 1) rc = launch_wgrad<1, 1>(a, splits, st);
  else rc = launch_wgrad<1, 2>(a, splits, st);
  if (rc != SALUN_OK) return rc;
  if (!a.direct) {
    const int64_t kc = (int64_t)K * C;
    hipLaunchKernelGGL(k_wgrad_reduce_bf16, dim3((unsigned)((kc + 63) / 64)), dim3(256), 0, st, a.part, dw, K, C, R * R,
                       splits, accumulate);
    SALUN_LAUNCH_CHECK();
  }
  if (db || dnb) {
    float *cpart = a.part + (size_t)splits * R * R * K * C;
    return launch_colsum(dy, cpart, db, dnb, (int64_t)N * OH * OW, K, N, accumulate, st);
  }
  return SALUN_OK;
}

// The column sums alone: db[K] (=, += or NULL) and / or dnb[images][K] of dy[M][K] (M = images * pixels per image).
SALUN_EXPORT size_t salun_colsum_bf16_workspace_bytes(int K) { return K < 8 ? 0 : (size_t)COLSUM_CHUNKS * K * sizeof(float); }
SALUN_EXPORT int salun_colsum_bf16(const uint16_t *dy, float *db, float *dnb, int64_t M, int K, int images, int accumulate,
                                   void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!dy || (!db && !dnb) || !ws || M < 1 || K < 8 || K % 8 || images < 1 || !salun_aligned16(dy)) return SALUN_EINVAL;
  if (ws_bytes < salun_colsum_bf16_workspace_bytes(K)) return SALUN_ENOSPC;
  return launch_colsum(dy, static_cast<float *>(ws), db, dnb, M, K, images, accumulate, salun_hip_stream(stream));
}

SALUN_EXPORT int salun_conv2d_bf16_backward_weight(const uint16_t *x, const uint16_t *dy, float *dw, float *db, int N,
                                                   int H, int W, int C, int K, int R, int stride, int pad, int accumulate,
                                                   void *ws, size_t ws_bytes, salun_stream_t stream) {
  return salun_conv2d_bf16_backward_weight_ex(x, dy, dw, db, nullptr, N, H, W, C, K, R, stride, pad, accumulate, ws, ws_bytes,
                                              stream);
}
