// salun_conv.hip — K8: fp32 2-D convolution forward / backward-data / backward-weight as implicit GEMM on
// the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains at the 157 TF/s fp32 rate).
//
// Why this exists: profiling the ResNet-18 unlearning step (profiles/r01_bench_miopen_naive_kernel_stats.csv)
// showed the library path choosing `naive_conv_*` / im2col+GEMM kernels for the fp32 NCHW problems of this
// workload on gfx950 (21 ms per weight-gradient launch, > 1 s per step).  The forward/backward of the models
// is MFMA-bound by construction, so the convolutions get their own kernels.
//
// Layouts are the reference's: activations NCHW, weights OIHW read straight from the flat parameter arena.
//
// Forward / backward-data (one kernel, `conv_igemm`): D[k][pix] = sum_{c,r,s} Wt[k][c,r,s] * X[c][pix + (r,s)]
//   * workgroup = 256 threads = 4 waves; tile = 64|128 output pixels x KB (32..128) output channels;
//     A operand = weights (rows = output channels), B operand = input pixels, so a wave's 32 result columns are 32
//     consecutive pixels and the NCHW store is coalesced;
//   * per chunk of CC=8 reduction channels the input *patch* (tile rows + halo, zero padded) and the weight slab
//     are staged in LDS once; the R*S taps are addressed inside the patch (no im2col copy), one ds_read_b32 per
//     operand per MFMA, bank-conflict-free (odd row strides);
//   * backward-data of a stride-1 convolution is the same loop over dY with the weight slab transposed and
//     tap-flipped while it is staged; an optional full-size addend (the residual branch's gradient) rides in the
//     epilogue.
// Backward-data, stride 2 (`conv_dgrad_s2`): all four output parities in one launch — shared dY patch, one
//   accumulator per (parity, channel tile), exact FLOPs (no zero insertion), paired row stores; the per-parity
//   `conv_igemm_tap` launches remain as the fallback for shapes outside its staging assumptions.
// Backward-weight (`conv_wgrad`): dW[k][c][r,s] = sum_pix dY[k][pix] * X[c][pix + (r,s)]: reduction over pixels,
//   9 accumulators (one per tap) per wave, pixel range split over workgroups -> partials -> fixed-order reduce
//   (deterministic, no float atomics); staging interleaved between the MFMAs (see the kernel).  `conv_wgrad_smallc`
//   serves C*R*R <= 32 (the RGB stem) with the (c,r,s) combinations as the MFMA columns.
// Ceilings (tools/micro/mfma_peak.hip): 155.6 TFLOP/s register operands, 144 with this file's LDS operand pattern.
#include "salun_common.h"
#include <mutex>
#include <unordered_set>

// salun_conv_ring.hip: the LDS-DMA ring form of the 3x3 / stride 1 backward-weight (SALUN_EINVAL = not its shape)
int salun_ring_wgrad_launch(const float *x, const float *dy, float *part, int N, int C, int H, int W, int K, int nsplit,
                            int nchunks, hipStream_t st);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CC = 8;  // reduction channels staged per chunk
// conv_igemm's chunk: a 1x1 stride-1 convolution has one MFMA step per two channels, so an 8-channel chunk is four
// steps between two barrier pairs (72 TFLOP/s on the DDPM attention / skip projections); 32 channels make it sixteen.
constexpr int igemm_chunk(int R, int stride) { return (R == 1 && stride == 1) ? 32 : CC; }

// Tuning switches (0 in the product build; tools/_run_wgrad_exp.sh / _run_igemm_exp.sh build A/B libraries with them
// to see where a kernel's time goes — results are WRONG with any bit set; DESIGN.md §6 has the round-3 table):
//   conv_wgrad : 1 no global loads in the loop   2 no LDS stores in the loop   4 no barrier per chunk
//                8 no partial-sum stores
//   conv_igemm : 1 no global loads after the first chunk   2 no LDS stores after the first chunk
//                4 no barriers after the first chunk        8 no output stores
#ifndef SALUN_WGRAD_EXP
#define SALUN_WGRAD_EXP 0
#endif
#ifndef SALUN_IGEMM_OCC
#define SALUN_IGEMM_OCC 2  // waves per SIMD the FAST conv_igemm instantiations are compiled for (A/B: 3)
#endif
#ifndef SALUN_IGEMM_EXP
#define SALUN_IGEMM_EXP 0
#endif
// 1 = 3x3 backward-weight on the eight-wave kernel (two waves per SIMD, conv_wgrad_w) where its geometry applies.
// Measured (round 3, MI355X): ALONE it is 3-7 % faster than conv_wgrad_v (197 vs 204 us on 64x64 @32, 322 vs 347 us on
// the DDPM's 128x128 @32), but INSIDE the step, where backward-weight shares every CU with backward-data of the main
// stream, its second wave per SIMD takes issue slots and registers from that kernel and the step gets slower
// (ResNet-18 112.4 vs 116.3 steps/s, DDPM 8.22 vs 8.31; stream priorities change nothing) — so the product keeps
// the four-wave kernel and this one stays as a build option.
#ifndef SALUN_WGRAD_8WAVES
#define SALUN_WGRAD_8WAVES 0
#endif
// 1 = keep the 3x3 stride-1 backward-weight on conv_wgrad_v (A/B builds against the ring kernel of salun_conv_ring.hip)
#ifndef SALUN_WGRAD_NO_RING
#define SALUN_WGRAD_NO_RING 0
#endif

struct ConvGeomUnused {
  // logical convolution: out[n][k][p][q] = sum x[n][c][p*S - pad + r][q*S - pad + s] * w[k][c][r][s]
  int N, C, H, W, K, P, Q, pad;
  // tile decomposition of the output pixel space (n, p, q), q fastest
  int NI;    // images per tile
  int TP;    // output rows per image per tile
  int IH_t;  // patch rows per image, IW_t patch cols
  int IW_t;
  int PSZ;   // patch floats per channel (NI*IH_t*IW_t), ch_stride = PSZ | 1
  int logQ, logTPQ;
};

// reduction channels per chunk of the rectangular-tap kernel (conv_igemm_tap): 32 / taps, at least 8
constexpr int tap_chunk(int taps) { return taps == 1 ? 32 : taps == 2 ? 16 : 8; }

// One staged patch position of a thread: where it comes from in a channel plane and where it goes in LDS.
struct PatchPos {
  int goff;   // offset inside one (n, c) plane group: n*Cin*HW + ih*W + iw   (channel term added per chunk)
  int loff;   // offset inside one channel's patch
  int valid;  // in bounds (else zero fill)
};

// ---------------------------------------------------------------------------------------------------
// forward / backward-data, stride-1 walk (the fast path: forward of any stride, backward-data of stride 1)
//   R        filter size (1 or 3), square
//   STRIDE   forward: convolution stride; DGRAD: upsampling factor of dY (conv stride is 1)
//   KT       32-channel output tiles per wave
//   WP x WK  wave grid inside the workgroup: WP pixel tiles x WK channel groups (WP*WK == 4)
//   DGRAD    backward-data mode
//   PT       32-pixel tiles per wave (round 3: PT = 2 for the 64-channel layers — a wave then owns 64 pixels x 64
//            channels, four MFMAs per four operand reads instead of two per three, and the weight slab of a chunk is
//            staged once per 256 pixels instead of once per 128)
// `x` is the tensor the patch is read from (forward: input; DGRAD: dY), `y` the tensor written.
// `xC`/`xH`/`xW` are the dims of `x`, `yC`/`yH`/`yW` of `y`.  For DGRAD the virtual input is dY upsampled by
// STRIDE and the padding is R-1-pad.
template <int R, int STRIDE, int KT, int WP, int WK, bool DGRAD, bool FAST, int PT = 1, bool SPLIT = false,
          bool EPI = false>
__global__ __launch_bounds__(256, FAST ? SALUN_IGEMM_OCC : 1) void conv_igemm(const float *__restrict__ x, const float *__restrict__ w,
                                                  const float *__restrict__ bias, float *__restrict__ y, int N,
                                                  int xC, int xH, int xW, int yC, int yH, int yW, int pad, int NI,
                                                  int TP, int IH_t, int IW_t, int logQ, int wC /*w dim1 (C of OIHW)*/,
                                                  int wK /*w dim0*/, const float *__restrict__ nbias /*[N][yC] or null*/,
                                                  const float *__restrict__ addend /*forward: [N][yC][yH][yW] or null*/,
                                                  int csplit /*reduction channels per blockIdx.z (0: no split)*/) {
  constexpr int RS = R * R;
  constexpr int KB = WK * KT * 32;     // output channels per workgroup tile
  constexpr int CC = igemm_chunk(R, STRIDE);  // (shadows the file-wide 8)
  constexpr int WROW = CC * RS + 1;    // LDS weight row (odd => conflict-free across 32 rows)
  constexpr int CONV_S = DGRAD ? 1 : STRIDE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  float *patch = lds;                   // [CC][ch_stride]
  float *wl = lds + CC * ch_stride;     // [KB][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int wp = wave % WP, wk = wave / WP;
  const int Q = yW, P = yH;
  const int tiles_per_img = (P * Q) / (TP * Q);  // row bands per image (1 when NI >= 1 image)
  const int tile = blockIdx.x;
  const int k0 = blockIdx.y * KB;
  // tile -> first image / first output row
  int n0, p0;
  if (NI > 1) { n0 = tile * NI; p0 = 0; }
  else { n0 = tile / tiles_per_img; p0 = (tile - n0 * tiles_per_img) * TP; }

  // ---- this lane's output pixels inside the tile (fixed for the whole kernel), one per pixel tile of the wave
  int q_l[PT], ni_l[PT], p_l[PT], pix_off[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int mloc = (wp * PT + pt) * 32 + lo;  // 0 .. WP*PT*32-1: pixel inside the workgroup tile
    q_l[pt] = mloc & (Q - 1);
    const int pr = mloc >> logQ;               // row index inside the tile (over NI*TP rows)
    ni_l[pt] = pr / TP;
    p_l[pt] = pr - ni_l[pt] * TP;
    pix_off[pt] = (ni_l[pt] * IH_t + p_l[pt] * CONV_S) * IW_t + q_l[pt] * CONV_S;  // tap (0,0) position in the patch
  }

  // ---- the (<= 3) patch positions this thread stages for every channel of a chunk
  constexpr int MAXPOS = (R == 1 && STRIDE == 1) ? 1 : 3;  // 1x1 stride 1: the patch IS the pixel tile (<= 256)
  PatchPos pos[MAXPOS];
  const int planeHW = xH * xW;
  // virtual (possibly upsampled) input extent
  const int vH = DGRAD ? (xH - 1) * STRIDE + 1 : xH;
  const int vW = DGRAD ? (xW - 1) * STRIDE + 1 : xW;
  const int vpad = DGRAD ? (R - 1 - pad) : pad;
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int e = tid + j * 256;
    pos[j].loff = e;
    pos[j].valid = 0;
    pos[j].goff = 0;
    if (e < PSZ) {
      const int ni = e / (IH_t * IW_t);
      const int rem = e - ni * (IH_t * IW_t);
      const int ih = rem / IW_t, iw = rem - ih * IW_t;
      const int n = n0 + ni;
      const int vh = p0 * CONV_S - vpad + ih, vw = -vpad + iw;
      bool ok = (n < N) && vh >= 0 && vh < vH && vw >= 0 && vw < vW;
      int sh = vh, sw = vw;
      if (DGRAD && STRIDE > 1) {
        ok = ok && (vh % STRIDE == 0) && (vw % STRIDE == 0);
        sh = vh / STRIDE;
        sw = vw / STRIDE;
      }
      pos[j].valid = ok;
      pos[j].goff = ok ? (n * xC * planeHW + sh * xW + sw) : 0;
    }
  }

  f32x16 acc[PT][KT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[pt][t][v] = 0.f;

  // reduction channels = channels of the tensor the patch is read from; with a reduction split (under-filled launches:
  // launch_igemm) workgroup z covers [z * csplit, (z+1) * csplit) and writes a partial output image (no epilogue terms)
  // (SPLIT is a template parameter: the extra address arithmetic cost the unsplit instantiations 1.5 % of the ResNet-18
  // step when it was a run-time branch — measured on the same box, tools/_run_ab_tree.sh)
  const int cz0 = SPLIT ? (int)blockIdx.z * csplit : 0;
  const int Cred = SPLIT ? ((cz0 + csplit < xC) ? cz0 + csplit : xC) : xC;
  if (SPLIT) y += (size_t)blockIdx.z * ((size_t)N * yC * yH * yW);
  // Register-staged software pipeline: the global loads of chunk i+1 are issued before the MFMA section of
  // chunk i and only consumed (written to LDS) after it, so their latency hides under the matrix work.
  constexpr int WN = KB * CC * RS / 256;  // weight-slab elements per thread per chunk (KB*RS/32)
  // FAST staging (reduction channels a multiple of CC, channel tile inside the tensor, 16-B aligned weights):
  // the slab is fetched as float4 and no per-element bounds test remains in the hot loop.
  constexpr int ROWF4 = DGRAD ? (KB * RS / 4) : (CC * RS / 4);  // float4 per contiguous global run
  constexpr int NF4 = DGRAD ? (CC * ROWF4) : (KB * ROWF4);      // float4 per chunk
  constexpr int WN4 = (NF4 + 255) / 256;
  float preg[MAXPOS][CC];
  float wreg[FAST ? 1 : WN];
  float4 wreg4[FAST ? WN4 : 1];

  auto load_chunk = [&](int c0) {
    if (FAST) {
#pragma unroll
      for (int j = 0; j < MAXPOS; ++j) {
        if (pos[j].valid) {
          const float *src = x + pos[j].goff + c0 * planeHW;
#pragma unroll
          for (int c = 0; c < CC; ++c) preg[j][c] = src[c * planeHW];
        } else {
#pragma unroll
          for (int c = 0; c < CC; ++c) preg[j][c] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < WN4; ++i) {
        const int e4 = tid + i * 256;
        if (NF4 % 256 == 0 || e4 < NF4) {
          const int row = e4 / ROWF4, q4 = e4 - row * ROWF4;
          // forward: a ragged last channel block re-reads the last real row (its results are masked in the epilogue)
          const int krow = (k0 + row < wK) ? k0 + row : wK - 1;
          const float *src = DGRAD ? (w + (size_t)(c0 + row) * wC * RS + (size_t)k0 * RS + 4 * q4)
                                   : (w + (size_t)krow * wC * RS + (size_t)c0 * RS + 4 * q4);
          wreg4[i] = *reinterpret_cast<const float4 *>(src);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < MAXPOS; ++j) {
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        float v = 0.f;
        if (pos[j].valid && (c0 + c) < Cred) v = x[pos[j].goff + (c0 + c) * planeHW];
        preg[j][c] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int e = tid + i * 256;
      float v = 0.f;
      if (!DGRAD) {
        // forward: rows = output channel k; global w[k][c0..c0+CC][rs] is one contiguous run per k
        const int kk = e / (CC * RS), j = e - kk * (CC * RS);
        const int c = j / RS;
        if ((k0 + kk) < wK && (c0 + c) < wC) v = w[(size_t)(k0 + kk) * wC * RS + (size_t)c0 * RS + j];
      } else {
        // backward-data: rows = forward input channel (output of this pass), reduction over forward k;
        // global w[k][c][rs] is contiguous over (c, rs) for a fixed k
        const int kk = e / (KB * RS), rem = e - kk * (KB * RS);
        const int cl = rem / RS, rs = rem - cl * RS;
        if ((c0 + kk) < wK && (k0 + cl) < wC) v = w[(size_t)(c0 + kk) * wC * RS + (size_t)(k0 + cl) * RS + rs];
      }
      wreg[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < MAXPOS; ++j)
      if (pos[j].loff < PSZ) {
#pragma unroll
        for (int c = 0; c < CC; ++c) patch[c * ch_stride + pos[j].loff] = preg[j][c];
      }
    if (FAST) {
#pragma unroll
      for (int i = 0; i < WN4; ++i) {
        const int e4 = tid + i * 256;
        if (NF4 % 256 == 0 || e4 < NF4) {
          const int row = e4 / ROWF4, q4 = e4 - row * ROWF4;
          const float v4[4] = {wreg4[i].x, wreg4[i].y, wreg4[i].z, wreg4[i].w};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (!DGRAD) {
              wl[row * WROW + 4 * q4 + u] = v4[u];
            } else {
              const int idx = 4 * q4 + u;              // position in the (cl, rs) run of reduction row `row`
              const int cl = idx / RS, rs = idx - cl * RS;
              wl[cl * WROW + row * RS + (RS - 1 - rs)] = v4[u];  // taps flipped while staging
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int e = tid + i * 256;
      if (!DGRAD) {
        const int kk = e / (CC * RS), j = e - kk * (CC * RS);
        wl[kk * WROW + j] = wreg[i];
      } else {
        const int kk = e / (KB * RS), rem = e - kk * (KB * RS);
        const int cl = rem / RS, rs = rem - cl * RS;
        wl[cl * WROW + kk * RS + (RS - 1 - rs)] = wreg[i];  // taps flipped while staging
      }
    }
  };

  load_chunk(cz0);
  for (int c0 = cz0; c0 < Cred; c0 += CC) {
    if (!(SALUN_IGEMM_EXP & 4) || c0 == cz0) __syncthreads();  // previous chunk fully consumed
    if (!(SALUN_IGEMM_EXP & 2) || c0 == cz0) store_chunk();
    if (!(SALUN_IGEMM_EXP & 4) || c0 == cz0) __syncthreads();
    if (c0 + CC < Cred && !(SALUN_IGEMM_EXP & 1)) load_chunk(c0 + CC);  // in flight during the MFMA section below
    // ---- MFMA over the chunk: 2 reduction channels per instruction (lanes 0-31: cc, lanes 32-63: cc+1).
    // The (1 + KT) LDS operands of k-step i+1 are read while the KT MFMAs of k-step i run (one-step-ahead
    // software pipeline, pinned with scheduling barriers): no MFMA waits for an LDS round trip.
    constexpr int NSTEP = (CC / 2) * RS;
    const float *pb0 = patch + hi * ch_stride;
    const float *wb0 = wl + (wk * KT * 32 + lo) * WROW + hi * RS;
    auto operands = [&](int step, float (&bv)[PT], float (&av)[KT]) {
      const int cc = 2 * (step / RS), rs = step % RS, r = rs / R, s2 = rs % R;  // compile-time after unrolling
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) bv[pt] = pb0[pix_off[pt] + cc * ch_stride + r * IW_t + s2];
#pragma unroll
      for (int t = 0; t < KT; ++t) av[t] = wb0[t * 32 * WROW + cc * RS + rs];
    };
    float b_cur[PT], a_cur[KT], b_nxt[PT], a_nxt[KT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) b_nxt[pt] = 0.f;
    operands(0, b_cur, a_cur);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) operands(step + 1, b_nxt, a_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int t = 0; t < KT; ++t)
          acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b_cur[pt], acc[pt][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) b_cur[pt] = b_nxt[pt];
#pragma unroll
      for (int t = 0; t < KT; ++t) a_cur[t] = a_nxt[t];
    }
  }

  // ---- epilogue: D[row = channel][col = pixel]; row = (v&3) + 8*(v>>2) + 4*hi
  if ((SALUN_IGEMM_EXP & 8) && N > 0) return;
  // The epilogue terms of a 32-channel tile are ALL read before its first store (round 5).  Written element by element
  // (`o += bias[k]; o += addend[oi]; y[oi] = o`) every load sat behind the previous element's store — y may alias the
  // addend, so the compiler keeps the order — and waited for alone: up to 3 x 64 serial trips to memory per lane at
  // the end of the two busiest kernels of the DDPM step.
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int n_out = n0 + ni_l[pt], p_out = p0 + p_l[pt];
    if (n_out < N) {
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int kbase = k0 + (wk * KT + t) * 32 + 4 * hi;
        const size_t obase = (((size_t)n_out * yC + kbase) * P + p_out) * Q + q_l[pt];
        const size_t kstride = (size_t)P * Q;
        // loads are unconditional (a row past the last channel re-reads the last one): a load behind a per-lane test
        // becomes a branch with its own wait
        float bv[16], nv[16], av[16];
        const int klast = yC - 1 - kbase;  // < 0: the whole tile is past the last channel (nothing is stored)
        if (bias) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int kr = max(min((v & 3) + 8 * (v >> 2), klast), -kbase);
            bv[v] = DGRAD ? bias[obase + (long long)kr * (long long)kstride] : bias[kbase + kr];  // backward-data: `bias` is a full-size addend (may alias y)
          }
        }
        if (EPI && nbias) {
#pragma unroll
          for (int v = 0; v < 16; ++v) nv[v] = nbias[(size_t)n_out * yC + kbase + max(min((v & 3) + 8 * (v >> 2), klast), -kbase)];
        }
        if (EPI && addend) {
#pragma unroll
          for (int v = 0; v < 16; ++v)
            av[v] = addend[obase + (long long)max(min((v & 3) + 8 * (v >> 2), klast), -kbase) * (long long)kstride];
        }
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int kr = (v & 3) + 8 * (v >> 2);
          if (kbase + kr < yC) {
            float o = acc[pt][t][v];
            if (bias) o += bv[v];
            if (EPI) {
              // forward, EPI instantiations only (a template parameter: as run-time branches these two tests cost the
              // plain instantiations 0.5 % of the ResNet-18 step): the per-image channel bias (the time/class embedding
              // projection of a diffusion ResnetBlock), then a full-size addend (the block's skip branch) — the order
              // the reference's separate adds produce (DDPM/models/diffusion.py:113-127)
              if (nbias) o += nv[v];
              if (addend) o += av[v];
            }
            y[obase + kr * kstride] = o;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// tap-mapped variant (used for the parity classes of stride-2 backward-data)
//   RH x RW  taps staged per channel (forward: the full R x R filter; backward-data with stride 2: the 1..2 x 1..2
//            taps that reach one output parity class)
//   STRIDE   forward convolution stride (backward-data always walks its source with stride 1)
//   KT       32-channel output tiles per wave
//   WP x WK  wave grid inside the workgroup: WP pixel tiles x WK channel groups (WP*WK == 4)
//   DGRAD    backward-data mode: source = dY, weight slab transposed, taps taken as r_i = rtop - ts*i
// Backward-data of a stride-2 convolution is decomposed by output parity (ph, pw): dx[2a+ph][2b+pw] only receives
// taps r = (ph + pad) mod 2 (+2), so each class is a small dense stride-1 convolution over dY written to a
// strided sub-grid of dx — same FLOPs as the forward pass, no zero-insertion.
struct IgemmArgs {
  const float *x;     // tensor the patch is read from (forward: input; DGRAD: dY)
  const float *w;     // OIHW weights [wK][wC][Rfull][Rfull]
  const float *bias;  // optional, per output channel
  float *y;           // tensor written (forward: output; DGRAD: dX)
  int N, xC, xH, xW, yC, yH, yW;
  int subH, subW;          // output sub-grid walked by the tiles (yH/os, yW/os)
  int os, ph, pw;          // output position = sub * os + (ph, pw)
  int vpad_h, vpad_w;      // patch origin: source row = p0*conv_stride - vpad_h + ih
  int NI, TP, IH_t, IW_t, logQ;
  int wC, wK, Rfull;
  int rtop_h, rtop_w, ts;  // DGRAD tap mapping
};

template <int RH, int RW, int STRIDE, int KT, int WP, int WK, bool DGRAD, int MAXPOS>
__global__ __launch_bounds__(256) void conv_igemm_tap(const IgemmArgs g) {
  constexpr int RS = RH * RW;
  constexpr int CC = tap_chunk(RS);    // reduction channels per chunk: 32 / 16 / 8 for 1 / 2 / 4 taps, so every chunk
                                       // carries 16 k-steps between its barriers whatever the parity class
  constexpr int KB = WK * KT * 32;     // output channels per workgroup tile
  constexpr int WROW = CC * RS + 1;    // LDS weight row (odd => conflict-free across 32 rows)
  constexpr int CONV_S = DGRAD ? 1 : STRIDE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int NI = g.NI, TP = g.TP, IH_t = g.IH_t, IW_t = g.IW_t;
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  float *patch = lds;                   // [CC][ch_stride]
  float *wl = lds + CC * ch_stride;     // [KB][WROW]
  const float *__restrict__ x = g.x;
  const float *__restrict__ w = g.w;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int wp = wave % WP, wk = wave / WP;
  const int Q = g.subW, P = g.subH;
  const int tiles_per_img = P / TP;  // row bands per image (1 when a tile spans whole images)
  const int tile = blockIdx.x;
  const int k0 = blockIdx.y * KB;
  int n0, p0;
  if (NI > 1) { n0 = tile * NI; p0 = 0; }
  else { n0 = tile / tiles_per_img; p0 = (tile - n0 * tiles_per_img) * TP; }

  // ---- this lane's output pixel inside the tile (fixed for the whole kernel)
  const int mloc = wp * 32 + lo;
  const int q_l = mloc & (Q - 1);
  const int pr = mloc >> g.logQ;               // row index inside the tile (over NI*TP rows)
  const int ni_l = pr / TP, p_l = pr - ni_l * TP;
  const int pix_off = (ni_l * IH_t + p_l * CONV_S) * IW_t + q_l * CONV_S;  // tap (0,0) position in the patch

  // ---- the (<= MAXPOS) patch positions this thread stages for every channel of a chunk
  PatchPos pos[MAXPOS];
  const int xC = g.xC, planeHW = g.xH * g.xW;
#pragma unroll
  for (int j = 0; j < MAXPOS; ++j) {
    const int e = tid + j * 256;
    pos[j].loff = e;
    pos[j].valid = 0;
    pos[j].goff = 0;
    if (e < PSZ) {
      const int ni = e / (IH_t * IW_t);
      const int rem = e - ni * (IH_t * IW_t);
      const int ih = rem / IW_t, iw = rem - ih * IW_t;
      const int n = n0 + ni;
      const int vh = p0 * CONV_S - g.vpad_h + ih, vw = -g.vpad_w + iw;
      const bool ok = (n < g.N) && vh >= 0 && vh < g.xH && vw >= 0 && vw < g.xW;
      pos[j].valid = ok;
      pos[j].goff = ok ? (n * xC * planeHW + vh * g.xW + vw) : 0;
    }
  }

  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  const int Cred = xC;  // reduction channels = channels of the tensor the patch is read from
  const int wC = g.wC, wK = g.wK, RF2 = g.Rfull * g.Rfull;
  // Register-staged software pipeline: the global loads of chunk i+1 are issued before the MFMA section of
  // chunk i and only consumed (written to LDS) after it, so their latency hides under the matrix work.
  constexpr int WN = KB * CC * RS / 256;  // weight-slab elements per thread per chunk
  float preg[MAXPOS][CC];
  float wreg[WN];

  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int j = 0; j < MAXPOS; ++j) {
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        float v = 0.f;
        if (pos[j].valid && (c0 + c) < Cred) v = x[pos[j].goff + (c0 + c) * planeHW];
        preg[j][c] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int e = tid + i * 256;
      bool ok;
      int off;
      if (!DGRAD) {
        // forward: rows = output channel k; global w[k][c0..c0+CC][rs] is one contiguous run per k
        const int kk = e / (CC * RS), j = e - kk * (CC * RS);
        const int c = j / RS;
        ok = (k0 + kk) < wK && (c0 + c) < wC;
        off = (k0 + kk) * wC * RS + c0 * RS + j;
      } else {
        // backward-data: rows = forward input channel (output of this pass), reduction over forward k
        const int kk = e / (KB * RS), rem = e - kk * (KB * RS);
        const int cl = rem / RS, t = rem - cl * RS;
        const int ti = t / RW, tj = t - ti * RW;
        ok = (c0 + kk) < wK && (k0 + cl) < wC;
        off = ((c0 + kk) * wC + (k0 + cl)) * RF2 + (g.rtop_h - g.ts * ti) * g.Rfull + (g.rtop_w - g.ts * tj);
      }
      float v = 0.f;
      if (ok) v = w[off];
      wreg[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < MAXPOS; ++j)
      if (pos[j].loff < PSZ) {
#pragma unroll
        for (int c = 0; c < CC; ++c) patch[c * ch_stride + pos[j].loff] = preg[j][c];
      }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
      const int e = tid + i * 256;
      if (!DGRAD) {
        const int kk = e / (CC * RS), j = e - kk * (CC * RS);
        wl[kk * WROW + j] = wreg[i];
      } else {
        const int kk = e / (KB * RS), rem = e - kk * (KB * RS);
        const int cl = rem / RS, t = rem - cl * RS;
        wl[cl * WROW + kk * RS + t] = wreg[i];
      }
    }
  };

  load_chunk(0);
  for (int c0 = 0; c0 < Cred; c0 += CC) {
    __syncthreads();  // previous chunk fully consumed
    store_chunk();
    __syncthreads();
    if (c0 + CC < Cred) load_chunk(c0 + CC);  // in flight during the MFMA section below
    // ---- MFMA over the chunk: 2 reduction channels per instruction (lanes 0-31: cc, lanes 32-63: cc+1); the
    // (1 + KT) LDS operands of k-step i+1 are read while the KT MFMAs of k-step i run (as in conv_igemm)
    constexpr int NSTEP = (CC / 2) * RS;
    const float *pb0 = patch + hi * ch_stride + pix_off;
    const float *wb0 = wl + (wk * KT * 32 + lo) * WROW + hi * RS;
    auto operands = [&](int step, float &bv, float (&av)[KT]) {
      const int cc = 2 * (step / RS), rs = step % RS, r = rs / RW, s2 = rs % RW;  // compile-time after unrolling
      bv = pb0[cc * ch_stride + r * IW_t + s2];
#pragma unroll
      for (int t = 0; t < KT; ++t) av[t] = wb0[t * 32 * WROW + cc * RS + rs];
    };
    float b_cur, a_cur[KT], b_nxt = 0.f, a_nxt[KT];
    operands(0, b_cur, a_cur);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) operands(step + 1, b_nxt, a_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b_cur, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      b_cur = b_nxt;
#pragma unroll
      for (int t = 0; t < KT; ++t) a_cur[t] = a_nxt[t];
    }
  }

  // ---- epilogue: D[row = channel][col = pixel]; row = (v&3) + 8*(v>>2) + 4*hi
  const int n_out = n0 + ni_l;
  const int h_out = (p0 + p_l) * g.os + g.ph, w_out = q_l * g.os + g.pw;
  if (n_out < g.N) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int kbase = k0 + (wk * KT + t) * 32 + 4 * hi;
      const size_t obase = (((size_t)n_out * g.yC + kbase) * g.yH + h_out) * g.yW + w_out;
      const size_t kstride = (size_t)g.yH * g.yW;
      float bv[16];  // all epilogue terms of the tile are read before its first store, unconditionally (see conv_igemm)
      const int klast = g.yC - 1 - kbase;
      if (g.bias) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int kr = max(min((v & 3) + 8 * (v >> 2), klast), -kbase);
          bv[v] = DGRAD ? g.bias[obase + (long long)kr * (long long)kstride] : g.bias[kbase + kr];
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int kr = (v & 3) + 8 * (v >> 2);
        if (kbase + kr < g.yC) {
          float o = acc[t][v];
          if (g.bias) o += bv[v];
          g.y[obase + kr * kstride] = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward-data of a STRIDE-2 convolution, all four output parities in one launch.
//   dx[2a+ph][2b+pw] = sum_k sum_{r = (ph+PAD) mod 2 (+2), s likewise} w[k][c][r][s] * dy[k][a + d(r)][b + d(s)],
//   d(r) = (ph + PAD - r) / 2  in {-1, 0, 1}
// A workgroup owns a tile of sub-grid pixels (a, b) and 32*KT*WK channels; every wave keeps 4*KT accumulators (one
// per parity class and channel tile).  The dY patch (tile rows + one halo row/column) and the full R*R weight slab of
// a reduction chunk are staged once and serve all classes — same MFMA count as the forward pass, no zero
// insertion, one launch instead of four, and the epilogue writes (2b, 2b+1) pairs: full coalesced rows of dX.
// (The per-class conv_igemm_tap path stays as the fallback for shapes outside this kernel's staging assumptions.)
template <int R, int PAD, int KT, int WP, int WK>
__global__ __launch_bounds__(256, 2) void conv_dgrad_s2(const float *__restrict__ dy, const float *__restrict__ w,
                                                     const float *__restrict__ addend, float *__restrict__ dx, int N,
                                                     int K, int P, int Q, int C, int H, int W, int NI, int TP,
                                                     int IH_t, int IW_t, int logQ) {
  constexpr int RS = R * R;
  constexpr int KB = WK * KT * 32;
  constexpr int WROW = CC * RS + 1;
  constexpr int DMIN = (R == 1) ? 0 : (PAD == 1 ? 0 : -1);  // smallest row/column offset d(r) that occurs
  constexpr int FP = (R == 1) ? 1 : 2;                      // patch footprint per dimension
  constexpr int NCLS = (R == 1) ? 1 : 4;                    // parity classes that receive any tap
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  float *patch = lds;                // [CC][ch_stride]
  float *wl = lds + CC * ch_stride;  // [KB][WROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int wp = wave % WP, wk = wave / WP;
  const int tiles_per_img = (NI > 1) ? 1 : P / TP;
  const int tile = blockIdx.x;
  const int c0out = blockIdx.y * KB;  // first dX channel of this workgroup
  int n0, p0;
  if (NI > 1) { n0 = tile * NI; p0 = 0; }
  else { n0 = tile / tiles_per_img; p0 = (tile - n0 * tiles_per_img) * TP; }

  const int mloc = wp * 32 + lo;
  const int q_l = mloc & (Q - 1);
  const int pr = mloc >> logQ;
  const int ni_l = pr / TP, p_l = pr - ni_l * TP;
  const int pix_off = (ni_l * IH_t + p_l) * IW_t + q_l;  // patch position of offset (DMIN, DMIN)

  // the one patch position this thread stages per reduction channel (PSZ <= 256)
  const int planePQ = P * Q;
  int goff = 0;
  bool valid = false;
  if (tid < PSZ) {
    const int ni = tid / (IH_t * IW_t);
    const int rem = tid - ni * (IH_t * IW_t);
    const int ih = rem / IW_t, iw = rem - ih * IW_t;
    const int n = n0 + ni;
    const int pp = p0 + DMIN + ih, qq = DMIN + iw;
    valid = (n < N) && pp >= 0 && pp < P && qq >= 0 && qq < Q;
    goff = valid ? (n * K * planePQ + pp * Q + qq) : 0;
  }

  f32x16 acc[NCLS][KT];
#pragma unroll
  for (int cl = 0; cl < NCLS; ++cl)
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[cl][t][v] = 0.f;

  constexpr int ROWF4 = KB * RS / 4;  // float4 per contiguous run of one reduction row: (c in block, r, s)
  constexpr int NF4 = CC * ROWF4;
  constexpr int WN4 = (NF4 + 255) / 256;
  float preg[CC];
  float4 wreg4[WN4];
  auto load_chunk = [&](int k0) {
    if (valid) {
      const float *src = dy + goff + k0 * planePQ;
#pragma unroll
      for (int c = 0; c < CC; ++c) preg[c] = src[c * planePQ];
    } else {
#pragma unroll
      for (int c = 0; c < CC; ++c) preg[c] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < WN4; ++i) {
      const int e4 = tid + i * 256;
      if (NF4 % 256 == 0 || e4 < NF4) {
        const int row = e4 / ROWF4, q4 = e4 - row * ROWF4;
        wreg4[i] = *reinterpret_cast<const float4 *>(w + (size_t)(k0 + row) * C * RS + (size_t)c0out * RS + 4 * q4);
      }
    }
  };
  auto store_chunk = [&]() {
    if (tid < PSZ) {
#pragma unroll
      for (int c = 0; c < CC; ++c) patch[c * ch_stride + tid] = preg[c];
    }
#pragma unroll
    for (int i = 0; i < WN4; ++i) {
      const int e4 = tid + i * 256;
      if (NF4 % 256 == 0 || e4 < NF4) {
        const int row = e4 / ROWF4, q4 = e4 - row * ROWF4;
        const float v4[4] = {wreg4[i].x, wreg4[i].y, wreg4[i].z, wreg4[i].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = 4 * q4 + u;  // position in the (c, rs) run of reduction row `row`
          const int cl = idx / RS, rs = idx - cl * RS;
          wl[cl * WROW + row * RS + rs] = v4[u];
        }
      }
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < K; k0 += CC) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (k0 + CC < K) load_chunk(k0 + CC);
    constexpr int NSTEP = CC / 2;
    const float *pb0 = patch + hi * ch_stride + pix_off;
    const float *wb0 = wl + (wk * KT * 32 + lo) * WROW + hi * RS;
    // operands of reduction pair `step`: the FP*FP patch values around this lane's pixel and the R*R*KT weights
    auto operands = [&](int step, float (&bv)[FP * FP], float (&av)[KT][RS]) {
      const int cc = 2 * step;
#pragma unroll
      for (int di = 0; di < FP; ++di)
#pragma unroll
        for (int dj = 0; dj < FP; ++dj) bv[di * FP + dj] = pb0[cc * ch_stride + di * IW_t + dj];
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) av[t][rs] = wb0[t * 32 * WROW + cc * RS + rs];
    };
    float b_cur[FP * FP], a_cur[KT][RS], b_nxt[FP * FP], a_nxt[KT][RS];
    operands(0, b_cur, a_cur);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) operands(step + 1, b_nxt, a_nxt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int s2 = 0; s2 < R; ++s2) {
          // tap (r, s2) feeds parity class (ph, pw) from the patch value at offsets (d(r), d(s2))
          const int ph = (R == 1) ? 0 : ((r - PAD) & 1), pw = (R == 1) ? 0 : ((s2 - PAD) & 1);
          const int di = (R == 1) ? 0 : ((ph + PAD - r) / 2 - DMIN), dj = (R == 1) ? 0 : ((pw + PAD - s2) / 2 - DMIN);
          const int cls = (R == 1) ? 0 : (ph * 2 + pw);
#pragma unroll
          for (int t = 0; t < KT; ++t)
            acc[cls][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t][r * R + s2], b_cur[di * FP + dj], acc[cls][t],
                                                               0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if (step + 1 < NSTEP) {
#pragma unroll
        for (int i = 0; i < FP * FP; ++i) b_cur[i] = b_nxt[i];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int rs = 0; rs < RS; ++rs) a_cur[t][rs] = a_nxt[t][rs];
      }
    }
  }

  // ---- epilogue: lane = sub-pixel (a, b); rows 2a, 2a+1; each store is the (2b, 2b+1) pair
  const int n_out = n0 + ni_l;
  const int a_out = p0 + p_l;
  if (n_out < N) {
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int cbase = c0out + (wk * KT + t) * 32 + 4 * hi;
      const size_t obase = (((size_t)n_out * C + cbase) * H + 2 * a_out) * W + 2 * q_l;
      const size_t cstride = (size_t)H * W;
#pragma unroll
      for (int vb = 0; vb < 16; vb += 8) {
        float2 ad[8][2];  // the addend of half a tile is read before its first store, unconditionally (see conv_igemm)
        const int clast = C - 1 - cbase;
        if (addend) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int v = vb + u, cr = max(min((v & 3) + 8 * (v >> 2), clast), -cbase);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
              ad[u][ph] = *reinterpret_cast<const float2 *>(addend + obase + (long long)cr * (long long)cstride + ph * W);
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int v = vb + u, cr = (v & 3) + 8 * (v >> 2);
          if (cbase + cr < C) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
              float2 o;
              if (R == 1) {
                o.x = (ph == 0) ? acc[0][t][v] : 0.f;
                o.y = 0.f;
              } else {
                o.x = acc[ph * 2 + 0][t][v];
                o.y = acc[ph * 2 + 1][t][v];
              }
              if (addend) {
                o.x += ad[u][ph].x;
                o.y += ad[u][ph].y;
              }
              *reinterpret_cast<float2 *>(dx + obase + cr * cstride + ph * W) = o;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward-weight.  Workgroup: 4 waves = 2 k-tiles x 2 c-tiles of 32; each wave keeps RS accumulators (one per
// filter tap), so a 64-pixel chunk of x / dy staged in LDS feeds R*R MFMAs per pixel pair.
// grid = (K/64 rounded up, C/64 rounded up, nsplit); split s handles pixel chunks s, s+nsplit, ...
//
// The kernel runs ONE wave per SIMD (9 x 16 accumulator registers), so nothing hides instruction issue: whenever
// the wave is busy issuing loads / LDS traffic / address arithmetic, its matrix pipe drains (measured: a staging
// phase between MFMA groups costs 25-45 % — tools/micro/mfma_peak.hip has the ceilings).  Hence:
//   * everything that is not an MFMA is cut into slots of a few instructions and pinned (sched_barrier) BETWEEN
//     consecutive MFMAs, each of which keeps the pipe busy for 64 cycles;
//   * slots are branch free (one basic block per pixel pair): loads use clamped addresses + an AND mask instead of
//     predication, LDS stores of idle lanes go to a dump row;
//   * chunk i+1 travels global -> registers during the first half of chunk i's pixel pairs and registers -> the
//     OTHER LDS buffer during the second half; one barrier per chunk;
//   * x addresses are scalar base (image, channel) + one per-lane offset, so a load costs no vector ALU.
template <int R, int STRIDE, bool FULLC>
__global__ __launch_bounds__(256) void conv_wgrad(const float *__restrict__ x, const float *__restrict__ dy,
                                                  float *__restrict__ part, int N, int C, int H, int W, int K, int P,
                                                  int Q, int pad, int NI, int TP, int IH_t, int IW_t, int logQ,
                                                  int nchunks) {
  constexpr int RS = R * R;
  constexpr int PIXC = (STRIDE == 1) ? 64 : 32;  // pixels per chunk (stride 2: 32, so the patch stays <= 256 floats)
  constexpr int DROW = PIXC + 1;
  constexpr int NSTEP = PIXC / 2;                // pixel pairs per chunk
  constexpr int HALF = NSTEP / 2;
  constexpr int CPS = 64 / HALF;                 // x channels staged per pixel-pair step (4 or 8)
  constexpr int DN4 = 64 * PIXC / 4 / 256;       // float4 items of dy per thread and chunk (4 or 2)
  static_assert(DN4 * 4 == HALF, "one dy scalar is stored per step of the second half");
  constexpr int LOGPIX4 = (PIXC == 64) ? 4 : 3;  // log2(PIXC / 4)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  const int BUF = 64 * ch_stride + 64 * DROW + 256;  // floats per LDS buffer: xp[64 c][ch_stride] + dl[64 k][DROW] + dump

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int kt = wave & 1, ct = wave >> 1;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int tiles_per_img = (NI > 1) ? 1 : P / TP;
  const int planeHW = H * W, PQ = P * Q;
  const int logTP = __builtin_ctz(TP);
  const int cmax = C - c0;  // channels c < cmax of this tile are real (>= 64 when FULLC)

  f32x16 acc[RS];
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  // ---- staging state: one patch position per thread (PSZ <= 256) x 64 channels, and DN4 float4 of dy
  float xreg[64];
  float4 dreg[DN4];
  const float *xbase = x;      // uniform: &x[n0][c0][0][0] of the chunk being fetched
  unsigned xoff = 0;           // per lane: float offset of the patch position inside that image block (0 if masked)
  unsigned xmask = 0;          // per lane: ~0 if the position is inside the image, 0 for padding / idle lanes
  const float *dbase = dy;     // uniform: &dy[n0][k0][p0][0]
  unsigned doff[DN4], dmask[DN4];
  // patch position of this thread (chunk independent)
  const int e_ni = tid / (IH_t * IW_t);
  const int e_rem = tid - e_ni * (IH_t * IW_t);
  const int e_ih = e_rem / IW_t, e_iw = e_rem - e_ih * IW_t;
  const int ww = -pad + e_iw;
  const bool e_ok = tid < PSZ && ww >= 0 && ww < W;
  // LDS store addressing: idle lanes (tid >= PSZ) write every channel to their own dump slot
  const int st_base = tid < PSZ ? tid : 64 * ch_stride + 64 * DROW + tid;
  const int st_cmul = tid < PSZ ? ch_stride : 0;
  // dy float4 items of this thread (chunk independent part)
  int d_kk[DN4], d_ni[DN4], d_in[DN4];
#pragma unroll
  for (int i = 0; i < DN4; ++i) {
    const int e4 = tid + i * 256;
    const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4;
    const int q = m & (Q - 1), pr = m >> logQ;
    d_kk[i] = kk;
    d_ni[i] = pr >> logTP;
    d_in[i] = (pr & (TP - 1)) * Q + q;
  }
  auto aim = [&](int chunk) {  // source addresses for `chunk`
    int n0, p0;
    if (NI > 1) { n0 = chunk * NI; p0 = 0; }
    else { n0 = chunk / tiles_per_img; p0 = (chunk - n0 * tiles_per_img) * TP; }
    const int n = n0 + e_ni;
    const int h = p0 * STRIDE - pad + e_ih;
    const bool ok = e_ok && (n < N) && h >= 0 && h < H;
    xbase = x + ((size_t)n0 * C + c0) * planeHW;
    xoff = ok ? (unsigned)((e_ni * C) * planeHW + h * W + ww) : 0u;
    xmask = ok ? 0xffffffffu : 0u;
    dbase = dy + ((size_t)n0 * K + k0) * PQ + (size_t)p0 * Q;
#pragma unroll
    for (int i = 0; i < DN4; ++i) {
      const bool okd = (n0 + d_ni[i]) < N && (k0 + d_kk[i]) < K;
      doff[i] = okd ? (unsigned)((d_ni[i] * K + d_kk[i]) * PQ + d_in[i]) : 0u;
      dmask[i] = okd ? 0xffffffffu : 0u;
    }
  };
  // loads: unconditional, from a clamped (always valid) address; the value is masked when it is STORED to LDS half a
  // chunk later, so that nothing in a load slot depends on the load's result (no wait for memory between MFMAs)
  bool in_loop = false;  // (tuning switches only)
  auto load_x = [&](int c) {
    if ((SALUN_WGRAD_EXP & 1) && in_loop) return;
    const int cc = FULLC ? c : (c < cmax ? c : cmax - 1);
    xreg[c] = (xbase + (size_t)cc * planeHW)[xoff];
  };
  auto load_d = [&](int i) {
    if ((SALUN_WGRAD_EXP & 1) && in_loop) return;
    dreg[i] = *reinterpret_cast<const float4 *>(dbase + doff[i]);
  };
  auto store_x = [&](float *buf, int c) {
    if ((SALUN_WGRAD_EXP & 2) && in_loop) return;
    const unsigned m = FULLC ? xmask : (c < cmax ? xmask : 0u);
    buf[st_base + c * st_cmul] = __uint_as_float(__float_as_uint(xreg[c]) & m);
  };
  auto store_d = [&](float *dl, int sidx) {  // scalar #sidx (0 .. 4*DN4-1) of this thread's dy items
    if ((SALUN_WGRAD_EXP & 2) && in_loop) return;
    const int i = sidx >> 2, comp = sidx & 3;
    const int e4 = tid + i * 256;
    const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4 + comp;
    const float v = comp == 0 ? dreg[i].x : comp == 1 ? dreg[i].y : comp == 2 ? dreg[i].z : dreg[i].w;
    dl[kk * DROW + m] = __uint_as_float(__float_as_uint(v) & dmask[i]);
  };

  if (split < nchunks) {
    // prologue: chunk `split` into buffer 0
    aim(split);
#pragma unroll
    for (int c = 0; c < 64; ++c) load_x(c);
#pragma unroll
    for (int i = 0; i < DN4; ++i) load_d(i);
#pragma unroll
    for (int c = 0; c < 64; ++c) store_x(lds, c);
#pragma unroll
    for (int sidx = 0; sidx < 4 * DN4; ++sidx) store_d(lds + 64 * ch_stride, sidx);
    __syncthreads();

    int cur = 0;
    in_loop = true;
    for (int chunk = split; chunk < nchunks; chunk += nsplit) {
      const float *xp = lds + cur * BUF;
      const float *dl = xp + 64 * ch_stride;
      float *xp_n = lds + (cur ^ 1) * BUF;
      float *dl_n = xp_n + 64 * ch_stride;
      // the last chunk of this split re-stages itself (harmless) so that the loop body is branch free
      aim(chunk + nsplit < nchunks ? chunk + nsplit : chunk);
      const float *arow = dl + (kt * 32 + lo) * DROW + hi;
      const float *brow = xp + (ct * 32 + lo) * ch_stride + hi * STRIDE;
      // operands of pixel pair j: A = dy[k][j + hi], B = the R*R taps of x at that pixel.  The pair's patch offset
      // is wave-uniform: q = (j & (Q-1)) + hi (j even, Q >= 4), both lane halves in the same row.
      auto pair_off = [&](int j) {
        const int q0 = j & (Q - 1), pr = j >> logQ;
        const int ni = pr >> logTP, pl = pr & (TP - 1);
        return (ni * IH_t + pl * STRIDE) * IW_t + q0 * STRIDE;
      };
      float a_cur, b_cur[RS], a_nxt = 0.f, b_nxt[RS];
      {
        const float *bp = brow + pair_off(0);
        a_cur = arow[0];
#pragma unroll
        for (int t = 0; t < RS; ++t) b_cur[t] = bp[(t / R) * IW_t + (t % R)];
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const float *bp = brow + ((st + 1 < NSTEP) ? pair_off(2 * st + 2) : 0);
#pragma unroll
        for (int t = 0; t < RS; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // ---- slot t: a few non-MFMA instructions in the shadow of the MFMA just issued
          if (st + 1 < NSTEP) {
            if (t == 0) a_nxt = arow[2 * st + 2];
            b_nxt[t] = bp[(t / R) * IW_t + (t % R)];
          }
          // staging work of this step, spread over the slots (CPS items + 1 dy item per step)
          if (RS >= 9) {
            if (st < HALF) {
              if (CPS == 4 ? ((t & 1) == 1 && t / 2 < CPS) : (t < CPS)) load_x(st * CPS + (CPS == 4 ? t / 2 : t));
              if (t == RS - 1 && st < DN4) load_d(st);
            } else {
              if (CPS == 4 ? ((t & 1) == 1 && t / 2 < CPS) : (t < CPS))
                store_x(xp_n, (st - HALF) * CPS + (CPS == 4 ? t / 2 : t));
              if (t == RS - 1) store_d(dl_n, st - HALF);
            }
          } else if (t == RS - 1) {  // few MFMAs per step (1x1 filters): all staging items in the last slot
            if (st < HALF) {
#pragma unroll
              for (int cc = 0; cc < CPS; ++cc) load_x(st * CPS + cc);
              if (st < DN4) load_d(st);
            } else {
#pragma unroll
              for (int cc = 0; cc < CPS; ++cc) store_x(xp_n, (st - HALF) * CPS + cc);
              store_d(dl_n, st - HALF);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < RS; ++t) b_cur[t] = b_nxt[t];
      }
      if (!(SALUN_WGRAD_EXP & 4)) __syncthreads();
      cur ^= 1;
    }
  }
  // ---- partial[split][rs][k][c]: a wave's 32 result columns (c) are contiguous => coalesced 128-B stores;
  // the reduce kernel maps back to OIHW
  float *out = part + (size_t)split * K * C * RS;
  const int c = c0 + ct * 32 + lo;
  if ((SALUN_WGRAD_EXP & 8) && N > 0) return;
  if (c < C) {
#pragma unroll
    for (int t = 0; t < RS; ++t)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = k0 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
        if (k < K) out[((size_t)t * K + k) * C + c] = acc[t][v];
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward-weight, 3x3, vectorised staging (round 3).  Same tile, same LDS image, same MFMA / operand-read stream as
// `conv_wgrad<3, STRIDE, true>`; what changes is how a chunk travels global -> registers -> LDS.  The A/B builds of
// round 3 (tools/_run_wgrad_exp.sh) put 14 us of a 218 us layer on the loop's 68 `global_load_dword` per thread and
// chunk and 20 us on its 80 `ds_write_b32` — vector-memory and LDS-store INSTRUCTIONS are the scarce resource beside
// the matrix pipe, and the scalar mapping wasted them: one thread per patch position means 136 of 256 threads carry
// real data on a 32-wide image, the rest write a dump row.  Here an item is one 16-byte piece of a patch row's
// INTERIOR (the image's own columns; the halo columns are padding for every tile of these models and are zeroed once):
// 64 channels x F4C pieces per chunk, dealt out over all 256 threads — NIT = F4C / 4 loads of 16 bytes per thread
// (8 on a 32-wide image) and 4 NIT dword stores, all of them useful.  Rows above / below the image and images beyond N
// load from a clamped address and are AND-masked when stored, as before.
//   requires: R = 3, pad = 1, C % 64 == 0, W % 4 == 0, the tile spans whole image rows (Q == W / STRIDE),
//             F4C = NI * IH_t * (W / 4) a multiple of 4 with F4C / 4 == NIT.
template <int STRIDE, int NIT>
__global__ __launch_bounds__(256) void conv_wgrad_v(const float *__restrict__ x, const float *__restrict__ dy,
                                                    float *__restrict__ part, int N, int C, int H, int W, int K, int P,
                                                    int Q, int NI, int TP, int IH_t, int IW_t, int logQ, int nchunks) {
  constexpr int R = 3, RS = 9, PAD = 1;
  constexpr int PIXC = (STRIDE == 1) ? 64 : 32;
  constexpr int DROW = PIXC + 1;
  constexpr int NSTEP = PIXC / 2;
  constexpr int HALF = NSTEP / 2;
  constexpr int DN4 = 64 * PIXC / 4 / 256;       // float4 items of dy per thread and chunk (4 or 2)
  constexpr int LOGPIX4 = (PIXC == 64) ? 4 : 3;
  constexpr int NLD = NIT + DN4;                  // load instructions per thread and chunk
  constexpr int LPS = (NLD + HALF - 1) / HALF;    // ... per step of the first half
  constexpr int NST = 4 * (NIT + DN4);            // dword stores per thread and chunk
  constexpr int SPS = (NST + HALF - 1) / HALF;    // ... per step of the second half
  static_assert(LPS <= RS && SPS <= RS, "one staging instruction per MFMA slot at most");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  const int BUF = 64 * ch_stride + 64 * DROW + 256;  // same LDS image as conv_wgrad

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int kt = wave & 1, ct = wave >> 1;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int tiles_per_img = (NI > 1) ? 1 : P / TP;
  const int planeHW = H * W, PQ = P * Q;
  const int logTP = __builtin_ctz(TP);
  const int W4 = W >> 2;
  const int F4C = NI * IH_t * W4;

  f32x16 acc[RS];
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  // ---- the NIT x-items of this thread (chunk independent part)
  int it_lds[NIT];    // float offset of the piece inside one LDS buffer
  int it_g[NIT];      // float offset from &x[n0][c0][0][0] for p0 = 0 (can be negative: rows above the image)
  int it_ih[NIT], it_ni[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int e = tid + k * 256;
    const int c = e / F4C, r = e - c * F4C;
    const int ni = r / (IH_t * W4), r2 = r - ni * (IH_t * W4);
    const int ih = r2 / W4, g4 = r2 - ih * W4;
    it_lds[k] = c * ch_stride + (ni * IH_t + ih) * IW_t + PAD + 4 * g4;
    it_g[k] = (ni * C + c) * planeHW + (ih - PAD) * W + 4 * g4;
    it_ih[k] = ih;
    it_ni[k] = ni;
  }
  float4 xreg[NIT];
  float4 dreg[DN4];
  const float *xbase = x;
  unsigned xoff[NIT], xmask[NIT];
  const float *dbase = dy;
  unsigned doff[DN4], dmask[DN4];
  int d_kk[DN4], d_ni[DN4], d_in[DN4];
#pragma unroll
  for (int i = 0; i < DN4; ++i) {
    const int e4 = tid + i * 256;
    const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4;
    const int q = m & (Q - 1), pr = m >> logQ;
    d_kk[i] = kk;
    d_ni[i] = pr >> logTP;
    d_in[i] = (pr & (TP - 1)) * Q + q;
  }
  auto aim = [&](int chunk) {
    int n0, p0;
    if (NI > 1) { n0 = chunk * NI; p0 = 0; }
    else { n0 = chunk / tiles_per_img; p0 = (chunk - n0 * tiles_per_img) * TP; }
    xbase = x + ((size_t)n0 * C + c0) * planeHW;
    const int h0 = p0 * STRIDE;  // input row of patch row PAD
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int h = h0 - PAD + it_ih[k];
      const bool ok = (n0 + it_ni[k]) < N && h >= 0 && h < H;
      xoff[k] = ok ? (unsigned)(it_g[k] + h0 * W) : 0u;
      xmask[k] = ok ? 0xffffffffu : 0u;
    }
    dbase = dy + ((size_t)n0 * K + k0) * PQ + (size_t)p0 * Q;
#pragma unroll
    for (int i = 0; i < DN4; ++i) {
      const bool okd = (n0 + d_ni[i]) < N && (k0 + d_kk[i]) < K;
      doff[i] = okd ? (unsigned)((d_ni[i] * K + d_kk[i]) * PQ + d_in[i]) : 0u;
      dmask[i] = okd ? 0xffffffffu : 0u;
    }
  };
  auto load_item = [&](int li) {  // li < NLD: x items first, then dy items
    if (li < NIT) xreg[li] = *reinterpret_cast<const float4 *>(xbase + xoff[li]);
    else dreg[li - NIT] = *reinterpret_cast<const float4 *>(dbase + doff[li - NIT]);
  };
  auto comp_of = [](const float4 &v, int comp) { return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w; };
  auto store_item = [&](float *buf, int si) {  // si < NST: the dwords of the x items, then those of the dy items
    if (si < 4 * NIT) {
      const int k = si >> 2, comp = si & 3;
      buf[it_lds[k] + comp] = __uint_as_float(__float_as_uint(comp_of(xreg[k], comp)) & xmask[k]);
    } else {
      const int sd = si - 4 * NIT, i = sd >> 2, comp = sd & 3;
      const int e4 = tid + i * 256;
      const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4 + comp;
      (buf + 64 * ch_stride)[kk * DROW + m] = __uint_as_float(__float_as_uint(comp_of(dreg[i], comp)) & dmask[i]);
    }
  };

  // the halo columns (and everything else no item writes) stay zero for the whole kernel
  for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 0.f;
  __syncthreads();

  if (split < nchunks) {
    aim(split);
#pragma unroll
    for (int li = 0; li < NLD; ++li) load_item(li);
#pragma unroll
    for (int si = 0; si < NST; ++si) store_item(lds, si);
    __syncthreads();

    int cur = 0;
    for (int chunk = split; chunk < nchunks; chunk += nsplit) {
      const float *xp = lds + cur * BUF;
      const float *dl = xp + 64 * ch_stride;
      float *buf_n = lds + (cur ^ 1) * BUF;
      aim(chunk + nsplit < nchunks ? chunk + nsplit : chunk);  // the last chunk re-stages itself: branch-free body
      const float *arow = dl + (kt * 32 + lo) * DROW + hi;
      const float *brow = xp + (ct * 32 + lo) * ch_stride + hi * STRIDE;
      auto pair_off = [&](int j) {
        const int q0 = j & (Q - 1), pr = j >> logQ;
        const int ni = pr >> logTP, pl = pr & (TP - 1);
        return (ni * IH_t + pl * STRIDE) * IW_t + q0 * STRIDE;
      };
      float a_cur, b_cur[RS], a_nxt = 0.f, b_nxt[RS];
      {
        const float *bp = brow + pair_off(0);
        a_cur = arow[0];
#pragma unroll
        for (int t = 0; t < RS; ++t) b_cur[t] = bp[(t / R) * IW_t + (t % R)];
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const float *bp = brow + ((st + 1 < NSTEP) ? pair_off(2 * st + 2) : 0);
#pragma unroll
        for (int t = 0; t < RS; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < NSTEP) {
            if (t == 0) a_nxt = arow[2 * st + 2];
            b_nxt[t] = bp[(t / R) * IW_t + (t % R)];
          }
          if (st < HALF) {
            if (t < LPS && st * LPS + t < NLD) load_item(st * LPS + t);
          } else {
            if (t < SPS && (st - HALF) * SPS + t < NST) store_item(buf_n, (st - HALF) * SPS + t);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < RS; ++t) b_cur[t] = b_nxt[t];
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  float *out = part + (size_t)split * K * C * RS;
  const int c = c0 + ct * 32 + lo;
#pragma unroll
  for (int t = 0; t < RS; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = k0 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
      if (k < K) out[((size_t)t * K + k) * C + c] = acc[t][v];
    }
}

// ---------------------------------------------------------------------------------------------------
// backward-weight, 3x3, TWO waves per SIMD (round 3).  `conv_wgrad_v` keeps nine accumulators (144 registers) per wave,
// so one wave owns a SIMD and every LDS-latency hiccup of its single instruction stream idles the matrix pipe (its bare
// MFMA + operand-read loop runs at 108-111 TFLOP/s where the two-waves-per-SIMD `conv_igemm` loop reaches 135-141).
// Here the workgroup has EIGHT waves: waves 0-3 take taps 0-4, waves 4-7 taps 5-8 of the same four 32 k x 32 c
// quadrants (5 / 4 accumulators = 80 / 64 registers), so every SIMD carries one wave of each group — nine MFMAs per
// pixel pair per SIMD as before, issued by two independent streams; the dY operand is read by both (11 LDS reads per
// 9 MFMAs instead of 10).  Staging is `conv_wgrad_v`'s, dealt over 512 threads (half the instructions per thread).
// Same LDS image, same partial-sum layout, same summation order per accumulator => bit-identical results.
template <int STRIDE, int NIT, int NTAP, int TBASE>
__device__ __forceinline__ void wgrad_w_body(const float *__restrict__ x, const float *__restrict__ dy,
                                             float *__restrict__ part, int N, int C, int H, int W, int K, int P, int Q,
                                             int NI, int TP, int IH_t, int IW_t, int logQ, int nchunks, float *lds) {
  constexpr int R = 3, RS = 9, PAD = 1;
  constexpr int PIXC = (STRIDE == 1) ? 64 : 32;
  constexpr int DROW = PIXC + 1;
  constexpr int NSTEP = PIXC / 2;
  constexpr int HALF = NSTEP / 2;
  constexpr int DN4 = 64 * PIXC / 4 / 512;       // float4 items of dy per thread and chunk (2 or 1)
  constexpr int LOGPIX4 = (PIXC == 64) ? 4 : 3;
  constexpr int NLD = NIT + DN4;
  constexpr int LPS = (NLD + HALF - 1) / HALF;
  constexpr int NST = 4 * (NIT + DN4);
  constexpr int SPS = (NST + HALF - 1) / HALF;
  static_assert(LPS <= 4 && SPS <= 4, "one staging instruction per MFMA slot at most (4 slots in the short group)");
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  const int BUF = 64 * ch_stride + 64 * DROW + 512;  // x patch + dy tile + dump row (items beyond the patch)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int wq = wave & 3, kt = wq & 1, ct = wq >> 1;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int tiles_per_img = (NI > 1) ? 1 : P / TP;
  const int planeHW = H * W, PQ = P * Q;
  const int logTP = __builtin_ctz(TP);
  const int W4 = W >> 2;
  const int F4C = NI * IH_t * W4;

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  int it_lds[NIT], it_g[NIT], it_ih[NIT], it_ni[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int e = tid + k * 512;
    const bool real = e < 64 * F4C;              // the last item of a thread may fall beyond the patch
    const int ee = real ? e : 0;
    const int c = ee / F4C, r = ee - c * F4C;
    const int ni = r / (IH_t * W4), r2 = r - ni * (IH_t * W4);
    const int ih = r2 / W4, g4 = r2 - ih * W4;
    it_lds[k] = real ? c * ch_stride + (ni * IH_t + ih) * IW_t + PAD + 4 * g4 : 64 * ch_stride + 64 * DROW + (tid & ~3);
    it_g[k] = (ni * C + c) * planeHW + (ih - PAD) * W + 4 * g4;
    it_ih[k] = real ? ih : -(1 << 20);           // never inside the image: loads clamp, stores are masked
    it_ni[k] = ni;
  }
  float4 xreg[NIT];
  float4 dreg[DN4];
  const float *xbase = x;
  unsigned xoff[NIT], xmask[NIT];
  const float *dbase = dy;
  unsigned doff[DN4], dmask[DN4];
  int d_kk[DN4], d_ni[DN4], d_in[DN4];
#pragma unroll
  for (int i = 0; i < DN4; ++i) {
    const int e4 = tid + i * 512;
    const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4;
    const int q = m & (Q - 1), pr = m >> logQ;
    d_kk[i] = kk;
    d_ni[i] = pr >> logTP;
    d_in[i] = (pr & (TP - 1)) * Q + q;
  }
  auto aim = [&](int chunk) {
    int n0, p0;
    if (NI > 1) { n0 = chunk * NI; p0 = 0; }
    else { n0 = chunk / tiles_per_img; p0 = (chunk - n0 * tiles_per_img) * TP; }
    xbase = x + ((size_t)n0 * C + c0) * planeHW;
    const int h0 = p0 * STRIDE;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int h = h0 - PAD + it_ih[k];
      const bool ok = (n0 + it_ni[k]) < N && h >= 0 && h < H;
      xoff[k] = ok ? (unsigned)(it_g[k] + h0 * W) : 0u;
      xmask[k] = ok ? 0xffffffffu : 0u;
    }
    dbase = dy + ((size_t)n0 * K + k0) * PQ + (size_t)p0 * Q;
#pragma unroll
    for (int i = 0; i < DN4; ++i) {
      const bool okd = (n0 + d_ni[i]) < N && (k0 + d_kk[i]) < K;
      doff[i] = okd ? (unsigned)((d_ni[i] * K + d_kk[i]) * PQ + d_in[i]) : 0u;
      dmask[i] = okd ? 0xffffffffu : 0u;
    }
  };
  auto load_item = [&](int li) {
    if (li < NIT) xreg[li] = *reinterpret_cast<const float4 *>(xbase + xoff[li]);
    else dreg[li - NIT] = *reinterpret_cast<const float4 *>(dbase + doff[li - NIT]);
  };
  auto comp_of = [](const float4 &v, int comp) { return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w; };
  auto store_item = [&](float *buf, int si) {
    if (si < 4 * NIT) {
      const int k = si >> 2, comp = si & 3;
      buf[it_lds[k] + comp] = __uint_as_float(__float_as_uint(comp_of(xreg[k], comp)) & xmask[k]);
    } else {
      const int sd = si - 4 * NIT, i = sd >> 2, comp = sd & 3;
      const int e4 = tid + i * 512;
      const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4 + comp;
      (buf + 64 * ch_stride)[kk * DROW + m] = __uint_as_float(__float_as_uint(comp_of(dreg[i], comp)) & dmask[i]);
    }
  };

  for (int i = tid; i < 2 * BUF; i += 512) lds[i] = 0.f;  // halo columns stay zero for the whole kernel
  __syncthreads();

  if (split < nchunks) {
    aim(split);
#pragma unroll
    for (int li = 0; li < NLD; ++li) load_item(li);
#pragma unroll
    for (int si = 0; si < NST; ++si) store_item(lds, si);
    __syncthreads();

    int cur = 0;
    for (int chunk = split; chunk < nchunks; chunk += nsplit) {
      const float *xp = lds + cur * BUF;
      const float *dl = xp + 64 * ch_stride;
      float *buf_n = lds + (cur ^ 1) * BUF;
      aim(chunk + nsplit < nchunks ? chunk + nsplit : chunk);
      const float *arow = dl + (kt * 32 + lo) * DROW + hi;
      const float *brow = xp + (ct * 32 + lo) * ch_stride + hi * STRIDE;
      auto pair_off = [&](int j) {
        const int q0 = j & (Q - 1), pr = j >> logQ;
        const int ni = pr >> logTP, pl = pr & (TP - 1);
        return (ni * IH_t + pl * STRIDE) * IW_t + q0 * STRIDE;
      };
      float a_cur, b_cur[NTAP], a_nxt = 0.f, b_nxt[NTAP];
      {
        const float *bp = brow + pair_off(0);
        a_cur = arow[0];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) b_cur[t] = bp[((TBASE + t) / R) * IW_t + ((TBASE + t) % R)];
      }
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const float *bp = brow + ((st + 1 < NSTEP) ? pair_off(2 * st + 2) : 0);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < NSTEP) {
            if (t == 0) a_nxt = arow[2 * st + 2];
            b_nxt[t] = bp[((TBASE + t) / R) * IW_t + ((TBASE + t) % R)];
          }
          if (st < HALF) {
            if (t < LPS && st * LPS + t < NLD) load_item(st * LPS + t);
          } else {
            if (t < SPS && (st - HALF) * SPS + t < NST) store_item(buf_n, (st - HALF) * SPS + t);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_cur = a_nxt;
#pragma unroll
        for (int t = 0; t < NTAP; ++t) b_cur[t] = b_nxt[t];
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  float *out = part + (size_t)split * K * C * RS;
  const int c = c0 + ct * 32 + lo;
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = k0 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
      if (k < K) out[((size_t)(TBASE + t) * K + k) * C + c] = acc[t][v];
    }
}

template <int STRIDE, int NIT>
__global__ __launch_bounds__(512) void conv_wgrad_w(const float *__restrict__ x, const float *__restrict__ dy,
                                                    float *__restrict__ part, int N, int C, int H, int W, int K, int P,
                                                    int Q, int NI, int TP, int IH_t, int IW_t, int logQ, int nchunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // wave-uniform: both groups execute the same number of barriers (one after the clear, one after the prologue, one
  // per chunk), at different program counters
  if ((threadIdx.x >> 8) == 0)
    wgrad_w_body<STRIDE, NIT, 5, 0>(x, dy, part, N, C, H, W, K, P, Q, NI, TP, IH_t, IW_t, logQ, nchunks, lds);
  else
    wgrad_w_body<STRIDE, NIT, 4, 5>(x, dy, part, N, C, H, W, K, P, Q, NI, TP, IH_t, IW_t, logQ, nchunks, lds);
}

// ---------------------------------------------------------------------------------------------------
// backward-weight of a 1x1 convolution (round 3): dW[k][c] = sum_{n,p,q} dY[n][k][p][q] * X[n][c][p*S][q*S] — a plain
// GEMM whose reduction dimension (the pixels) is the contiguous one of BOTH operands.  `conv_wgrad<1, S>` treated it as
// a 3x3 kernel with one tap: one accumulator per wave, two LDS reads and a slot full of staging per MFMA — 20 TFLOP/s,
// 9 % of the DDPM step (its attention / shortcut projections) and 3 % of the ResNet step (the downsample convs).
// Here a workgroup owns 128 k x 128 c, a wave 64 x 64 (2 x 2 MFMA tiles: four reads feed four MFMAs per pixel pair),
// both operand tiles are staged as [row][64 pixels (+1)] straight from 16-byte loads along the pixels, double
// buffered, the next chunk's 16 loads and 64 dword stores dealt out one per MFMA slot as in conv_wgrad_v.
// grid = (ceil(K/128), ceil(C/128), nsplit); partials [split][k][c] -> conv_wgrad_reduce (RS = 1).
//   requires: P*Q a multiple of 4, and of 64 or a divisor of 64; Q a multiple of 4 for STRIDE 2 (H = 2P, W = 2Q).
template <int STRIDE>
__global__ __launch_bounds__(256) void conv_wgrad_1x1(const float *__restrict__ x, const float *__restrict__ dy,
                                                      float *__restrict__ part, int N, int C, int H, int W, int K,
                                                      int P, int Q, int nchunks) {
  constexpr int PIXC = 64, DROW = PIXC + 1, NSTEP = PIXC / 2, HALF = NSTEP / 2;
  constexpr int NA = 128 * PIXC / 4 / 256;        // float4 items of dY per thread and chunk (8)
  constexpr int NB = NA;                          // ... of x (each made of STRIDE loads)
  constexpr int NLD = NA + NB * STRIDE;           // load instructions per thread and chunk (16 / 24)
  constexpr int LPS = (NLD + HALF - 1) / HALF;    // per step of the first half
  constexpr int NST = 4 * (NA + NB);              // dword stores per thread and chunk (64)
  constexpr int SPS = (NST + HALF - 1) / HALF;    // per step of the second half (4)
  static_assert(LPS <= 4 && SPS <= 4, "one staging instruction per MFMA slot");
  constexpr int BUF = 2 * 128 * DROW;             // floats per LDS buffer: A tile then B tile
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int kt = wave & 1, ct = wave >> 1;
  const int k0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int PQ = P * Q, HW = H * W;
  const int per_img = PQ >= PIXC ? PQ / PIXC : 1;   // chunks per image ...
  const int NI = PQ >= PIXC ? 1 : PIXC / PQ;        // ... or images per chunk

  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[t][u][v] = 0.f;

  // ---- staging items (chunk independent part): item i covers row i_row, pixels i_m .. i_m+3 of the chunk
  int i_lds[NA], i_dn[NA], i_pq[NA], i_row[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int e = tid + i * 256;
    const int row = e >> 4, m = (e & 15) * 4;
    i_row[i] = row;
    i_lds[i] = row * DROW + m;
    i_dn[i] = NI > 1 ? m / PQ : 0;          // image inside the chunk
    i_pq[i] = NI > 1 ? m % PQ : m;          // pixel inside that image (+ the chunk's first pixel when NI == 1)
  }
  float4 areg[NA], breg[NB * STRIDE];
  unsigned aoff[NA], amask[NA], boff[NB], bmask[NB];
  const float *abase = dy, *bbase = x;
  auto aim = [&](int chunk) {
    const int n0 = NI > 1 ? chunk * NI : chunk / per_img;
    const int m0 = NI > 1 ? 0 : (chunk - n0 * per_img) * PIXC;
    abase = dy + ((size_t)n0 * K + k0) * PQ;
    bbase = x + ((size_t)n0 * C + c0) * HW;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int n = n0 + i_dn[i], pq = m0 + i_pq[i];
      const bool oka = n < N && (k0 + i_row[i]) < K;
      aoff[i] = oka ? (unsigned)((i_dn[i] * K + i_row[i]) * PQ + pq) : 0u;
      amask[i] = oka ? 0xffffffffu : 0u;
      const bool okb = n < N && (c0 + i_row[i]) < C;
      const int p = pq / Q, q = pq - p * Q;  // STRIDE 2: the 4 output pixels q..q+3 read input columns 2q .. 2q+6
      const int xin = STRIDE == 1 ? pq : (p * STRIDE) * W + q * STRIDE;
      boff[i] = okb ? (unsigned)((i_dn[i] * C + i_row[i]) * HW + xin) : 0u;
      bmask[i] = okb ? 0xffffffffu : 0u;
    }
  };
  auto load_item = [&](int li) {
    if (li < NA) areg[li] = *reinterpret_cast<const float4 *>(abase + aoff[li]);
    else {
      const int j = li - NA, i = j / STRIDE, half = j % STRIDE;
      breg[j] = *reinterpret_cast<const float4 *>(bbase + boff[i] + 4 * half);
    }
  };
  auto comp_of = [](const float4 &v, int comp) { return comp == 0 ? v.x : comp == 1 ? v.y : comp == 2 ? v.z : v.w; };
  auto store_item = [&](float *buf, int si) {
    if (si < 4 * NA) {
      const int i = si >> 2, comp = si & 3;
      buf[i_lds[i] + comp] = __uint_as_float(__float_as_uint(comp_of(areg[i], comp)) & amask[i]);
    } else {
      const int sb = si - 4 * NA, i = sb >> 2, comp = sb & 3;
      // STRIDE 2: output pixel q + comp reads input column 2 (q + comp): components x, z of the two loads
      const float v = STRIDE == 1 ? comp_of(breg[i], comp) : comp_of(breg[i * 2 + (comp >> 1)], (comp & 1) * 2);
      (buf + 128 * DROW)[i_lds[i] + comp] = __uint_as_float(__float_as_uint(v) & bmask[i]);
    }
  };

  if (split < nchunks) {
    aim(split);
#pragma unroll
    for (int li = 0; li < NLD; ++li) load_item(li);
#pragma unroll
    for (int si = 0; si < NST; ++si) store_item(lds, si);
    __syncthreads();
    int cur = 0;
    for (int chunk = split; chunk < nchunks; chunk += nsplit) {
      const float *al = lds + cur * BUF;
      const float *bl = al + 128 * DROW;
      float *buf_n = lds + (cur ^ 1) * BUF;
      aim(chunk + nsplit < nchunks ? chunk + nsplit : chunk);  // the last chunk re-stages itself: branch-free body
      const float *arow = al + (kt * 64 + lo) * DROW + hi;
      const float *brow = bl + (ct * 64 + lo) * DROW + hi;
      float a_cur[2], b_cur[2], a_nxt[2] = {0.f, 0.f}, b_nxt[2] = {0.f, 0.f};
      a_cur[0] = arow[0]; a_cur[1] = arow[32 * DROW];
      b_cur[0] = brow[0]; b_cur[1] = brow[32 * DROW];
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          const int t = sl >> 1, u = sl & 1;
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], b_cur[u], acc[t][u], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (st + 1 < NSTEP) {  // one operand of the next pixel pair per slot
            if (sl < 2) a_nxt[sl] = arow[sl * 32 * DROW + 2 * st + 2];
            else b_nxt[sl - 2] = brow[(sl - 2) * 32 * DROW + 2 * st + 2];
          }
          if (st < HALF) {
            if (sl < LPS && st * LPS + sl < NLD) load_item(st * LPS + sl);
          } else {
            if (sl < SPS && (st - HALF) * SPS + sl < NST) store_item(buf_n, (st - HALF) * SPS + sl);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1];
        b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  float *out = part + (size_t)split * K * C;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = c0 + ct * 64 + u * 32 + lo;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int k = k0 + kt * 64 + t * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
        if (k < K && c < C) out[(size_t)k * C + c] = acc[t][u][v];
      }
    }
}

// backward-weight for tiny input-channel counts (the RGB stem: C*R*R <= 32).  The general kernel would spend a
// 64-channel tile and R*R accumulators on 3 real channels; here the 32 MFMA columns are the (c, r, s) combinations
// themselves (27 for RGB 3x3), one accumulator per wave, and the kernel is bound by reading dY once.
// Workgroup: 4 waves = 2 k-tiles x 2 halves of the chunk's pixel pairs; partial slot = split*2 + half.
template <int R, int STRIDE>
__global__ __launch_bounds__(256) void conv_wgrad_smallc(const float *__restrict__ x, const float *__restrict__ dy,
                                                         float *__restrict__ part, int N, int C, int H, int W, int K,
                                                         int P, int Q, int pad, int NI, int TP, int IH_t, int IW_t,
                                                         int logQ, int nchunks) {
  constexpr int RS = R * R;
  constexpr int PIXC = (STRIDE == 1) ? 64 : 32;
  constexpr int DROW = PIXC + 1;
  constexpr int NSTEP = PIXC / 2;
  constexpr int DN4 = 64 * PIXC / 4 / 256;
  constexpr int LOGPIX4 = (PIXC == 64) ? 4 : 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int PSZ = NI * IH_t * IW_t;
  const int ch_stride = PSZ | 1;
  float *xp = lds;                  // [C][ch_stride]
  float *dl = lds + C * ch_stride;  // [64 k][DROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int kt = wave & 1, half = wave >> 1;
  const int k0 = blockIdx.x * 64;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int tiles_per_img = (NI > 1) ? 1 : P / TP;
  const int planeHW = H * W, PQ = P * Q;
  const int logTP = __builtin_ctz(TP);
  const int ncol = C * RS;  // real MFMA columns (<= 32)

  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;

  // this lane's column: (c, r, s) and its offset inside the staged patch
  const int col_c = lo / RS, col_rs = lo - col_c * RS;
  const int col_r = col_rs / R, col_s = col_rs - col_r * R;
  const bool col_ok = lo < ncol;
  const int lane_off = col_ok ? (col_c * ch_stride + col_r * IW_t + col_s + hi * STRIDE) : 0;
  const unsigned bmask = col_ok ? 0xffffffffu : 0u;
  // patch position staged by this thread
  const int e_ni = tid / (IH_t * IW_t);
  const int e_rem = tid - e_ni * (IH_t * IW_t);
  const int e_ih = e_rem / IW_t, e_iw = e_rem - e_ih * IW_t;
  const int ww = -pad + e_iw;

  for (int chunk = split; chunk < nchunks; chunk += nsplit) {
    int n0, p0;
    if (NI > 1) { n0 = chunk * NI; p0 = 0; }
    else { n0 = chunk / tiles_per_img; p0 = (chunk - n0 * tiles_per_img) * TP; }
    // ---- global -> registers
    float xr[4];
    {
      const int n = n0 + e_ni;
      const int h = p0 * STRIDE - pad + e_ih;
      const bool ok = tid < PSZ && n < N && h >= 0 && h < H && ww >= 0 && ww < W;
      const float *src = x + (ok ? ((size_t)n * C * planeHW + (size_t)h * W + ww) : 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) xr[c] = (ok && c < C) ? src[(size_t)c * planeHW] : 0.f;
    }
    float4 dr[DN4];
#pragma unroll
    for (int i = 0; i < DN4; ++i) {
      const int e4 = tid + i * 256;
      const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4;
      const int q = m & (Q - 1), pr = m >> logQ;
      const int ni = pr >> logTP, pl = pr & (TP - 1);
      const int n = n0 + ni;
      const bool okd = n < N && (k0 + kk) < K;
      dr[i] = okd ? *reinterpret_cast<const float4 *>(dy + ((size_t)n * K + (k0 + kk)) * PQ + (size_t)(p0 + pl) * Q + q)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // previous chunk fully consumed
    if (tid < PSZ) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < C) xp[c * ch_stride + tid] = xr[c];
    }
#pragma unroll
    for (int i = 0; i < DN4; ++i) {
      const int e4 = tid + i * 256;
      const int kk = e4 >> LOGPIX4, m = (e4 & (PIXC / 4 - 1)) * 4;
      float *d = dl + kk * DROW + m;
      d[0] = dr[i].x; d[1] = dr[i].y; d[2] = dr[i].z; d[3] = dr[i].w;
    }
    __syncthreads();
    const float *arow = dl + (kt * 32 + lo) * DROW + hi;
    const float *brow = xp + lane_off;
#pragma unroll
    for (int st = 0; st < NSTEP / 2; ++st) {
      const int j = 2 * (half * (NSTEP / 2) + st);
      const int q0 = j & (Q - 1), pr = j >> logQ;
      const int ni = pr >> logTP, pl = pr & (TP - 1);
      const int off = (ni * IH_t + pl * STRIDE) * IW_t + q0 * STRIDE;
      const float a = arow[j];
      const float b = __uint_as_float(__float_as_uint(brow[off]) & bmask);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // partial[split*2 + half][rs][k][c]
  float *out = part + (size_t)(split * 2 + half) * K * C * RS;
  if (col_ok) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = k0 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
      if (k < K) out[((size_t)col_rs * K + k) * C + col_c] = acc[v];
    }
  }
}

// dw (+)= sum_s part[s] in a fixed order.  Partials are [split][rs][k][c]; the result is written in OIHW.
// A thread owns FOUR consecutive partial indices (one 16-byte load per split) and one of G split groups
// (G = 2^LOGG <= 16, chosen by the host so that G <= nsplit): group g adds splits g, g+G, g+2G, ... in that order,
// eight loads in flight; the G group sums are then folded g = 0..G-1 through LDS.  256/G outputs x G groups per
// workgroup: for the 256-way split of a 64x64 layer that is 576 workgroups x 16 loads per thread, for the 4-way
// split of a 512x512 layer 9,216 workgroups x 1 load — always thousands of 16-byte loads in flight, which the
// round-2 kernel (4-byte loads, 32 outputs per workgroup, half its threads idle when nsplit < 8) did not have.
// Deterministic: the summation order depends only on (nsplit, G).
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream4(const float4 *p) {  // one 16-byte load, streamed past the caches
  const f32x4v v = __builtin_nontemporal_load(reinterpret_cast<const f32x4v *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

template <int LOGG>
__global__ __launch_bounds__(256) void conv_wgrad_reduce(const float *__restrict__ part, float *__restrict__ dw,
                                                         int64_t n, int nsplit, int accumulate, int KC, int RS) {
  constexpr int G = 1 << LOGG, OUTS = 256 / G;
  __shared__ float4 red[G][OUTS];
  const int o = threadIdx.x & (OUTS - 1), g = threadIdx.x >> (8 - LOGG);
  const int64_t i4 = (int64_t)blockIdx.x * OUTS + o;  // float4 index in [rs][k][c] order
  const int64_t n4 = n >> 2;
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < n4) {
    const float4 *src = reinterpret_cast<const float4 *>(part) + i4;
    int j = g;
    for (; j + 7 * G < nsplit; j += 8 * G) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldg_stream4(src + (size_t)(j + u * G) * n4);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }
    }
    for (; j < nsplit; j += G) {
      const float4 v = ldg_stream4(src + (size_t)j * n4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
  }
  if (G > 1) {
    red[g][o] = s4;
    __syncthreads();
  }
  if (g == 0 && i4 < n4) {
#pragma unroll
    for (int k = 1; k < G; ++k) {
      const float4 v = red[k][o];
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    const float t[4] = {s4.x, s4.y, s4.z, s4.w};
    int64_t dst[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t i = i4 * 4 + e;
      const int64_t rs = i / KC, kc = i - rs * KC;
      dst[e] = kc * RS + rs;
    }
    float old[4] = {0.f, 0.f, 0.f, 0.f};  // `+=`: the four old values are read before the first store
    if (accumulate) {
#pragma unroll
      for (int e = 0; e < 4; ++e) old[e] = dw[dst[e]];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[dst[e]] = accumulate ? old[e] + t[e] : t[e];
  }
}

// 3x3 layers with many (k, c) pairs: the same sums, written as whole OIHW rows.  conv_wgrad_reduce stores each sum as a
// lone dword 36 bytes from its neighbour's and the nine taps of a weight from nine different workgroups — for a
// 512 x 512 layer 2.4 M partial-line writes into a 9.4 MB gradient.  Here a block owns 32 consecutive (k, c) pairs and
// ALL nine taps: thread (g, rs, q) sums float4 column q of tap rs over the splits g, g + 3, ... (128-byte runs of each
// partial row), the three groups are folded in LDS in index order, and the 288 sums leave as the 288 consecutive
// floats they are in dw.  Deterministic: the order depends only on nsplit.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_rows(const float *__restrict__ part, float *__restrict__ dw,
                                                              int64_t KC, int nsplit, int accumulate) {
  constexpr int RS = 9, KB = 32, G = 3;
  __shared__ float4 red[G][RS][KB / 4];
  __shared__ float outb[KB * RS];
  const int tid = threadIdx.x, q = tid & 7, r = tid >> 3, rs = r % RS, g = r / RS;  // g == 3: idle (threads 216 .. 255)
  const int64_t kc0 = (int64_t)blockIdx.x * KB, n4 = KC * RS / 4;
  float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g < G) {
    const float4 *src = reinterpret_cast<const float4 *>(part + rs * KC + kc0) + q;
    int j = g;
    for (; j + 7 * G < nsplit; j += 8 * G) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldg_stream4(src + (size_t)(j + u * G) * n4);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w; }
    }
    for (; j < nsplit; j += G) {
      const float4 v = ldg_stream4(src + (size_t)j * n4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    red[g][rs][q] = s4;
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int k = 1; k < G; ++k) {
      const float4 v = red[k][rs][q];
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    outb[(4 * q + 0) * RS + rs] = s4.x;
    outb[(4 * q + 1) * RS + rs] = s4.y;
    outb[(4 * q + 2) * RS + rs] = s4.z;
    outb[(4 * q + 3) * RS + rs] = s4.w;
  }
  __syncthreads();
  float *d = dw + kc0 * RS;
  const bool second = tid < KB * RS - 256;
  float old0 = 0.f, old1 = 0.f;  // `+=`: both old values are read before the first store
  if (accumulate) {
    old0 = d[tid];
    if (second) old1 = d[256 + tid];
  }
  d[tid] = accumulate ? old0 + outb[tid] : outb[tid];
  if (second) d[256 + tid] = accumulate ? old1 + outb[256 + tid] : outb[256 + tid];
}

// n not a multiple of 4 (odd test shapes; no layer of the three models): one output per thread, splits in order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_scalar(const float *__restrict__ part, float *__restrict__ dw,
                                                                int64_t n, int nsplit, int accumulate, int KC, int RS) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int j = 0; j < nsplit; ++j) t += part[(size_t)j * n + i];
  const int64_t rs = i / KC, kc = i - rs * KC;
  const int64_t dst = kc * RS + rs;
  dw[dst] = accumulate ? dw[dst] + t : t;
}

inline void launch_wgrad_reduce(const float *part, float *dw, int64_t n, int nsplit, int accumulate, int KC, int RS,
                                hipStream_t st) {
#ifdef SALUN_WGRAD_EXP_NOFOLD  // lab builds only (results WRONG): what the step costs without the folds in backward
  if (n > 0) return;
#endif
  if ((n & 3) || !salun_aligned16(part)) {
    hipLaunchKernelGGL(conv_wgrad_reduce_scalar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, dw, n,
                       nsplit, accumulate, KC, RS);
    return;
  }
#ifndef SALUN_WGRAD_REDUCE_ROWS
#define SALUN_WGRAD_REDUCE_ROWS 1  // 0: lab builds without the whole-row variant
#endif
  // whole-row variant: 3x3, at least 1,024 blocks of 32 pairs (below that the blocks do not cover the chip and the
  // plain kernel's finer split of the work wins)
  if (SALUN_WGRAD_REDUCE_ROWS && RS == 9 && (KC & 31) == 0 && KC >= 32768) {
    hipLaunchKernelGGL(conv_wgrad_reduce_rows, dim3((unsigned)(KC / 32)), dim3(256), 0, st, part, dw, (int64_t)KC, nsplit,
                       accumulate);
    return;
  }
  int logg = 0;
  while (logg < 4 && (2 << logg) <= nsplit) ++logg;
  const int outs = 256 >> logg;
  const unsigned grid = (unsigned)(((n >> 2) + outs - 1) / outs);
  switch (logg) {
    case 0: hipLaunchKernelGGL(conv_wgrad_reduce<0>, dim3(grid), dim3(256), 0, st, part, dw, n, nsplit, accumulate, KC, RS); break;
    case 1: hipLaunchKernelGGL(conv_wgrad_reduce<1>, dim3(grid), dim3(256), 0, st, part, dw, n, nsplit, accumulate, KC, RS); break;
    case 2: hipLaunchKernelGGL(conv_wgrad_reduce<2>, dim3(grid), dim3(256), 0, st, part, dw, n, nsplit, accumulate, KC, RS); break;
    case 3: hipLaunchKernelGGL(conv_wgrad_reduce<3>, dim3(grid), dim3(256), 0, st, part, dw, n, nsplit, accumulate, KC, RS); break;
    default: hipLaunchKernelGGL(conv_wgrad_reduce<4>, dim3(grid), dim3(256), 0, st, part, dw, n, nsplit, accumulate, KC, RS); break;
  }
}

// ---------------------------------------------------------------------------------------------------
struct TileGeom {
  int NI, TP, IH_t, IW_t, logQ;
  int ntiles;
  bool ok;
};

// dynamic LDS above the default limit needs a per-kernel opt-in; the attribute call is a slow driver round trip,
// so it is made once per kernel (process-wide table, the only mutable global state of the library).
inline void allow_lds_once(const void *fn) {
  static std::mutex mu;
  static std::unordered_set<uintptr_t> done;  // (kernel, device): the opt-in is per device
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert(reinterpret_cast<uintptr_t>(fn) ^ ((uintptr_t)(salun_device_bit() + 1) << 56)).second)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <typename F>
inline void allow_lds(F fn, size_t bytes) {
  if (bytes > 48 * 1024) allow_lds_once(reinterpret_cast<const void *>(fn));
}

inline int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// pixel tiling of an output of N x P x Q into tiles of `pixt` pixels; conv stride cs, filter R
inline TileGeom make_geom(int N, int P, int Q, int pixt, int cs, int R) {
  TileGeom g{};
  g.ok = false;
  if (Q <= 0 || (Q & (Q - 1)) != 0 || Q > pixt) return g;  // Q power of two, at most one tile wide
  g.logQ = ilog2(Q);
  const int PQ = P * Q;
  if (PQ >= pixt) {
    if (PQ % pixt != 0) return g;
    g.NI = 1;
    g.TP = pixt / Q;
    g.ntiles = N * (PQ / pixt);
  } else {
    if (pixt % PQ != 0) return g;
    g.NI = pixt / PQ;
    g.TP = P;
    g.ntiles = (N + g.NI - 1) / g.NI;
  }
  g.IH_t = (g.TP - 1) * cs + R;
  g.IW_t = (Q - 1) * cs + R;
  g.ok = (g.NI * g.IH_t * g.IW_t) <= 3 * 256;
  return g;
}

// Second half of a reduction-split launch: y = sum_z part[z] (fixed order) + the epilogue terms the split workgroups
// left out.  `full` is the backward-data addend (same shape as y), or null.
__global__ __launch_bounds__(256) void conv_split_finish(const float *__restrict__ part, int S, size_t out_elems,
                                                         const float *__restrict__ bias, const float *__restrict__ nbias,
                                                         const float *__restrict__ addend, const float *full,
                                                         float *y, int yC, int HW) {
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t e = i4 * 4;
  if (e >= out_elems) return;
  float4 acc = *reinterpret_cast<const float4 *>(part + e);
  for (int z = 1; z < S; ++z) {
    const float4 p = *reinterpret_cast<const float4 *>(part + (size_t)z * out_elems + e);
    acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
  }
  const size_t nk = e / (size_t)HW;  // HW % 4 == 0: the four elements share (n, k)
  if (bias) { const float b = bias[nk % (size_t)yC]; acc.x += b; acc.y += b; acc.z += b; acc.w += b; }
  if (nbias) { const float b = nbias[nk]; acc.x += b; acc.y += b; acc.z += b; acc.w += b; }
  if (addend) { const float4 a = *reinterpret_cast<const float4 *>(addend + e); acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  if (full) { const float4 a = *reinterpret_cast<const float4 *>(full + e); acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  *reinterpret_cast<float4 *>(y + e) = acc;
}

// Reduction split for under-filled launches (the 4x4 level of the DDPM U-Net at batch 128 is 128 workgroups of
// 64 pixels x 64 channels: half the CUs idle, the rest with one workgroup and nothing to hide its staging behind):
// S workgroups per output tile, each over C/S reduction channels, partial images in the caller's workspace.
inline int igemm_split(int wgs, int xC, int chunk, size_t out_elems, int HW, const void *ws, size_t ws_bytes) {
  if (!ws || wgs > 256 || HW % 4 != 0 || out_elems % 4 != 0) return 1;
  int S = 512 / (wgs < 1 ? 1 : wgs);
  if (S > 8) S = 8;
  while (S > 1 && (xC % (S * chunk) != 0 || xC / S < 4 * chunk)) --S;  // equal shares, >= 4 chunks each
  while (S > 1 && (size_t)S * out_elems * sizeof(float) > ws_bytes) --S;
  return S < 1 ? 1 : S;
}

template <int R, int STRIDE, bool DGRAD>
int launch_igemm(const float *x, const float *w, const float *bias, float *y, int N, int xC, int xH, int xW, int yC,
                 int yH, int yW, int pad, int wC, int wK, hipStream_t st, const float *nbias = nullptr,
                 const float *addend = nullptr, void *ws = nullptr, size_t ws_bytes = 0) {
  // tile shape: prefer 128 pixels x up to 128 channels; shrink when that leaves the chip under-filled
  const int cs = DGRAD ? 1 : STRIDE;
  int pixt = 128;
  TileGeom g = make_geom(N, yH, yW, pixt, cs, R);
  const int kblocks128 = (yC + 127) / 128;
  if (!g.ok || g.ntiles * kblocks128 < 384) {
    TileGeom g64 = make_geom(N, yH, yW, 64, cs, R);
    if (g64.ok) { g = g64; pixt = 64; }
  }
  if (!g.ok) return SALUN_EINVAL;
  const int PSZ = g.NI * g.IH_t * g.IW_t;
  const int ch_stride = PSZ | 1;
  constexpr int RS = R * R;
  constexpr int CC = igemm_chunk(R, STRIDE);  // the kernel's chunk (shadows the file-wide 8)
  const size_t out_elems = (size_t)N * yC * yH * yW;
  float *part = static_cast<float *>(ws);
  const bool epi = !DGRAD && (nbias != nullptr || addend != nullptr);
  const bool split_al16 = salun_aligned16(y) && salun_aligned16(ws) && (!addend || salun_aligned16(addend)) &&
                          (!(DGRAD && bias) || salun_aligned16(bias));  // DGRAD: `bias` carries the full-size addend
#define SALUN_IGEMM(KT_, WP_, WK_, SPL_)                                                                        \
  {                                                                                                             \
    constexpr int KB = WK_ * KT_ * 32;                                                                          \
    const size_t ldsb = sizeof(float) * ((size_t)CC * ch_stride + (size_t)KB * (CC * RS + 1));                  \
    dim3 grid(g.ntiles, (yC + KB - 1) / KB);                                                                    \
    /* FAST staging: full reduction chunks, channel tile inside the tensor, float4-aligned weight runs */      \
    const bool fast = (xC % CC == 0) && (!DGRAD || yC % KB == 0) && salun_aligned16(w) &&                       \
                      ((wC * RS) % 4 == 0) && !(DGRAD && STRIDE > 1);                                           \
    if (epi && (!fast || DGRAD || STRIDE != 1)) return SALUN_EINVAL; /* epilogue terms: stride-1 forward, FAST */ \
    if (fast) {                                                                                                 \
      int S = 1;                                                                                                \
      /* the finishing kernel reads / writes y, the addends and the partials as float4: 16-byte bases or no split */ \
      if constexpr (SPL_)                                                                                       \
        if (split_al16) S = igemm_split((int)(grid.x * grid.y), xC, CC, out_elems, yH * yW, ws, ws_bytes);      \
      if (S > 1) {                                                                                              \
        if constexpr (SPL_) {                                                                                   \
          grid.z = S;                                                                                           \
          allow_lds(conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true, 1, true>, ldsb);                          \
          hipLaunchKernelGGL((conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true, 1, true>), grid, dim3(256), ldsb, \
                             st, x, w, nullptr, part, N, xC, xH, xW, yC, yH, yW, pad, g.NI, g.TP, g.IH_t,       \
                             g.IW_t, g.logQ, wC, wK, nullptr, nullptr, xC / S);                                 \
          SALUN_LAUNCH_CHECK();                                                                                 \
          hipLaunchKernelGGL(conv_split_finish, dim3((unsigned)((out_elems / 4 + 255) / 256)), dim3(256), 0,    \
                             st, part, S, out_elems, DGRAD ? nullptr : bias, nbias, addend,                     \
                             DGRAD ? bias : nullptr, y, yC, yH * yW);                                           \
        }                                                                                                       \
      } else if (epi) {                                                                                         \
        if constexpr (!DGRAD && STRIDE == 1) {                                                                  \
          allow_lds(conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true, 1, false, true>, ldsb);                   \
          hipLaunchKernelGGL((conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true, 1, false, true>), grid,         \
                             dim3(256), ldsb, st, x, w, bias, y, N, xC, xH, xW, yC, yH, yW, pad, g.NI, g.TP,    \
                             g.IH_t, g.IW_t, g.logQ, wC, wK, nbias, addend, 0);                                 \
        }                                                                                                       \
      } else {                                                                                                  \
        allow_lds(conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true>, ldsb);                                     \
        hipLaunchKernelGGL((conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, true>), grid, dim3(256), ldsb, st, x, w, \
                           bias, y, N, xC, xH, xW, yC, yH, yW, pad, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, wC, wK, \
                           nullptr, nullptr, 0);                                                                \
      }                                                                                                         \
    } else {                                                                                                    \
      allow_lds(conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, false>, ldsb);                                      \
      hipLaunchKernelGGL((conv_igemm<R, STRIDE, KT_, WP_, WK_, DGRAD, false>), grid, dim3(256), ldsb, st, x, w, \
                         bias, y, N, xC, xH, xW, yC, yH, yW, pad, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, wC, wK,   \
                         nullptr, nullptr, 0);                                                                  \
    }                                                                                                           \
  }
  if (pixt == 128) {
    if (yC > 64) SALUN_IGEMM(4, 4, 1, false)
    else if (yC > 32) {
      // 64 output channels: a wave takes 64 pixels x 64 channels (PT = 2) when 256-pixel tiles still fill the chip
      // and the stride-1 fast path applies
      TileGeom g256 = make_geom(N, yH, yW, 256, cs, R);
      const int ps256 = g256.ok ? g256.NI * g256.IH_t * g256.IW_t : 0;
      const bool fast2 = (xC % CC == 0) && (!DGRAD || yC % 64 == 0) && salun_aligned16(w) && ((wC * RS) % 4 == 0) &&
                         !(DGRAD && STRIDE > 1);
      if (g256.ok && fast2 && g256.ntiles >= 512 && ps256 <= 768 && !epi) {
        const int chs = ps256 | 1;
        const size_t ldsb2 = sizeof(float) * ((size_t)CC * chs + (size_t)64 * (CC * RS + 1));
        dim3 grid2(g256.ntiles, (yC + 63) / 64);
        allow_lds(conv_igemm<R, STRIDE, 2, 4, 1, DGRAD, true, 2>, ldsb2);
        hipLaunchKernelGGL((conv_igemm<R, STRIDE, 2, 4, 1, DGRAD, true, 2>), grid2, dim3(256), ldsb2, st, x, w, bias, y,
                           N, xC, xH, xW, yC, yH, yW, pad, g256.NI, g256.TP, g256.IH_t, g256.IW_t, g256.logQ, wC, wK, nullptr,
                           nullptr, 0);
      } else SALUN_IGEMM(2, 4, 1, false)
    }
    else SALUN_IGEMM(1, 4, 1, false)
  } else {
    // small pixel space (deep layers): 64 x 128 tiles only if that still yields enough workgroups
    if (yC > 64 && g.ntiles * kblocks128 >= 384) SALUN_IGEMM(2, 2, 2, false)
    else SALUN_IGEMM(1, 2, 2, true)  // the only tiling an under-filled launch ends up with
  }
#undef SALUN_IGEMM
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}


// rectangular-tap variant of make_geom (RH rows x RW cols of taps)
inline TileGeom make_geom_rect(int N, int P, int Q, int pixt, int cs, int RH, int RW) {
  TileGeom g = make_geom(N, P, Q, pixt, cs, RH);
  if (!g.ok && g.NI == 0) return g;
  g.IW_t = (Q - 1) * cs + RW;
  g.ok = (g.NI > 0) && (g.NI * g.IH_t * g.IW_t) <= 3 * 256;
  return g;
}

template <int RH, int RW, int STRIDE, bool DGRAD>
int launch_igemm_tap(IgemmArgs a, hipStream_t st) {
  // tile shape: prefer 128 pixels x up to 128 channels; shrink when that leaves the chip under-filled
  const int cs = DGRAD ? 1 : STRIDE;
  int pixt = 128;
  TileGeom g = make_geom_rect(a.N, a.subH, a.subW, pixt, cs, RH, RW);
  const int kblocks128 = (a.yC + 127) / 128;
  if (!g.ok || g.ntiles * kblocks128 < 384) {
    TileGeom g64 = make_geom_rect(a.N, a.subH, a.subW, 64, cs, RH, RW);
    if (g64.ok) { g = g64; pixt = 64; }
  }
  if (!g.ok) return SALUN_EINVAL;
  a.NI = g.NI; a.TP = g.TP; a.IH_t = g.IH_t; a.IW_t = g.IW_t; a.logQ = g.logQ;
  const int PSZ = g.NI * g.IH_t * g.IW_t;
  const int ch_stride = PSZ | 1;
  constexpr int RS = RH * RW;
#define SALUN_IGEMM(KT_, WP_, WK_)                                                                     \
  {                                                                                                    \
    constexpr int KB = WK_ * KT_ * 32;                                                                 \
    constexpr int CCT = tap_chunk(RS);                                                                 \
    const size_t ldsb = sizeof(float) * ((size_t)CCT * ch_stride + (size_t)KB * (CCT * RS + 1));       \
    dim3 grid(g.ntiles, (a.yC + KB - 1) / KB);                                                         \
    if (PSZ <= 256) {                                                                                  \
      allow_lds(conv_igemm_tap<RH, RW, STRIDE, KT_, WP_, WK_, DGRAD, 1>, ldsb);                        \
      hipLaunchKernelGGL((conv_igemm_tap<RH, RW, STRIDE, KT_, WP_, WK_, DGRAD, 1>), grid, dim3(256), ldsb, st, a); \
    } else {                                                                                           \
      allow_lds(conv_igemm_tap<RH, RW, STRIDE, KT_, WP_, WK_, DGRAD, 3>, ldsb);                        \
      hipLaunchKernelGGL((conv_igemm_tap<RH, RW, STRIDE, KT_, WP_, WK_, DGRAD, 3>), grid, dim3(256), ldsb, st, a); \
    }                                                                                                  \
  }
  if (pixt == 128) {
    if (a.yC > 64) SALUN_IGEMM(4, 4, 1)
    else if (a.yC > 32) SALUN_IGEMM(2, 4, 1)
    else SALUN_IGEMM(1, 4, 1)
  } else {
    // small pixel space (deep layers): 64 x 128 tiles only if that still yields enough workgroups
    if (a.yC > 64 && g.ntiles * kblocks128 >= 384) SALUN_IGEMM(2, 2, 2)
    else SALUN_IGEMM(1, 2, 2)
  }
#undef SALUN_IGEMM
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// stride-2 backward-data, all parity classes in one launch (conv_dgrad_s2).  SALUN_EINVAL = not applicable.
template <int R>
int launch_dgrad_s2_merged(const float *dy, const float *w, const float *addend, float *dx, int N, int C, int H, int W,
                           int K, int pad, int P, int Q, hipStream_t st) {
  if (H != 2 * P || W != 2 * Q) return SALUN_EINVAL;
  if (!((R == 3 && (pad == 0 || pad == 1)) || (R == 1 && pad == 0))) return SALUN_EINVAL;
  constexpr int RS = R * R;
  if (K % CC != 0 || !salun_aligned16(w) || (C * RS) % 4 != 0) return SALUN_EINVAL;
  if (!salun_aligned16(dx) || (addend && !salun_aligned16(addend))) return SALUN_EINVAL;  // float2 row pairs
  // tiles of sub-grid pixels; channel tile 64 or 128 (4*KT accumulators per wave: KT <= 2)
  const int fp = (R == 1) ? 1 : 2;
  int pixt = 128;
  TileGeom g = make_geom(N, P, Q, pixt, 1, fp);
  const int cb64 = (C + 63) / 64;
  if (!g.ok || g.ntiles * cb64 < 256) {
    TileGeom g64 = make_geom(N, P, Q, 64, 1, fp);
    if (g64.ok) { g = g64; pixt = 64; }
  }
  if (!g.ok || g.NI * g.IH_t * g.IW_t > 256) return SALUN_EINVAL;
  const int PSZ = g.NI * g.IH_t * g.IW_t;
  const int ch_stride = PSZ | 1;
#define SALUN_DGS2(PAD_, KT_, WP_, WK_)                                                                          \
  {                                                                                                              \
    constexpr int KB = WK_ * KT_ * 32;                                                                           \
    if (C % KB != 0) return SALUN_EINVAL;                                                                        \
    const size_t ldsb = sizeof(float) * ((size_t)CC * ch_stride + (size_t)KB * (CC * RS + 1));                   \
    dim3 grid(g.ntiles, C / KB);                                                                                 \
    allow_lds(conv_dgrad_s2<R, PAD_, KT_, WP_, WK_>, ldsb);                                                      \
    hipLaunchKernelGGL((conv_dgrad_s2<R, PAD_, KT_, WP_, WK_>), grid, dim3(256), ldsb, st, dy, w, addend, dx, N, \
                       K, P, Q, C, H, W, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ);                                    \
  }
#define SALUN_DGS2_TILES(PAD_)                                                               \
  if (pixt == 128) {                                                                         \
    if (C % 64 == 0) SALUN_DGS2(PAD_, 2, 4, 1)                                               \
    else SALUN_DGS2(PAD_, 1, 4, 1)                                                           \
  } else {                                                                                   \
    if (C % 128 == 0 && g.ntiles * (C / 128) >= 256) SALUN_DGS2(PAD_, 2, 2, 2)               \
    else SALUN_DGS2(PAD_, 1, 2, 2)                                                           \
  }
  if (pad == 1) { SALUN_DGS2_TILES(1) } else { SALUN_DGS2_TILES(0) }
#undef SALUN_DGS2_TILES
#undef SALUN_DGS2
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// backward-data: one launch for stride 1, one launch per output parity class for stride 2
template <int R>
int launch_dgrad(const float *dy, const float *w, const float *addend, float *dx, int N, int C, int H, int W, int K,
                 int stride, int pad, int P, int Q, hipStream_t st, void *ws = nullptr, size_t ws_bytes = 0) {
  IgemmArgs a{};
  a.x = dy; a.w = w; a.bias = addend; a.y = dx;
  a.N = N; a.xC = K; a.xH = P; a.xW = Q; a.yC = C; a.yH = H; a.yW = W;
  a.wC = C; a.wK = K; a.Rfull = R;
  if (stride == 1) {
    a.subH = H; a.subW = W; a.os = 1; a.ph = a.pw = 0;
    a.rtop_h = a.rtop_w = R - 1; a.ts = 1;
    return launch_igemm<R, 1, true>(dy, w, addend, dx, N, K, P, Q, C, H, W, pad, C, K, st, nullptr, nullptr, ws, ws_bytes);
  }
  if ((H & 1) || (W & 1)) return SALUN_EINVAL;
  {
    const int rc = launch_dgrad_s2_merged<R>(dy, w, addend, dx, N, C, H, W, K, pad, P, Q, st);
    if (rc != SALUN_EINVAL) return rc;  // SALUN_EINVAL: outside the merged kernel's domain -> per-class path below
  }
  a.subH = H / 2; a.subW = W / 2; a.os = 2; a.ts = 2;
  // taps of parity class p: r = (p + pad) mod 2, +2, ... < R ; rtop = the largest
  int ntap[2], rtop[2];
  for (int p = 0; p < 2; ++p) {
    const int r0 = (p + pad) & 1;
    ntap[p] = (r0 < R) ? (R - 1 - r0) / 2 + 1 : 0;
    rtop[p] = r0 + 2 * (ntap[p] - 1);
  }
  bool any_empty = false;
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw)
      if (ntap[ph] == 0 || ntap[pw] == 0) any_empty = true;
  if (any_empty) {  // positions no tap reaches: 0, or the addend itself
    const size_t bytes = sizeof(float) * (size_t)N * C * H * W;
    if (!addend) {
      if (hipMemsetAsync(dx, 0, bytes, st) != hipSuccess) return SALUN_EIO;
    } else if (addend != dx) {
      if (hipMemcpyAsync(dx, addend, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return SALUN_EIO;
      a.bias = dx;  // the covered classes now accumulate in place
    }
  }
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      if (ntap[ph] == 0 || ntap[pw] == 0) continue;
      a.ph = ph; a.pw = pw;
      a.rtop_h = rtop[ph]; a.rtop_w = rtop[pw];
      a.vpad_h = -((ph + pad - rtop[ph]) / 2);  // (ph + pad - rtop) is even and <= 0
      a.vpad_w = -((pw + pad - rtop[pw]) / 2);
      int rc;
      if (ntap[ph] == 1 && ntap[pw] == 1) rc = launch_igemm_tap<1, 1, 1, true>(a, st);
      else if (ntap[ph] == 1 && ntap[pw] == 2) rc = launch_igemm_tap<1, 2, 1, true>(a, st);
      else if (ntap[ph] == 2 && ntap[pw] == 1) rc = launch_igemm_tap<2, 1, 1, true>(a, st);
      else if (ntap[ph] == 2 && ntap[pw] == 2) rc = launch_igemm_tap<2, 2, 1, true>(a, st);
      else return SALUN_EINVAL;
      if (rc != SALUN_OK) return rc;
    }
  return SALUN_OK;
}

// 1x1 backward-weight (conv_wgrad_1x1): geometry and split count; 0 chunks = shape outside its domain
inline int wgrad1x1_chunks(int N, int P, int Q, int stride) {
  const int PQ = P * Q;
  if (PQ % 4 != 0 || (stride == 2 && Q % 4 != 0)) return 0;
  if (PQ >= 64) return (PQ % 64 == 0) ? N * (PQ / 64) : 0;
  return (64 % PQ == 0) ? (N + 64 / PQ - 1) / (64 / PQ) : 0;
}
inline int wgrad1x1_nsplit(int K, int C, int nchunks) {
  const int tiles = ((K + 127) / 128) * ((C + 127) / 128);
  int ns = (256 + tiles - 1) / tiles;
  if (ns > nchunks) ns = nchunks;
  return ns < 1 ? 1 : ns;
}

inline int wgrad_nsplit(int K, int C, int nchunks) {
  const int tiles = ((K + 63) / 64) * ((C + 63) / 64);
  // one workgroup per CU is resident (register budget): a single full round of 256 workgroups, each with a long
  // run of chunks, beats more and shorter splits (prologue/epilogue and partial-sum traffic scale with nsplit)
  // (round 5 sweep on the ResNet-18 step, profiles/r05_wgrad_sweep.txt: 128 / 192 / 256 / 384 / 512 target workgroups
  // -> 94.4 / 107.3 / 114.7 / 103.4 / 106.1 steps/s)
  int ns = (256 + tiles - 1) / tiles;
  if (ns > nchunks) ns = nchunks;
  if (ns < 1) ns = 1;
  return ns;
}

// workgroups launch_igemm would start for an output of [N, outC, outH, outW] (mirrors its tile choice); -1: outside
// the tiling domain
inline int igemm_wgs(int N, int outC, int outH, int outW, int cs, int R) {
  TileGeom g = make_geom(N, outH, outW, 128, cs, R);
  const int kblocks128 = (outC + 127) / 128;
  int pixt = 128;
  if (!g.ok || g.ntiles * kblocks128 < 384) {
    TileGeom g64 = make_geom(N, outH, outW, 64, cs, R);
    if (g64.ok) { g = g64; pixt = 64; }
  }
  if (!g.ok) return -1;
  int KB;
  if (pixt == 128) KB = outC > 64 ? 128 : outC > 32 ? 64 : 32;
  else KB = (outC > 64 && g.ntiles * kblocks128 >= 384) ? 128 : 64;
  return g.ntiles * ((outC + KB - 1) / KB);
}

}  // namespace

// ================================================================== C-ABI =======
// Scratch for the reduction split of under-filled forward / stride-1 backward-data launches (<= 256 workgroups): up to
// eight partial images of the OUTPUT tensor [N, outC, outH, outW] of a convolution with filter size R; `conv_stride`
// is the forward stride for the forward pass and 1 for backward-data.  0 when the launch would not be split.
SALUN_EXPORT size_t salun_conv2d_data_workspace_bytes(int N, int outC, int outH, int outW, int R, int conv_stride) {
  if (N < 1 || outC < 1 || outH < 1 || outW < 1 || (R != 1 && R != 3) || (conv_stride != 1 && conv_stride != 2)) return 0;
  const int wgs = igemm_wgs(N, outC, outH, outW, conv_stride, R);
  if (wgs < 0 || wgs > 256 || (outH * outW) % 4 != 0) return 0;
  return (size_t)8 * N * outC * outH * outW * sizeof(float);
}

// y[N,K,P,Q] = conv2d(x[N,C,H,W], w[K,C,R,R], stride, pad_lo) (+ bias[K]) (+ nbias[N,K]) (+ addend[N,K,P,Q]); P,Q given
// by the caller (so asymmetric high-side padding is expressed through P,Q).  Returns SALUN_EINVAL for shapes outside the
// tiling's domain (Q not a power of two, ...): the caller then uses the library convolution.
SALUN_EXPORT int salun_conv2d_forward_fused(const float *x, const float *w, const float *bias, const float *nbias,
                                            const float *addend, float *y, int N, int C, int H, int W, int K, int R,
                                            int stride, int pad, int P, int Q, void *ws, size_t ws_bytes,
                                            salun_stream_t stream) {
  if (!x || !w || !y || N < 1 || C < 1 || K < 1 || P < 1 || Q < 1 || addend == y) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  if (R == 3 && stride == 1)
    return launch_igemm<3, 1, false>(x, w, bias, y, N, C, H, W, K, P, Q, pad, C, K, st, nbias, addend, ws, ws_bytes);
  if (R == 3 && stride == 2)
    return launch_igemm<3, 2, false>(x, w, bias, y, N, C, H, W, K, P, Q, pad, C, K, st, nbias, addend, ws, ws_bytes);
  if (R == 1 && stride == 1)
    return launch_igemm<1, 1, false>(x, w, bias, y, N, C, H, W, K, P, Q, pad, C, K, st, nbias, addend, ws, ws_bytes);
  if (R == 1 && stride == 2)
    return launch_igemm<1, 2, false>(x, w, bias, y, N, C, H, W, K, P, Q, pad, C, K, st, nbias, addend, ws, ws_bytes);
  return SALUN_EINVAL;
}

SALUN_EXPORT int salun_conv2d_forward(const float *x, const float *w, const float *bias, float *y, int N, int C,
                                      int H, int W, int K, int R, int stride, int pad, int P, int Q,
                                      salun_stream_t stream) {
  return salun_conv2d_forward_fused(x, w, bias, nullptr, nullptr, y, N, C, H, W, K, R, stride, pad, P, Q, nullptr, 0,
                                    stream);
}

// dx[N,C,H,W] = conv2d_backward_data(dy[N,K,P,Q], w[K,C,R,R]) (+ addend[N,C,H,W]; addend == dx accumulates in place)
SALUN_EXPORT int salun_conv2d_backward_data_ws(const float *dy, const float *w, const float *addend, float *dx, int N,
                                               int C, int H, int W, int K, int R, int stride, int pad, int P, int Q,
                                               void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!dy || !w || !dx || N < 1 || C < 1 || K < 1 || P < 1 || Q < 1) return SALUN_EINVAL;
  if (stride != 1 && stride != 2) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  if (R == 3) return launch_dgrad<3>(dy, w, addend, dx, N, C, H, W, K, stride, pad, P, Q, st, ws, ws_bytes);
  if (R == 1) return launch_dgrad<1>(dy, w, addend, dx, N, C, H, W, K, stride, pad, P, Q, st, ws, ws_bytes);
  return SALUN_EINVAL;
}

SALUN_EXPORT int salun_conv2d_backward_data_add(const float *dy, const float *w, const float *addend, float *dx, int N,
                                                int C, int H, int W, int K, int R, int stride, int pad, int P, int Q,
                                                salun_stream_t stream) {
  return salun_conv2d_backward_data_ws(dy, w, addend, dx, N, C, H, W, K, R, stride, pad, P, Q, nullptr, 0, stream);
}

SALUN_EXPORT int salun_conv2d_backward_data(const float *dy, const float *w, float *dx, int N, int C, int H, int W,
                                            int K, int R, int stride, int pad, int P, int Q,
                                            salun_stream_t stream) {
  return salun_conv2d_backward_data_add(dy, w, nullptr, dx, N, C, H, W, K, R, stride, pad, P, Q, stream);
}

SALUN_EXPORT size_t salun_conv2d_wgrad_workspace_bytes(int N, int C, int K, int R, int P, int Q) {
  // upper bound over both chunk sizes (64 pixels for stride 1, 32 for stride 2)
  TileGeom g = make_geom(N, P, Q, 32, 1, R);
  if (!g.ok) g = make_geom(N, P, Q, 64, 1, R);
  if (!g.ok) return 0;
  int ns = wgrad_nsplit(K, C, g.ntiles);
  if (C * R * R <= 32 && C <= 4) {  // small-C kernel: up to 1024 workgroups x 2 pixel halves
    int nss = 1024 / ((K + 63) / 64);
    TileGeom g64 = make_geom(N, P, Q, 64, 1, R);
    const int nt = g64.ok && g64.ntiles > g.ntiles ? g64.ntiles : g.ntiles;
    if (nss > nt) nss = nt;
    if (nss * 2 > ns) ns = nss * 2;
  }
  if (R == 1) {  // conv_wgrad_1x1 splits deeper (128 x 128 tiles): size for whichever kernel the shape gets
    const int nc1 = wgrad1x1_chunks(N, P, Q, 1);
    if (nc1 > 0) { const int ns1 = wgrad1x1_nsplit(K, C, nc1); if (ns1 > ns) ns = ns1; }
  }
  return sizeof(float) * (size_t)ns * K * C * R * R;
}

// ---------------------------------------------------------------------------------------------------
// bias gradient of a convolution: out[k] (= or +=) sum_{n,p,q} dy[n][k][p][q] — one streaming pass over dy (4 B per
// element) in place of ATen's generic `sum(dim=(0,2,3))` + the AccumulateGrad add (3.7 % of the DDPM step in round 3's
// profile).  grid (K, S): workgroup (k, s) walks images s, s+S, ... of channel k with 16-byte loads, folds its 256
// partial sums in a fixed tree, writes part[k][s]; a second small launch adds the S partials in order.
__global__ __launch_bounds__(256) void k_channel_sum_partial(const float *__restrict__ dy, float *__restrict__ part,
                                                             int N, int K, int HW, int S) {
  __shared__ float red[256];
  const int k = blockIdx.x, sp = blockIdx.y, tid = threadIdx.x;
  float acc = 0.f;
  const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
  for (int n = sp; n < N; n += S) {
    const float *src = dy + ((size_t)n * K + k) * HW;
    if (vec) {
      const float4 *s4 = reinterpret_cast<const float4 *>(src);
      for (int i = tid; i < (HW >> 2); i += 256) {
        const float4 v = ldg_stream4(s4 + i);
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      for (int i = tid; i < HW; i += 256) acc += src[i];
    }
  }
  red[tid] = acc;
  __syncthreads();
#pragma unroll
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0) part[(size_t)k * S + sp] = red[0];
}
__global__ __launch_bounds__(256) void k_channel_sum_final(const float *__restrict__ part, float *__restrict__ out, int K,
                                                           int S, int accumulate) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  float t = 0.f;
  for (int s2 = 0; s2 < S; ++s2) t += part[(size_t)k * S + s2];
  out[k] = accumulate ? out[k] + t : t;
}
inline int channel_sum_splits(int N, int K) {
  int S = 2048 / (K > 0 ? K : 1);
  if (S < 1) S = 1;
  if (S > N) S = N;
  if (S > 256) S = 256;
  return S;
}

SALUN_EXPORT size_t salun_channel_sum_workspace_bytes(int N, int K) {
  if (N < 1 || K < 1) return 0;
  return sizeof(float) * (size_t)K * channel_sum_splits(N, K);
}

SALUN_EXPORT int salun_channel_sum(const float *dy, float *out, int N, int K, int HW, int accumulate, void *ws,
                                   size_t ws_bytes, salun_stream_t stream) {
  if (!dy || !out || !ws || N < 1 || K < 1 || HW < 1) return SALUN_EINVAL;
  const int S = channel_sum_splits(N, K);
  if (ws_bytes < sizeof(float) * (size_t)K * S) return SALUN_ENOSPC;
  if (K > 65535 * 16) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  float *part = static_cast<float *>(ws);
  hipLaunchKernelGGL(k_channel_sum_partial, dim3(K, S), dim3(256), 0, st, dy, part, N, K, HW, S);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_channel_sum_final, dim3((K + 255) / 256), dim3(256), 0, st, part, out, K, S, accumulate);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// dw[K,C,R,R] (= or +=) conv2d_backward_weight(x[N,C,H,W], dy[N,K,P,Q])
SALUN_EXPORT int salun_conv2d_backward_weight(const float *x, const float *dy, float *dw, int N, int C, int H, int W,
                                              int K, int R, int stride, int pad, int P, int Q, int accumulate,
                                              void *ws, size_t ws_bytes, salun_stream_t stream) {
  return salun_conv2d_backward_weight_ex(x, dy, dw, N, C, H, W, K, R, stride, pad, P, Q, accumulate, 0, ws, ws_bytes, stream);
}

SALUN_EXPORT int salun_conv2d_backward_weight_ex(const float *x, const float *dy, float *dw, int N, int C, int H, int W,
                                                 int K, int R, int stride, int pad, int P, int Q, int accumulate,
                                                 unsigned flags, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!x || !dy || !dw || !ws || N < 1 || C < 1 || K < 1 || P < 1 || Q < 1) return SALUN_EINVAL;
  if (!((R == 3 || R == 1) && (stride == 1 || stride == 2))) return SALUN_EINVAL;
  if (R == 1 && pad == 0 && C >= 32 && H == P * stride && W == Q * stride && W % 4 == 0 && salun_aligned16(x) &&
      salun_aligned16(dy) && (size_t)64 * C * H * W < (1u << 30) && (size_t)64 * K * P * Q < (1u << 30)) {
    const int nc1 = wgrad1x1_chunks(N, P, Q, stride);
    if (nc1 > 0) {  // GEMM-shaped 1x1 kernel
      hipStream_t st1 = salun_hip_stream(stream);
      const int ns1 = wgrad1x1_nsplit(K, C, nc1);
      if (ws_bytes < sizeof(float) * (size_t)ns1 * K * C) return SALUN_ENOSPC;
      const size_t lds1 = sizeof(float) * 2 * 2 * 128 * 65;
      dim3 grid1((K + 127) / 128, (C + 127) / 128, ns1);
      float *part1 = static_cast<float *>(ws);
      if (stride == 1) {
        allow_lds(conv_wgrad_1x1<1>, lds1);
        hipLaunchKernelGGL(conv_wgrad_1x1<1>, grid1, dim3(256), lds1, st1, x, dy, part1, N, C, H, W, K, P, Q, nc1);
      } else {
        allow_lds(conv_wgrad_1x1<2>, lds1);
        hipLaunchKernelGGL(conv_wgrad_1x1<2>, grid1, dim3(256), lds1, st1, x, dy, part1, N, C, H, W, K, P, Q, nc1);
      }
      SALUN_LAUNCH_CHECK();
      launch_wgrad_reduce(part1, dw, (int64_t)K * C, ns1, accumulate, K * C, 1, st1);
      SALUN_LAUNCH_CHECK();
      return SALUN_OK;
    }
  }
  const int pixc = (stride == 1) ? 64 : 32;
  TileGeom g = make_geom(N, P, Q, pixc, stride, R);
  if (!g.ok || g.NI * g.IH_t * g.IW_t > 256) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  if (C * R * R <= 32 && C <= 4 && Q >= 4 && (g.TP & (g.TP - 1)) == 0) {  // RGB stem: columns = (c, r, s)
    int ns = 1024 / ((K + 63) / 64);
    if (ns > g.ntiles) ns = g.ntiles;
    if (ns < 1) ns = 1;
    const size_t need_s = sizeof(float) * (size_t)ns * 2 * K * C * R * R;
    if (ws_bytes < need_s) return SALUN_ENOSPC;
    const int PSZs = g.NI * g.IH_t * g.IW_t;
    const size_t ldss = sizeof(float) * ((size_t)C * (PSZs | 1) + (size_t)64 * (pixc + 1));
    dim3 grid_s((K + 63) / 64, 1, ns);
    float *part_s = static_cast<float *>(ws);
#define SALUN_WGRAD_S(R_, S_)                                                                                  \
  hipLaunchKernelGGL((conv_wgrad_smallc<R_, S_>), grid_s, dim3(256), ldss, st, x, dy, part_s, N, C, H, W, K, P, \
                     Q, pad, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, g.ntiles)
    if (R == 3 && stride == 1) SALUN_WGRAD_S(3, 1);
    else if (R == 3 && stride == 2) SALUN_WGRAD_S(3, 2);
    else if (R == 1 && stride == 1) SALUN_WGRAD_S(1, 1);
    else SALUN_WGRAD_S(1, 2);
#undef SALUN_WGRAD_S
    SALUN_LAUNCH_CHECK();
    const int64_t nn = (int64_t)K * C * R * R;
    launch_wgrad_reduce(part_s, dw, nn, ns * 2, accumulate, K * C, R * R, st);
    SALUN_LAUNCH_CHECK();
    return SALUN_OK;
  }
  const int ns = wgrad_nsplit(K, C, g.ntiles);
  const size_t need = sizeof(float) * (size_t)ns * K * C * R * R;
  if (ws_bytes < need) return SALUN_ENOSPC;
  const int PSZ = g.NI * g.IH_t * g.IW_t;
  if (Q < 4 || (g.TP & (g.TP - 1)) != 0) return SALUN_EINVAL;  // float4 dy staging; shift-only pixel decoding
  const size_t ldsb = sizeof(float) * 2 * ((size_t)64 * (PSZ | 1) + (size_t)64 * (pixc + 1) + 256);  // double buffered
  // per-lane offsets inside one chunk's image block are 32-bit
  if ((size_t)g.NI * C * H * W >= (1u << 30) || (size_t)g.NI * K * P * Q >= (1u << 30)) return SALUN_EINVAL;
  if (ldsb > 160 * 1024) return SALUN_EINVAL;
  dim3 grid((K + 63) / 64, (C + 63) / 64, ns);
  float *part = static_cast<float *>(ws);
  // 3x3 / stride 1 / pad 1 on square images: the LDS-DMA ring kernel (salun_conv_ring.hip)
  bool vec_done = false;
  // ... unless the caller says the launch shares the device with another stream's kernels (SALUN_WGRAD_SHARED): beside a
  // second matrix kernel or a BatchNorm pass on the same SIMDs the ring kernel's dense MFMA stream buys nothing and costs
  // its neighbours more than conv_wgrad_v's does (round 6: ResNet-18 step 117.5 vs 116.1 steps/s, DDPM 9.59 vs 9.46 with
  // backward-weight on the side stream; alone it is 12 - 18 % faster, and the single-stream step 111.8 vs 107.9)
#if !SALUN_WGRAD_NO_RING
  if (R == 3 && stride == 1 && pad == 1 && P == H && Q == W && pixc == 64 && !(flags & SALUN_WGRAD_SHARED)) {
    const int rc = salun_ring_wgrad_launch(x, dy, part, N, C, H, W, K, ns, g.ntiles, st);
    if (rc == SALUN_OK) vec_done = true;
    else if (rc != SALUN_EINVAL) return rc;
  }
#endif
  // vectorised staging (conv_wgrad_v) where the geometry allows it: 3x3, pad 1, full channel tiles, whole image rows
  if (!vec_done && R == 3 && pad == 1 && C % 64 == 0 && W % 4 == 0 && Q * stride == W && P * stride == H && salun_aligned16(x) &&
      salun_aligned16(dy) && (size_t)g.NI * C * H * W < (1u << 30)) {
    const int F4C = g.NI * g.IH_t * (W / 4);
    const int nit = (F4C % 4 == 0) ? F4C / 4 : 0;
#define SALUN_WGRAD_V(S_, NIT_)                                                                                \
  {                                                                                                            \
    allow_lds(conv_wgrad_v<S_, NIT_>, ldsb);                                                                   \
    hipLaunchKernelGGL((conv_wgrad_v<S_, NIT_>), grid, dim3(256), ldsb, st, x, dy, part, N, C, H, W, K, P, Q,  \
                       g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, g.ntiles);                                          \
    vec_done = true;                                                                                           \
  }
#define SALUN_WGRAD_W(S_, NIT_)                                                                                \
  {                                                                                                            \
    allow_lds(conv_wgrad_w<S_, NIT_>, ldsw);                                                                   \
    hipLaunchKernelGGL((conv_wgrad_w<S_, NIT_>), grid, dim3(512), ldsw, st, x, dy, part, N, C, H, W, K, P, Q,  \
                       g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, g.ntiles);                                          \
    vec_done = true;                                                                                           \
  }
    // eight-wave variant (two waves per SIMD): NIT = ceil(64 * F4C / 512) items per thread
    const int nitw = (64 * F4C + 511) / 512;
    const size_t ldsw = sizeof(float) * 2 * ((size_t)64 * (PSZ | 1) + (size_t)64 * (pixc + 1) + 512);
    if (SALUN_WGRAD_8WAVES && ldsw <= 160 * 1024) {
      if (stride == 1) {
        if (nitw == 3) SALUN_WGRAD_W(1, 3)
        else if (nitw == 4) SALUN_WGRAD_W(1, 4)
      } else {
        if (nitw == 5) SALUN_WGRAD_W(2, 5)
      }
    }
#undef SALUN_WGRAD_W
    if (vec_done) { }
    else if (stride == 1) {
      if (nit == 5) SALUN_WGRAD_V(1, 5)
      else if (nit == 6) SALUN_WGRAD_V(1, 6)
      else if (nit == 8) SALUN_WGRAD_V(1, 8)
    } else {
      if (nit == 9) SALUN_WGRAD_V(2, 9)
      else if (nit == 10) SALUN_WGRAD_V(2, 10)
    }
#undef SALUN_WGRAD_V
  }
#define SALUN_WGRAD(R_, S_)                                                                                    \
  if (C % 64 == 0) {                                                                                           \
    allow_lds(conv_wgrad<R_, S_, true>, ldsb);                                                                 \
    hipLaunchKernelGGL((conv_wgrad<R_, S_, true>), grid, dim3(256), ldsb, st, x, dy, part, N, C, H, W, K, P,   \
                       Q, pad, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, g.ntiles);                                  \
  } else {                                                                                                     \
    allow_lds(conv_wgrad<R_, S_, false>, ldsb);                                                                \
    hipLaunchKernelGGL((conv_wgrad<R_, S_, false>), grid, dim3(256), ldsb, st, x, dy, part, N, C, H, W, K, P,  \
                       Q, pad, g.NI, g.TP, g.IH_t, g.IW_t, g.logQ, g.ntiles);                                  \
  }
  if (vec_done) { }
  else if (R == 3 && stride == 1) { SALUN_WGRAD(3, 1) }
  else if (R == 3 && stride == 2) { SALUN_WGRAD(3, 2) }
  else if (R == 1 && stride == 1) { SALUN_WGRAD(1, 1) }
  else { SALUN_WGRAD(1, 2) }
#undef SALUN_WGRAD
  SALUN_LAUNCH_CHECK();
  const int64_t n = (int64_t)K * C * R * R;
  launch_wgrad_reduce(part, dw, n, ns, accumulate, K * C, R * R, st);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
