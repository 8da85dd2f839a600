// salun_conv_ring.hip — K8r: fp32 3x3 / stride 1 / pad 1 convolution, forward and backward-data, on the CDNA4 matrix
// cores (v_mfma_f32_32x32x2_f32) with its operands fed by LDS-DMA into a two-stage LDS ring (round 6).
//
// Why a second kernel beside conv_igemm (salun_conv.hip): that kernel stages every 8-channel chunk through registers
// between two barriers — 16-24 global loads + as many ds_write_b32 per thread and chunk for the patch, 20-36 more for the
// weight slab, LDS rows with odd strides so that the dword operand reads do not collide.  Rounds 3-5 measured what that
// costs (profiles/r03_igemm_ab.txt: 8-15 us of loads, 2-17 us of LDS stores and 14 us of per-workgroup prologue /
// epilogue on a 160-190 us layer) and that vector-memory and LDS-store INSTRUCTIONS, not bytes, are the scarce resource
// beside the matrix pipe.  Here
//   * nothing is staged through registers: both operands go global -> LDS with `global_load_lds_dwordx4` (1 KB per wave
//     instruction), chunk i+1 in flight while chunk i is multiplied, ONE raw s_barrier per chunk, counted by vmcnt;
//   * the weights come from a pre-packed image (k_ring_pack, once per optimizer step and layer): for a 32-row tile and a
//     chunk the image holds, per lane, the A operands of four consecutive k-steps as one 16-byte piece — one
//     conflict-free ds_read_b128 per four MFMA steps and tile instead of four ds_read_b32, no transposing writes, and
//     backward-data (rows = input channels, taps flipped) is the same kernel on a second image;
//   * the patch is stored WITHOUT halo columns (rows of W floats, 16-byte pieces straight from NCHW rows; halo rows
//     outside the image are fetched from a page of zeros): the left / right taps of the first / last pixel of a row
//     are masked in registers (one v_cndmask per operand read) — halo columns would break the lane-linear LDS image
//     the DMA writes;
//   * workgroups are persistent (grid = 2 per CU): the first chunk of the next tile is in flight during the last chunk
//     and the output stores of the current one, so the 64-channel layers (8 chunks per tile) no longer pay a cold
//     prologue per tile.
// Numerics: exact fp32 FMA chains like conv_igemm; the summation order over (channel, tap) inside a chunk differs
// (k-steps pair channels (2i, 2i+1) as before, in the same order), results are bit-identical to conv_igemm's.
#include "salun_common.h"
#include <mutex>
#include <unordered_set>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) char *ring_lds_ptr_t;

// Tuning switch (0 in the product build; tools/lab_build.sh builds A/B libraries with it — results are WRONG with a bit
// set): conv3x3_wgrad_ring  1 no DMA after the first chunk   2 no MFMAs   4 no partial-sum stores
#ifndef SALUN_WGR_EXP
#define SALUN_WGR_EXP 0
#endif

constexpr int RCC = 8;      // reduction channels per chunk
constexpr int RSTEPS = 36;  // k-steps per chunk: (RCC / 2) channel pairs x 9 taps
constexpr int RGROUPS = 9;  // 16-byte A groups per chunk and 32-row tile (4 k-steps each)

__device__ float g_ring_zero[256];  // one DMA unit of zeros: the source of patch rows outside the image

// One LDS-DMA unit: 64 lanes x 16 bytes from per-lane global addresses to `dst` (wave-uniform LDS byte address) + 16 * lane.
// Inline assembly on purpose: with the `__builtin_amdgcn_global_load_lds` intrinsic hipcc's wait-count pass treats every
// later LDS read as possibly aliasing the DMA writes in flight and puts `s_waitcnt vmcnt(0)` in front of the first
// ds_read_b128 of the chunk — which drains the ring right after it was refilled (seen in the ISA of the first build of
// this kernel).  The loop's own vmcnt(0) + s_barrier (a stage is read only after both) is the synchronisation; the
// compiler's counts for its own loads and stores stay conservative (an uncounted DMA only makes it wait longer).
__device__ __forceinline__ void ring_dma16(const float *src, unsigned dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}

// ---------------------------------------------------------------------------------------------------
// Weight image.  img[tile][chunk][g][lane][e]: the A operand of lane (lo = lane & 31: row tile*32 + lo of the GEMM,
// hi = lane >> 5: odd / even channel of the pair) at k-step s = 4g + e of the chunk: channel pair cp = s / 9, tap
// rs = s % 9, reduction channel ch = 8*chunk + 2*cp + hi.
//   forward        rows = output channels k:  img = w[k = row][c = ch][rs]
//   backward-data  rows = input channels c:   img = w[k = ch][c = row][8 - rs]   (taps flipped: dx = conv(dy, w^T flipped))
// Rows past the last one are zero.  One thread per 16-byte piece; grid (x, job, direction): ONE launch re-packs both
// images of up to 32 layers (the whole ResNet-18 after an optimizer step).
struct PackJobs {
  salun_pack_job_t job[SALUN_PACK_MAX_JOBS];
};

__global__ __launch_bounds__(256) void k_ring_pack(const PackJobs jobs) {
  const salun_pack_job_t jb = jobs.job[blockIdx.y];
  const int dgrad = blockIdx.z;
  float *__restrict__ img = dgrad ? jb.img_dgrad : jb.img_fwd;
  if (!img) return;
  const float *__restrict__ w = jb.w;
  const int K = jb.K, C = jb.C;
  const int rows = dgrad ? C : K, red = dgrad ? K : C;
  if (red % RCC != 0) return;
  const int nchunk = red / RCC, ntile = (rows + 31) / 32;
  const long long total = (long long)ntile * nchunk * RGROUPS * 64;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    long long r = i >> 6;
    const int gq = (int)(r % RGROUPS);
    r /= RGROUPS;
    const int chunk = (int)(r % nchunk), tile = (int)(r / nchunk);
    const int lo = lane & 31, hi = lane >> 5, row = tile * 32 + lo;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = 4 * gq + e, cp = s / 9, rs = s - 9 * cp, ch = chunk * RCC + 2 * cp + hi;
      float t = 0.f;
      if (row < rows) t = dgrad ? w[((size_t)ch * C + row) * 9 + (8 - rs)] : w[((size_t)row * C + ch) * 9 + rs];
      v[e] = t;
    }
    *reinterpret_cast<float4 *>(img + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

struct RingArgs {
  const float *x;       // [N][Cred][H][W]: forward input, or dY for backward-data
  const float *aimg;    // packed weight image (k_ring_pack)
  const float *bias;    // [Kout] or null
  const float *nbias;   // [N][Kout] or null
  const float *addend;  // [N][Kout][H][W] or null (may alias y)
  const float *zero;    // g_ring_zero
  float *y;             // [N][Kout][H][W]
  int N, Cred, H, Kout;
  int NI, TP;           // images per tile, rows per image per tile
  int ntile_k, ntiles;  // channel blocks, tiles in all (pixel tiles x channel blocks)
  int nchunk;           // Cred / RCC
};

//   LOGW     log2 of the image width (4 .. 32; the tile spans whole rows)
//   PT x KT  32-pixel x 32-channel MFMA tiles per wave
//   WP x WK  wave grid (WP * WK == 4)
//   EPI      epilogue terms (bias, per-image bias, full-size addend) compiled in
template <int LOGW, int PT, int KT, int WP, int WK, bool EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_ring(const RingArgs g) {
  constexpr int W = 1 << LOGW, W4 = W / 4;
  constexpr int KTW = WK * KT, KB = KTW * 32;
  constexpr int A_BYTES = KTW * RGROUPS * 1024;
  constexpr int NA = KTW * RGROUPS;  // A DMA units (1 KB each) per chunk
  constexpr int NAW = (NA + 3) / 4;  // ... per wave
  constexpr int NPW = 3;             // patch DMA units per wave (<= 12 units per chunk: 384 floats per channel)
  static_assert(WP * WK == 4, "four waves");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int wp = wave % WP, wk = wave / WP;
  const int H = g.H, TP = g.TP, NI = g.NI, N = g.N, Cred = g.Cred;
  const int PRI = TP + 2;               // patch rows per image that carry data
  // rows per image in LDS: a 32-lane operand read of 4-wide images covers TWO images (16 pixels each); with the
  // images 6 rows = 24 floats apart, lanes 16-31 fall on the banks of lanes 0-7 (2-way, 40 % of the LDS cycles of the
  // 512-channel layers: profiles/r06_conv_sq_*); 12 rows = 48 floats apart they take banks 16-31
  const int IRS = (W == 4 && NI > 1) ? 12 : PRI;
  const int PSZ = NI * IRS * W;         // patch floats per channel
  const int PSZ4 = PSZ / 4;
  const int npiece = RCC * PSZ4;        // 16-byte pieces of a patch chunk
  const int P_BYTES = ((npiece + 63) / 64) * 1024;
  const int STAGE = A_BYTES + P_BYTES;
  const int HW = H * W;
  const int tiles_per_img = (NI > 1) ? 1 : H / TP;
  const int nchunk = g.nchunk;
  const unsigned lds_base = (unsigned)(uintptr_t)(ring_lds_ptr_t)lds;  // LDS byte address of the ring (0 in practice)

  // ---- issue side: per-lane sources of this wave's DMA units, for the tile whose chunks are being requested
  const float *psrc[NPW];
  int pstep[NPW];
  bool pact[NPW];
  const float *asrc[NAW];
  auto setup_issue = [&](int tile) {
    const int kb = tile % g.ntile_k, ptile = tile / g.ntile_k;
    int n0, p0;
    if (NI > 1) { n0 = ptile * NI; p0 = 0; }
    else { n0 = ptile / tiles_per_img; p0 = (ptile - n0 * tiles_per_img) * TP; }
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      const int e = (wave + 4 * j) * 64 + lane;
      const int c = e / PSZ4, jj = e - c * PSZ4;
      const int row = jj >> (LOGW - 2), col4 = jj & (W4 - 1);
      const int ni = row / IRS, rr = row - ni * IRS;
      const int ih = p0 - 1 + rr, n = n0 + ni;
      const bool act = e < npiece && rr < PRI;   // (rows PRI .. IRS-1 are spacing: never written, never read)
      const bool valid = act && n < N && ih >= 0 && ih < H;
      pact[j] = act;
      psrc[j] = valid ? g.x + ((size_t)(n * Cred + c) * H + ih) * W + 4 * col4 : g.zero + 4 * lane;
      pstep[j] = valid ? RCC * HW : 0;
    }
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
      const int a = wave + 4 * j, t = a / RGROUPS, gq = a - t * RGROUPS;
      // a channel block that hangs over the last 32-row tile of the image re-reads that tile (the image ends there; its
      // results are masked in the epilogue)
      const int rt = min(kb * KTW + t, (g.Kout + 31) / 32 - 1);
      asrc[j] = g.aimg + ((size_t)rt * nchunk * RGROUPS + gq) * 256 + 4 * lane;
    }
  };
  auto issue = [&](int chunk, int stage) {
    const unsigned base = lds_base + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NAW; ++j) {
      const int a = wave + 4 * j;
      if (NA % 4 == 0 || a < NA) {
        const float *p = asrc[j] + (size_t)chunk * (RGROUPS * 256);
        ring_dma16(p, base + a * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      if (pact[j]) {
        const float *p = psrc[j] + (size_t)chunk * pstep[j];
        ring_dma16(p, base + A_BYTES + (wave + 4 * j) * 1024);
      }
    }
  };

  // (Tried and measured, round 6: s_setprio 1 .. 3 for the whole kernel, so that backward-data wins the matrix pipe
  // against the side stream's backward-weight wave on its SIMD — the ResNet-18 step got 0.5 - 1.5 % SLOWER at every level,
  // and the same on the backward-weight side: the two streams finish together best when the hardware arbitrates.)
  const int total = g.ntiles;
  int it_tile = blockIdx.x, it_chunk = 0;
  if (it_tile < total) {
    setup_issue(it_tile);
    issue(0, 0);
    it_chunk = 1;
    if (it_chunk == nchunk) { it_chunk = 0; it_tile += gridDim.x; if (it_tile < total) setup_issue(it_tile); }
  }
  int stage = 0;

  for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
    // ---- compute side: this lane's output pixels
    const int kb = tile % g.ntile_k, ptile = tile / g.ntile_k;
    int n0, p0;
    if (NI > 1) { n0 = ptile * NI; p0 = 0; }
    else { n0 = ptile / tiles_per_img; p0 = (ptile - n0 * tiles_per_img) * TP; }
    int q_l[PT], n_l[PT], p_l[PT], boff[PT];
    bool okL[PT], okR[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int mloc = (wp * PT + pt) * 32 + lo;
      q_l[pt] = mloc & (W - 1);
      const int pr = mloc >> LOGW;
      const int ni = pr / TP;
      p_l[pt] = pr - ni * TP;
      n_l[pt] = n0 + ni;
      // byte offset of tap (r = 0, s = 0) of this pixel in channel `hi` of the patch (patch row 0 = image row p0 - 1)
      boff[pt] = ((hi * PSZ) + (ni * IRS + p_l[pt]) * W + q_l[pt] - 1) * 4;
      okL[pt] = q_l[pt] != 0;
      okR[pt] = q_l[pt] != W - 1;
    }

    f32x16 acc[PT][KT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[pt][t][v] = 0.f;

    for (int ch = 0; ch < nchunk; ++ch) {
      // vmcnt(0): this wave's share of the chunk (requested a whole chunk ago) is in LDS.  Not at the first chunk of a
      // later tile: its DMA was waited for BEFORE the previous tile's output stores were issued (below) — a wait here
      // would also drain those 16 .. 64 stores per lane, a few microseconds per tile with nothing to hide them behind
      // (both workgroups of a CU reach their epilogues together).
      if (ch > 0 || tile == (int)blockIdx.x) __builtin_amdgcn_s_waitcnt(0x0f70);
      __builtin_amdgcn_s_barrier();        // ... everyone's is, and everyone has finished reading the other stage
      if (it_tile < total) {
        issue(it_chunk, stage ^ 1);
        if (++it_chunk == nchunk) { it_chunk = 0; it_tile += gridDim.x; if (it_tile < total) setup_issue(it_tile); }
      }
      // ---- 36 k-steps: operands of step s+1 are read while the MFMAs of step s run
      const char *A = lds + stage * STAGE + (wk * KT * RGROUPS) * 1024 + lane * 16;
      const char *Pb = lds + stage * STAGE + A_BYTES;
      const char *bb[PT][RCC / 2];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int cp = 0; cp < RCC / 2; ++cp) bb[pt][cp] = Pb + boff[pt] + cp * (2 * PSZ * 4);
      float4 a_cur[KT], a_nxt[KT];
      float b_cur[PT], b_nxt[PT];
      // the raw read of step s+1 is issued BEFORE the MFMAs of step s, its edge mask applied AFTER them: a v_cndmask in
      // front of the MFMAs would put the read's lgkmcnt wait there, with the matrix pipe empty behind it
      auto read_b = [&](int s, float (&bv)[PT]) {
        const int cp = s / 9, rs = s % 9, r = rs / 3, sx = rs % 3;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) bv[pt] = *reinterpret_cast<const float *>(bb[pt][cp] + (r * W + sx) * 4);
      };
      auto mask_b = [&](int s, float (&bv)[PT]) {
        const int sx = (s % 9) % 3;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          if (sx == 0) bv[pt] = okL[pt] ? bv[pt] : 0.f;
          if (sx == 2) bv[pt] = okR[pt] ? bv[pt] : 0.f;
        }
      };
      auto read_a = [&](int gq, float4 (&av)[KT]) {
#pragma unroll
        for (int t = 0; t < KT; ++t) av[t] = *reinterpret_cast<const float4 *>(A + (t * RGROUPS + gq) * 1024);
      };
      read_a(0, a_cur);
      read_b(0, b_cur);
      mask_b(0, b_cur);
#pragma unroll
      for (int s = 0; s < RSTEPS; ++s) {
        if (s + 1 < RSTEPS) read_b(s + 1, b_nxt);
        if ((s & 3) == 0 && s + 4 < RSTEPS) read_a(s / 4 + 1, a_nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            const float4 a4 = a_cur[t];
            const float av = (s & 3) == 0 ? a4.x : (s & 3) == 1 ? a4.y : (s & 3) == 2 ? a4.z : a4.w;
            acc[pt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b_cur[pt], acc[pt][t], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < RSTEPS) mask_b(s + 1, b_nxt);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) b_cur[pt] = b_nxt[pt];
        if ((s & 3) == 3) {
#pragma unroll
          for (int t = 0; t < KT; ++t) a_cur[t] = a_nxt[t];
        }
      }
      stage ^= 1;
    }

    // ---- epilogue: D[row = channel][col = pixel]; row = (v & 3) + 8 * (v >> 2) + 4 * hi
    __builtin_amdgcn_s_waitcnt(0x0f70);  // the next tile's first chunk (in flight since the last barrier) has landed
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int n_out = n_l[pt], p_out = p0 + p_l[pt];
      if (n_out < N) {
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const int kbase = kb * KB + (wk * KT + t) * 32 + 4 * hi;
          const size_t obase = (((size_t)n_out * g.Kout + kbase) * H + p_out) * W + q_l[pt];
          const size_t kstride = (size_t)HW;
          const int klast = g.Kout - 1 - kbase;  // < 0: the whole tile is past the last channel
          float bv[16], nv[16], av[16];
          if (EPI) {
            // all terms of a tile are requested before its first store, unconditionally (rows past the last channel
            // re-read the last one): y may alias the addend, so a load written after a store waits behind it
            if (g.bias) {
#pragma unroll
              for (int v = 0; v < 16; ++v) bv[v] = g.bias[kbase + max(min((v & 3) + 8 * (v >> 2), klast), -kbase)];
            }
            if (g.nbias) {
#pragma unroll
              for (int v = 0; v < 16; ++v)
                nv[v] = g.nbias[(size_t)n_out * g.Kout + kbase + max(min((v & 3) + 8 * (v >> 2), klast), -kbase)];
            }
            if (g.addend) {
#pragma unroll
              for (int v = 0; v < 16; ++v)
                av[v] = g.addend[obase + (long long)max(min((v & 3) + 8 * (v >> 2), klast), -kbase) * (long long)kstride];
            }
          }
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int kr = (v & 3) + 8 * (v >> 2);
            if (kbase + kr < g.Kout) {
              float o = acc[pt][t][v];
              if (EPI) {
                if (g.bias) o += bv[v];
                if (g.nbias) o += nv[v];
                if (g.addend) o += av[v];
              }
              g.y[obase + kr * kstride] = o;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward-weight, 3x3 / stride 1 / pad 1, square images of width 4 .. 32:  dW[k][c][r][s] = sum_pix dY[k][pix] X[c][pix + (r-1, s-1)].
// Same tile as conv_wgrad_v (salun_conv.hip): workgroup = 64 k x 64 c, wave = 32 k x 32 c x 9 taps (144 accumulator
// registers, one wave per SIMD), the pixels in chunks of 64, partial sums [split][tap][k][c] -> conv_wgrad_reduce.
// What changes is how the operands travel and how they are read:
//   * chunk i+1 goes global -> LDS by LDS-DMA (16-byte pieces of dY rows and of patch rows; patch rows outside the
//     image come from the page of zeros) while chunk i is multiplied: no global_load / ds_write_b32 instruction pairs
//     dealt over the MFMA slots (conv_wgrad_v: 12 + 48 per thread and chunk), one raw s_barrier per chunk;
//   * LDS rows are [channel][pixels] with a row length of an ODD number of 16-byte pieces, read with ds_read_b128:
//     conflict-free across the 32 rows of an operand, and a lane gets FOUR consecutive pixels — four MFMA steps — per
//     read.  The reduction index of an MFMA step is a pixel, so any pairing works as long as A and B agree: half `hi`
//     of the wave takes pixels 8m + 4hi + e at step 4m + e;
//   * the nine taps of those four pixels need, per patch row, the pixels q0-1 .. q0+4: one aligned ds_read_b128 plus the
//     two neighbours from the adjacent aligned pieces (zero where the neighbour is outside the image row: a compile-time
//     property of (m, hi), applied with one v_cndmask).  The B operand of tap (r, s) at step e is then simply register
//     e + s of that row's six: 10 conflict-free LDS reads per 36 MFMAs where conv_wgrad_v issues 40.
// Summation order over the pixels differs from conv_wgrad_v's (same exact-fp32 FMA chains, other pairing).
struct WgradRingArgs {
  const float *x;     // [N][C][H][W]
  const float *dy;    // [N][K][H][W]
  const float *zero;  // g_ring_zero (its address as an argument: taken in the kernel it is re-loaded from the GOT at every use)
  float *part;        // [nsplit][9][K][C]
  int N, C, K, nchunks;
};

template <int LOGW>
__global__ __launch_bounds__(256) void conv3x3_wgrad_ring(const WgradRingArgs g) {
  constexpr int W = 1 << LOGW, W4 = W / 4, H = W;
  constexpr int TP = (W * W >= 64) ? 64 / W : W;       // image rows per chunk
  constexpr int NI = (W * W >= 64) ? 1 : 64 / (W * W);  // images per chunk
  constexpr int PRI = TP + 2;                           // patch rows per image
  constexpr int PSZ4 = NI * PRI * W4;                   // 16-byte pieces per channel row of the patch
  constexpr int PRX = PSZ4 | 1;                         // ... padded to an odd count
  constexpr int DRW = 17;                               // pieces per dY row (16 + 1)
  constexpr int D_BYTES = 64 * DRW * 16, X_BYTES = 64 * PRX * 16, STAGE = D_BYTES + X_BYTES;
  constexpr int NDU = DRW, NXU = PRX;                   // DMA units (64 pieces) of a chunk: dY, patch
  constexpr int NDW = (NDU + 3) / 4, NXW = (NXU + 3) / 4;  // ... per wave
  constexpr int TPI = (NI > 1) ? 1 : H / TP;            // chunks per image
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const unsigned lds_base = (unsigned)(uintptr_t)(ring_lds_ptr_t)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int kt = wave & 1, ct = wave >> 1;
  const int k0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int split = blockIdx.z, nsplit = gridDim.z;
  const int N = g.N, C = g.C, K = g.K;
  constexpr int HW = H * W;

  // ---- this wave's DMA units.  Source of a piece = chunk base (wave-uniform, SALU) + lane offset (fixed for the kernel);
  // what depends on the chunk per lane is only whether the piece exists (image beyond N, patch row outside the image):
  // a few compares and two selects per unit in the MFMA slot that issues it.
  constexpr int NEVER = 1 << 24;        // image index of the pad pieces: never inside the batch
  int d_lane[NDW], d_ni[NDW];
#pragma unroll
  for (int j = 0; j < NDW; ++j) {
    const int e = (wave + 4 * j) * 64 + lane;
    const int k = e / DRW, pc = e - k * DRW;  // pc == 16: the pad piece
    const int pix = 4 * pc;                   // pixel inside the chunk: (ni, p, q) flattened
    const int ni = pix / (TP * W), rem = pix - ni * (TP * W);
    d_ni[j] = (pc == 16 || k0 + k >= K) ? NEVER : ni;
    d_lane[j] = (ni * K + k) * HW + rem;      // from &dy[n0][k0][p0][0]
  }
  int x_lane[NXW], x_ni[NXW], x_rr[NXW];
#pragma unroll
  for (int j = 0; j < NXW; ++j) {
    const int e = (wave + 4 * j) * 64 + lane;
    const int c = e / PRX, pc = e - c * PRX;
    const int row = pc / W4, col4 = pc - row * W4;
    const int ni = row / PRI, rr = row - ni * PRI;
    x_ni[j] = (pc >= PSZ4) ? NEVER : ni;
    x_rr[j] = rr;
    x_lane[j] = ((ni * C + c) * H + rr) * W + 4 * col4;  // from &x[n0][c0][p0 - 1][0]
  }
  const float *const zero = g.zero + 4 * lane;
  // one DMA unit of the chunk `chunk` into stage `stage`: u < NDW: dY unit u, else patch unit u - NDW
  auto issue_unit = [&](int u, int chunk, int stage) {
    int n0, p0;
    if (NI > 1) { n0 = chunk * NI; p0 = 0; }
    else { n0 = chunk / TPI; p0 = (chunk - n0 * TPI) * TP; }
    const unsigned base = lds_base + stage * STAGE;
    if (u < NDW) {
      const int j = u;
      if ((wave + 4 * j) < NDU) {
        const float *src = g.dy + ((size_t)(n0 * K + k0) * H + p0) * W;
        const float *p = (d_ni[j] < N - n0) ? src + d_lane[j] : zero;
        ring_dma16(p, base + (wave + 4 * j) * 1024);
      }
    } else {
      const int j = u - NDW;
      if ((wave + 4 * j) < NXU) {
        const float *src = g.x + ((long long)(n0 * C + c0) * H + (p0 - 1)) * W;
        const bool ok = x_ni[j] < N - n0 && x_rr[j] >= 1 - p0 && x_rr[j] < H + 1 - p0;
        const float *p = ok ? src + x_lane[j] : zero;
        ring_dma16(p, base + D_BYTES + (wave + 4 * j) * 1024);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  if (split < g.nchunks) {
#pragma unroll
    for (int u = 0; u < NDW + NXW; ++u) issue_unit(u, split, 0);
    int stage = 0;
    const int a_lane = (kt * 32 + lo) * (DRW * 16) + hi * 16;
    const int b_lane = D_BYTES + (ct * 32 + lo) * (PRX * 16) + hi * 16;
    for (int chunk = split; chunk < g.nchunks; chunk += nsplit) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's units of the chunk (requested a chunk ago) have landed
      __builtin_amdgcn_s_barrier();        // ... everyone's have, and everyone is done reading the other stage
      const int nxt_chunk = chunk + nsplit;
      const bool more = nxt_chunk < g.nchunks;
      const char *A = lds + stage * STAGE + a_lane;
      const char *B = lds + stage * STAGE + b_lane;
      // operands of pixel group m: A = 4 pixels of dY; B rows r = 0..2: {left neighbour, 4 pixels, right neighbour}
      // (plain scalars on purpose: as [3][6] arrays passed to the reading lambdas the neighbours were kept in scratch)
      struct Row { float l; float4 v; float r; };
      struct Ops { float4 a; Row b0, b1, b2; };
      auto group_off = [&](int m) {  // float offset of pixel group 2m (half 0) inside a patch channel row, tap row 0
        const int pix = 8 * m, ni = pix / (TP * W), rem = pix - ni * (TP * W), p = rem / W, q0 = rem - p * W;
        return (ni * PRI + p) * W + q0;
      };
      auto read_row = [&](int off) {
        Row o;
        o.v = *reinterpret_cast<const float4 *>(B + off);
        if (W > 4) {
          // the neighbours as the last / first element of the ADJACENT aligned pieces: a dword read at a row stride of
          // 4 x odd floats is a 4-way bank conflict (64 % of this kernel's LDS cycles in the first build:
          // profiles/r06_conv_sq_*), the 16-byte read is conflict-free like the operand's own
          // (volatile: hipcc narrows a 16-byte load of which one element is used back to the conflicting dword read)
          typedef float f32x4v __attribute__((ext_vector_type(4)));
          typedef const volatile __attribute__((address_space(3))) f32x4v *lds_v4_t;  // (explicitly LDS: a volatile
          const f32x4v lv = *(lds_v4_t)(ring_lds_ptr_t)(B + off - 16);                 //  generic pointer loads flat)
          const f32x4v rv = *(lds_v4_t)(ring_lds_ptr_t)(B + off + 16);
          o.l = lv.w;
          o.r = rv.x;
        } else {
          o.l = 0.f;
          o.r = 0.f;
        }
        return o;
      };
      auto read_group = [&](int m) {
        Ops o;
        o.a = *reinterpret_cast<const float4 *>(A + m * 32);
        const int off = group_off(m) * 4;
        o.b0 = read_row(off);
        o.b1 = read_row(off + W * 4);
        o.b2 = read_row(off + 2 * W * 4);
        return o;
      };
      auto mask_row = [&](int m, Row o) {  // neighbours outside the image row
        if (W > 4) {
          const int q00 = (8 * m) & (W - 1);
          if (q00 == 0) o.l = hi ? o.l : 0.f;        // half 0 starts the row
          if (q00 == W - 8) o.r = hi ? 0.f : o.r;    // half 1 ends it
        }
        return o;
      };
      auto pick = [](const Row &o, int i) {  // i = e + s: pixel q0 - 1 + i
        return i == 0 ? o.l : i == 1 ? o.v.x : i == 2 ? o.v.y : i == 3 ? o.v.z : i == 4 ? o.v.w : o.r;
      };
      Ops cur = read_group(0), nxt = cur;
      cur.b0 = mask_row(0, cur.b0); cur.b1 = mask_row(0, cur.b1); cur.b2 = mask_row(0, cur.b2);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (m + 1 < 8) nxt = read_group(m + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float av = e == 0 ? cur.a.x : e == 1 ? cur.a.y : e == 2 ? cur.a.z : cur.a.w;
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const float bv = pick(t / 3 == 0 ? cur.b0 : t / 3 == 1 ? cur.b1 : cur.b2, e + t % 3);
            if (!(SALUN_WGR_EXP & 2)) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            else asm volatile("" ::"v"(av), "v"(bv));
            if (t == 0) {
              // one DMA unit of the next chunk per MFMA step (first half of the chunk), between two MFMAs
              __builtin_amdgcn_sched_barrier(0);
              const int u = 4 * m + e;
              if (u < NDW + NXW && more && !(SALUN_WGR_EXP & 1)) issue_unit(u, nxt_chunk, stage ^ 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (m + 1 < 8) {
          cur.a = nxt.a;
          cur.b0 = mask_row(m + 1, nxt.b0); cur.b1 = mask_row(m + 1, nxt.b1); cur.b2 = mask_row(m + 1, nxt.b2);
        }
      }
      stage ^= 1;
    }
  }
  static_assert(NDW + NXW <= 32, "one DMA unit per MFMA step");
  float *out = g.part + (size_t)split * K * C * 9;
  const int c = c0 + ct * 32 + lo;
  if ((SALUN_WGR_EXP & 4) && K > 0) return;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int k = k0 + kt * 32 + (v & 3) + 8 * (v >> 2) + 4 * hi;
      if (k < K) out[((size_t)t * K + k) * C + c] = acc[t][v];
    }
}

inline void ring_allow_lds(const void *fn) {
  static std::mutex mu;
  static std::unordered_set<uintptr_t> done;
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert(reinterpret_cast<uintptr_t>(fn) ^ ((uintptr_t)(salun_device_bit() + 1) << 56)).second)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// device address of the page of zeros (per device, looked up once)
inline const float *ring_zero_page() {
  static const float *page[64] = {nullptr};
  static std::mutex mu;
  const int b = salun_device_bit();
  std::lock_guard<std::mutex> lk(mu);
  if (!page[b]) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_ring_zero)) != hipSuccess) return nullptr;
    page[b] = static_cast<const float *>(p);
  }
  return page[b];
}

inline int ring_cu_count() {
  static int cus[64] = {0};
  const int b = salun_device_bit();
  if (cus[b] == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus[b] = n;
  }
  return cus[b];
}

struct RingTile { int tpix, kb; };

// tile of the variant `cfg` (0: chosen from the problem); cfg = 1 .. 5 pin one (tools/convring_bench.py)
//   measured, MI355X round 6 (profiles/r06_convring_ab.txt): 256 x 64 wins wherever it yields two workgroups per CU
//   (118-135 TFLOP/s), 128 x 64 next (127-130), 64 x 64 for the 4x4 / 8x8 levels (112-119); below ~1.5 workgroups per
//   CU conv_igemm's reduction split is faster (DDPM 256 -> 256 @4: 31 us against 50) — -1 sends the caller there.
inline int ring_choose(int cfg, int N, int H, int W, int Kout) {
  if (cfg) return cfg;
  const long long pix = (long long)N * H * W;
  const int kb64 = (Kout + 63) / 64;
  if (pix / 256 * kb64 >= 512) return 1;
  if (pix / 128 * kb64 >= 512) return 2;
  if (pix / 64 * kb64 >= 384) return 3;
  return -1;
}

template <int LOGW, int PT, int KT, int WP, int WK>
int ring_launch(RingArgs a, int W, bool epi, int wgs_per_cu, hipStream_t st) {
  constexpr int TPIX = WP * PT * 32, KB = WK * KT * 32;
  const int HW = a.H * W;
  if (TPIX % W != 0) return SALUN_EINVAL;
  if (HW >= TPIX) {
    if (HW % TPIX != 0) return SALUN_EINVAL;
    a.NI = 1;
    a.TP = TPIX / W;
  } else {
    if (TPIX % HW != 0) return SALUN_EINVAL;
    a.NI = TPIX / HW;
    a.TP = a.H;
  }
  const int ptiles = (a.NI > 1) ? (a.N + a.NI - 1) / a.NI : a.N * (HW / TPIX);
  a.ntile_k = (a.Kout + KB - 1) / KB;
  a.ntiles = ptiles * a.ntile_k;
  a.nchunk = a.Cred / RCC;
  const int IRS = (W == 4 && a.NI > 1) ? 12 : a.TP + 2;  // rows per image in LDS (the kernel's IRS)
  const int PSZ = a.NI * IRS * W;
  const int npiece = RCC * PSZ / 4;
  if (npiece > 12 * 64) return SALUN_EINVAL;  // NPW = 3 units per wave
  const size_t stage = (size_t)(WK * KT) * RGROUPS * 1024 + (size_t)((npiece + 63) / 64) * 1024;
  const size_t ldsb = 2 * stage;
  if (ldsb > 80 * 1024) return SALUN_EINVAL;  // two workgroups per CU
  int grid = wgs_per_cu * ring_cu_count();
  if (grid > a.ntiles) grid = a.ntiles;
  if (epi) {
    ring_allow_lds(reinterpret_cast<const void *>(conv3x3_ring<LOGW, PT, KT, WP, WK, true>));
    hipLaunchKernelGGL((conv3x3_ring<LOGW, PT, KT, WP, WK, true>), dim3(grid), dim3(256), ldsb, st, a);
  } else {
    ring_allow_lds(reinterpret_cast<const void *>(conv3x3_ring<LOGW, PT, KT, WP, WK, false>));
    hipLaunchKernelGGL((conv3x3_ring<LOGW, PT, KT, WP, WK, false>), dim3(grid), dim3(256), ldsb, st, a);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

template <int LOGW>
int ring_dispatch(const RingArgs &a, int cfg, bool epi, int wgs_per_cu, hipStream_t st) {
  constexpr int W = 1 << LOGW;
  switch (ring_choose(cfg, a.N, a.H, W, a.Kout)) {
    case 1: return ring_launch<LOGW, 2, 2, 4, 1>(a, W, epi, wgs_per_cu, st);  // 256 pixels x 64 channels
    case 2: return ring_launch<LOGW, 1, 2, 4, 1>(a, W, epi, wgs_per_cu, st);  // 128 x 64
    case 3: return ring_launch<LOGW, 1, 1, 2, 2>(a, W, epi, wgs_per_cu, st);  // 64 x 64
    case 4: return ring_launch<LOGW, 1, 4, 4, 1>(a, W, epi, wgs_per_cu, st);  // 128 x 128
    case 5: return ring_launch<LOGW, 1, 2, 2, 2>(a, W, epi, wgs_per_cu, st);  // 64 x 128
    default: return SALUN_EINVAL;
  }
}

}  // namespace

// Called by salun_conv2d_backward_weight (salun_conv.hip) for 3x3 / stride 1 / pad 1 layers before its own kernels:
// SALUN_EINVAL = not this kernel's shape.  `part` = [nsplit][9][K][C], nchunks = 64-pixel chunks of the batch.
int salun_ring_wgrad_launch(const float *x, const float *dy, float *part, int N, int C, int H, int W, int K, int nsplit,
                            int nchunks, hipStream_t st) {
  if (H != W || C % 64 != 0 || K % 32 != 0 || !salun_aligned16(x) || !salun_aligned16(dy)) return SALUN_EINVAL;
  if ((long long)N * C * H * W >= (1ll << 31) || (long long)N * K * H * W >= (1ll << 31)) return SALUN_EINVAL;
  const float *zero = ring_zero_page();
  if (!zero) return SALUN_EIO;
  WgradRingArgs a{x, dy, zero, part, N, C, K, nchunks};
  dim3 grid((K + 63) / 64, C / 64, nsplit);
#define SALUN_WGR(LOGW_)                                                                                        \
  {                                                                                                             \
    constexpr int W_ = 1 << LOGW_, TP_ = (W_ * W_ >= 64) ? 64 / W_ : W_, NI_ = (W_ * W_ >= 64) ? 1 : 64 / (W_ * W_);  \
    constexpr int PRX_ = (NI_ * (TP_ + 2) * (W_ / 4)) | 1;                                                      \
    const size_t ldsb = 2 * (size_t)(64 * 17 * 16 + 64 * PRX_ * 16);                                            \
    const int want = (NI_ > 1) ? (N + NI_ - 1) / NI_ : N * (W_ * W_ / 64);                                      \
    if (want != nchunks) return SALUN_EINVAL;                                                                   \
    ring_allow_lds(reinterpret_cast<const void *>(conv3x3_wgrad_ring<LOGW_>));                                  \
    hipLaunchKernelGGL((conv3x3_wgrad_ring<LOGW_>), grid, dim3(256), ldsb, st, a);                              \
  }
  switch (W) {
    case 4: SALUN_WGR(2) break;
    case 8: SALUN_WGR(3) break;
    case 16: SALUN_WGR(4) break;
    case 32: SALUN_WGR(5) break;
    default: return SALUN_EINVAL;
  }
#undef SALUN_WGR
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT size_t salun_conv3x3_pack_bytes(int K, int C, int dgrad) {
  const int rows = dgrad ? C : K, red = dgrad ? K : C;
  if (rows <= 0 || red <= 0 || red % RCC != 0) return 0;
  return (size_t)((rows + 31) / 32) * (red / RCC) * RGROUPS * 1024;
}

SALUN_EXPORT int salun_conv3x3_pack_weights(const salun_pack_job_t *jobs, int njobs, salun_stream_t stream) {
  if (!jobs || njobs <= 0) return SALUN_EINVAL;
  for (int j0 = 0; j0 < njobs; j0 += SALUN_PACK_MAX_JOBS) {
    PackJobs pj{};
    const int n = (njobs - j0 < SALUN_PACK_MAX_JOBS) ? njobs - j0 : SALUN_PACK_MAX_JOBS;
    long long most = 0;
    for (int j = 0; j < n; ++j) {
      const salun_pack_job_t &jb = jobs[j0 + j];
      if (!jb.w || jb.K <= 0 || jb.C <= 0 || (!jb.img_fwd && !jb.img_dgrad)) return SALUN_EINVAL;
      if ((jb.img_fwd && (jb.C % RCC != 0 || !salun_aligned16(jb.img_fwd))) ||
          (jb.img_dgrad && (jb.K % RCC != 0 || !salun_aligned16(jb.img_dgrad))))
        return SALUN_EINVAL;
      pj.job[j] = jb;
      const long long it = (long long)(((jb.K > jb.C ? jb.K : jb.C) + 31) / 32) * ((jb.K > jb.C ? jb.K : jb.C) / RCC + 1) *
                           RGROUPS * 64;
      if (it > most) most = it;
    }
    int gx = (int)((most + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_ring_pack, dim3(gx, n, 2), dim3(256), 0, salun_hip_stream(stream), pj);
    SALUN_LAUNCH_CHECK();
  }
  return SALUN_OK;
}

SALUN_EXPORT int salun_conv3x3_packed(const float *x, const float *img, const float *bias, const float *nbias,
                                      const float *addend, float *y, int N, int Cred, int H, int W, int Kout, int cfg,
                                      salun_stream_t stream) {
  if (!x || !img || !y || N <= 0 || Cred <= 0 || Kout <= 0 || H <= 0) return SALUN_EINVAL;
  if (Cred % RCC != 0 || !salun_aligned16(x) || !salun_aligned16(img)) return SALUN_EINVAL;
  if ((long long)N * Cred * H * W >= (1ll << 31) || (long long)N * Kout * H * W >= (1ll << 31)) return SALUN_EINVAL;
  RingArgs a{};
  a.x = x; a.aimg = img; a.bias = bias; a.nbias = nbias; a.addend = addend; a.y = y;
  a.zero = ring_zero_page();
  if (!a.zero) return SALUN_EIO;
  a.N = N; a.Cred = Cred; a.H = H; a.Kout = Kout;
  const bool epi = bias || nbias || addend;
  const int wgs = (cfg >> 8) ? (cfg >> 8) : 2;
  hipStream_t st = salun_hip_stream(stream);
  switch (W) {
    case 4: return ring_dispatch<2>(a, cfg & 255, epi, wgs, st);
    case 8: return ring_dispatch<3>(a, cfg & 255, epi, wgs, st);
    case 16: return ring_dispatch<4>(a, cfg & 255, epi, wgs, st);
    case 32: return ring_dispatch<5>(a, cfg & 255, epi, wgs, st);
    default: return SALUN_EINVAL;
  }
}
