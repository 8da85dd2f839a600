// salun_attn.hip — K13: fused scaled-dot-product attention (forward, backward) for bf16 tokens on the CDNA4 matrix cores.
//
// The attention of the Stable-Diffusion U-Net's transformer blocks (reference: SD/ldm/modules/attention.py:168-192
// `CrossAttention.forward`: einsum('b i d, b j d -> b i j') * scale -> softmax -> einsum('b i j, b j d -> b i d'), which
// materialises a (B*8) x 4096 x 4096 score tensor per block): 4096 / 1024 / 256 / 64 query tokens, 8 heads of
// 40 / 80 / 160 channels, keys = the same tokens (self-attention) or the 77 text tokens (cross-attention).
//
// Shape of the computation (MI355X-first): everything is kept TRANSPOSED so that a lane owns one query.
//   S^T[key][query] = K[key][:] . Q[query][:]        v_mfma_f32_32x32x16_bf16, A = K rows from LDS, B = Q from registers
//   -> a lane holds 16 of the 32 keys of its query per tile (the other 16 sit in lane^32): row max / sum are in-lane
//      loops plus one exchange, the running (m, l) are per-lane scalars, and P^T is already the B operand of
//   O^T[d][query] += V^T[d][key] . P^T[key][query]    A = V read TRANSPOSED out of its row-major LDS image
//      (ds_read_b64_tr_b16) — no shuffles, no P round trip through LDS; rescaling O^T is a per-lane multiply.
// Tensors are [B, tokens, H, D] views (token stride given): the Linear projections' outputs are read in place and O is
// written in the layout the output projection reads — no head split / merge copies.
// Backward: `salun_attn_backward` = D_q = sum_d dO*O; dQ kernel (same orientation as forward: S^T, dP^T = V . dO^T,
// dS^T = P^T*(dP^T - D_q)*scale, dQ^T += K^T . dS^T) and dK/dV kernel (a lane owns one key: S = Q . K^T, dV^T += dO^T . P,
// dP = dO . V^T, dK^T += Q^T . dS).  P and dS are rounded to bf16 for the second GEMMs (as every flash attention does);
// all accumulation and the softmax are fp32.
#include "salun_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }  // v_cvt_pk_bf16_f32: RNE
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__device__ __forceinline__ bf16x8 tr_operand(const char *lds_base, uint32_t byte_off, int second) {
  lds_s16x4_ptr p0 = (lds_s16x4_ptr)(lds_base + byte_off);
  lds_s16x4_ptr p1 = (lds_s16x4_ptr)(lds_base + byte_off + second);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p1);
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo; u.s.b = hi;
  return u.v;
}
// 8 fp32 -> one bf16x8 operand (element i = f[i])
__device__ __forceinline__ bf16x8 pack_operand(const float *f) {
  union { uint32_t w[4]; bf16x8 v; } u;
#pragma unroll
  for (int j = 0; j < 4; ++j) u.w[j] = pack2(f[2 * j], f[2 * j + 1]);
  return u.v;
}
__device__ __forceinline__ bf16x8 zero_operand() {
  union { uint32_t w[4]; bf16x8 v; } u;
  u.w[0] = u.w[1] = u.w[2] = u.w[3] = 0;
  return u.v;
}

struct AttnArgs {
  const uint16_t *q, *k, *v;     // [B][Nq|Nk][H][D] views: element (b, t, h, d) at b*bs + t*ld + h*D + d
  uint16_t *o;                   // forward output / (backward: the forward's output, read)
  const uint16_t *d_o;           // backward: dO
  uint16_t *dq, *dk, *dv;        // backward outputs
  float *lse;                    // [B*H][Nq] log2-domain logsumexp of the scaled scores
  float *dsum;                   // [B*H][Nq] sum_d dO*O
  long long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs;
  int q_ld, k_ld, v_ld, o_ld, do_ld, dq_ld, dk_ld, dv_ld;
  int B, H, Nq, Nk;
  float scale;
};

template <int D>
struct Geo {
  static constexpr int DK = (D + 15) / 16 * 16;     // reduction extent of Q.K (zero padded)
  static constexpr int DV = (D + 31) / 32 * 32;     // output channels in 32-wide MFMA tiles
  static constexpr int NJ = DK / 16, NT = DV / 32, C8 = D / 8;
  static constexpr int ROWB = DK * 2 + 16;          // row-major LDS row (conflict-free b128 reads over 16 rows)
};

// Tile of R token rows x D channels: global -> registers (`load`), registers -> LDS (`store`) as a row-major image
// (RM, rows of ROWB bytes) and / or a transposable image (TR: DV/32 column blocks of [R][32 channels], 64-byte rows).
template <int D, int R>
struct Tile {
  static constexpr int CH = R * (D / 8);
  static constexpr int NREG = (CH + 255) / 256;
  uint4 r[NREG];
  __device__ __forceinline__ void load(const uint16_t *base, int ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int id = tid + 256 * i;
      const int row = id / (D / 8), cc = id - row * (D / 8);
      r[i] = make_uint4(0, 0, 0, 0);
      if (id < CH && row0 + row < nrows) r[i] = *reinterpret_cast<const uint4 *>(base + (size_t)(row0 + row) * ld + cc * 8);
    }
  }
  __device__ __forceinline__ void store(char *rm, char *tr, int tid) const {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int id = tid + 256 * i;
      if (id >= CH) continue;
      const int row = id / (D / 8), cc = id - row * (D / 8);
      if (rm) *reinterpret_cast<uint4 *>(rm + row * Geo<D>::ROWB + cc * 16) = r[i];
      if (tr) *reinterpret_cast<uint4 *>(tr + (cc >> 2) * (R * 64) + row * 64 + (cc & 3) * 16) = r[i];
    }
  }
};

// zero the pad columns of an LDS image once (staging never touches them)
template <int D, int R>
__device__ __forceinline__ void zero_pads(char *rm, char *tr, int tid) {
  if (rm) {
    constexpr int PADB = Geo<D>::ROWB - D * 2;  // bytes behind the data in every row
    for (int i = tid; i < R * (PADB / 4); i += 256) {
      const int row = i / (PADB / 4), w = i - row * (PADB / 4);
      *reinterpret_cast<uint32_t *>(rm + row * Geo<D>::ROWB + D * 2 + w * 4) = 0u;
    }
  }
  if (tr && (D % 32)) {
    constexpr int LASTC = (D % 32) * 2;  // valid bytes in a row of the last column block
    char *blk = tr + (Geo<D>::NT - 1) * (R * 64);
    for (int i = tid; i < R * ((64 - LASTC) / 4); i += 256) {
      const int row = i / ((64 - LASTC) / 4), w = i - row * ((64 - LASTC) / 4);
      *reinterpret_cast<uint32_t *>(blk + row * 64 + LASTC + w * 4) = 0u;
    }
  }
}

// B operand held in registers: lane (column = token lane&31, half h) carries channels 16j + 8h .. +7 of its token
template <int D>
__device__ __forceinline__ void load_token_operand(bf16x8 (&f)[Geo<D>::NJ], const uint16_t *row, bool valid, int half) {
#pragma unroll
  for (int j = 0; j < Geo<D>::NJ; ++j) {
    const int d0 = 16 * j + 8 * half;
    f[j] = zero_operand();
    if (valid && d0 < D) {
      union { uint4 u; bf16x8 v; } t;
      t.u = *reinterpret_cast<const uint4 *>(row + d0);
      f[j] = t.v;
    }
  }
}

// store a transposed accumulator (rows = channels, column = this lane's token) as bf16: 4 consecutive channels = 8 bytes
template <int D>
__device__ __forceinline__ void store_token_rows(const f32x16 (&acc)[Geo<D>::NT], uint16_t *row, bool valid, int half,
                                                 float mul) {
  if (!valid) return;
#pragma unroll
  for (int t = 0; t < Geo<D>::NT; ++t)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int d0 = t * 32 + 8 * a + 4 * half;
      if (d0 < D) {
        const uint32_t lo = pack2(acc[t][4 * a] * mul, acc[t][4 * a + 1] * mul);
        const uint32_t hi = pack2(acc[t][4 * a + 2] * mul, acc[t][4 * a + 3] * mul);
        *reinterpret_cast<uint2 *>(row + d0) = make_uint2(lo, hi);
      }
    }
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr int KT = 64;  // keys (forward / dQ) or queries (dK/dV) staged per tile

// ------------------------------------------------------------------------------------------------------ forward
template <int D>
__global__ __launch_bounds__(256) void attn_fwd(const AttnArgs g) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *k_rm = lds;                       // [KT][ROWB]
  char *v_tr = lds + KT * G::ROWB;        // NT x [KT][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q_idx = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_idx < g.Nq;
  const uint16_t *kb = g.k + b * g.k_bs + h * D, *vb = g.v + b * g.v_bs + h * D;
  bf16x8 qf[G::NJ];
  load_token_operand<D>(qf, g.q + b * g.q_bs + (size_t)q_idx * g.q_ld + h * D, q_ok, half);
  zero_pads<D, KT>(k_rm, v_tr, tid);
  f32x16 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  float m = -INFINITY, l = 0.f;
  const float c = g.scale * LOG2E;
  const int gq = lane >> 4, sl = lane & 15;
  const uint32_t a_off = (uint32_t)((lane & 31) * G::ROWB + half * 16);
  const uint32_t tr_off = (uint32_t)((4 * half + (sl >> 2)) * 64 + (16 * (gq & 1) + 4 * (sl & 3)) * 2);
  Tile<D, KT> tk, tv;
  const int ntiles = (g.Nk + KT - 1) / KT;
  tk.load(kb, g.k_ld, 0, g.Nk, tid);
  tv.load(vb, g.v_ld, 0, g.Nk, tid);
  tk.store(k_rm, nullptr, tid);
  tv.store(nullptr, v_tr, tid);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) {
      tk.load(kb, g.k_ld, (kt + 1) * KT, g.Nk, tid);
      tv.load(vb, g.v_ld, (kt + 1) * KT, g.Nk, tid);
    }
    // ---- S^T for the 64 keys of the tile: two 32-key MFMA tiles, 16 keys of each per lane
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int v = 0; v < 16; ++v) s[sub][v] = 0.f;
    // consecutive MFMAs alternate between independent accumulators (a dependent chain would wait out each result)
#pragma unroll
    for (int j = 0; j < G::NJ; ++j)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(k_rm + a_off + sub * 32 * G::ROWB + j * 32);
        s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[j], s[sub], 0, 0, 0);
      }
    // softmax bookkeeping on the RAW scores (the scale c > 0 commutes with max; it enters through one fma per
    // element: p = exp2(s*c - m)), five VALU operations per score instead of seven
    float mx = -INFINITY;
    if ((kt + 1) * KT > g.Nk) {  // ragged last tile (wave-uniform): keys past the end score -inf
      const int key0 = kt * KT + 4 * half;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int key = key0 + sub * 32 + (v & 3) + 8 * (v >> 2);
          if (key >= g.Nk) s[sub][v] = -INFINITY;
          mx = fmaxf(mx, s[sub][v]);
        }
    } else {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int v = 0; v < 16; ++v) mx = fmaxf(mx, s[sub][v]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c;
    const float m_new = fmaxf(m, mx);  // finite: every tile holds at least one valid key
    const bool grew = m_new > m;
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    float ps = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float p[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) { p[v] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][v], c, -m_new)); ps += p[v]; }
      pf[sub][0] = pack_operand(p);
      pf[sub][1] = pack_operand(p + 8);
    }
    l = l * alpha + ps;
    // the running maximum settles after the first tiles: rescale O^T only when some lane's maximum grew (wave-uniform)
    const bool rescale = __builtin_amdgcn_ballot_w64(grew) != 0;
    if (rescale) {
#pragma unroll
      for (int t = 0; t < G::NT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] *= alpha;
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {  // channel tiles innermost: independent accumulators back to back
          const bf16x8 a = tr_operand(v_tr, tr_off + t * (KT * 64) + (sub * 32 + 16 * e) * 64, 8 * 64);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[sub][e], acc[t], 0, 0, 0);
        }
    __syncthreads();
    if (kt + 1 < ntiles) {
      tk.store(k_rm, nullptr, tid);
      tv.store(nullptr, v_tr, tid);
    }
    __syncthreads();
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  store_token_rows<D>(acc, g.o + b * g.o_bs + (size_t)q_idx * g.o_ld + h * D, q_ok, half, 1.f / lt);
  if (g.lse && q_ok && half == 0) g.lse[(size_t)bh * g.Nq + q_idx] = m + __builtin_amdgcn_logf(lt);  // log2 domain
}

// ------------------------------------------------------------------------------------------------------ D = sum dO*O
__global__ __launch_bounds__(256) void attn_dsum(const uint16_t *__restrict__ o, const uint16_t *__restrict__ d_o,
                                                 float *__restrict__ dsum, long long o_bs, int o_ld, long long do_bs, int do_ld,
                                                 int B, int H, int Nq, int D) {
  const int64_t total = (int64_t)B * H * Nq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int q = (int)(i % Nq);
    const int bh = (int)(i / Nq), b = bh / H, h = bh - b * H;
    const uint16_t *po = o + b * o_bs + (size_t)q * o_ld + h * D, *pd = d_o + b * do_bs + (size_t)q * do_ld + h * D;
    float s = 0.f;
    for (int d = 0; d < D; d += 8) {
      const uint4 a = *reinterpret_cast<const uint4 *>(po + d), c = *reinterpret_cast<const uint4 *>(pd + d);
      const uint32_t wa[4] = {a.x, a.y, a.z, a.w}, wc[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        s += bf2f((uint16_t)(wa[j] & 0xffffu)) * bf2f((uint16_t)(wc[j] & 0xffffu)) + bf2f((uint16_t)(wa[j] >> 16)) * bf2f((uint16_t)(wc[j] >> 16));
    }
    dsum[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------------ backward: dQ
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq(const AttnArgs g) {
  using G = Geo<D>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *k_rm = lds;                          // [KT][ROWB]   A operand of S^T
  char *v_rm = lds + KT * G::ROWB;           // [KT][ROWB]   A operand of dP^T
  char *k_tr = lds + 2 * KT * G::ROWB;       // NT x [KT][64]  A operand (transposed) of dQ^T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int q_idx = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool q_ok = q_idx < g.Nq;
  const uint16_t *kb = g.k + b * g.k_bs + h * D, *vb = g.v + b * g.v_bs + h * D;
  bf16x8 qf[G::NJ], dof[G::NJ];
  load_token_operand<D>(qf, g.q + b * g.q_bs + (size_t)q_idx * g.q_ld + h * D, q_ok, half);
  load_token_operand<D>(dof, g.d_o + b * g.do_bs + (size_t)q_idx * g.do_ld + h * D, q_ok, half);
  const float L = q_ok ? g.lse[(size_t)bh * g.Nq + q_idx] : 0.f;
  const float Dq = q_ok ? g.dsum[(size_t)bh * g.Nq + q_idx] : 0.f;
  zero_pads<D, KT>(k_rm, k_tr, tid);
  zero_pads<D, KT>(v_rm, nullptr, tid);
  f32x16 acc[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  const float c = g.scale * LOG2E;
  const int gq = lane >> 4, sl = lane & 15;
  const uint32_t a_off = (uint32_t)((lane & 31) * G::ROWB + half * 16);
  const uint32_t tr_off = (uint32_t)((4 * half + (sl >> 2)) * 64 + (16 * (gq & 1) + 4 * (sl & 3)) * 2);
  Tile<D, KT> tk, tv;
  const int ntiles = (g.Nk + KT - 1) / KT;
  tk.load(kb, g.k_ld, 0, g.Nk, tid);
  tv.load(vb, g.v_ld, 0, g.Nk, tid);
  tk.store(k_rm, k_tr, tid);
  tv.store(v_rm, nullptr, tid);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) {
      tk.load(kb, g.k_ld, (kt + 1) * KT, g.Nk, tid);
      tv.load(vb, g.v_ld, (kt + 1) * KT, g.Nk, tid);
    }
    const int key0 = kt * KT + 4 * half;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int v = 0; v < 16; ++v) s[v] = dp[v] = 0.f;
#pragma unroll
      for (int j = 0; j < G::NJ; ++j) {
        const bf16x8 ak = *reinterpret_cast<const bf16x8 *>(k_rm + a_off + sub * 32 * G::ROWB + j * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ak, qf[j], s, 0, 0, 0);
        const bf16x8 av = *reinterpret_cast<const bf16x8 *>(v_rm + a_off + sub * 32 * G::ROWB + j * 32);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, dof[j], dp, 0, 0, 0);
      }
      float ds[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int key = key0 + sub * 32 + (v & 3) + 8 * (v >> 2);
        const float p = key < g.Nk ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[v], c, -L)) : 0.f;
        ds[v] = p * (dp[v] - Dq) * g.scale;
      }
      const bf16x8 d0 = pack_operand(ds), d1 = pack_operand(ds + 8);
#pragma unroll
      for (int t = 0; t < G::NT; ++t) {
        const bf16x8 a0 = tr_operand(k_tr, tr_off + t * (KT * 64) + (sub * 32) * 64, 8 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, d0, acc[t], 0, 0, 0);
        const bf16x8 a1 = tr_operand(k_tr, tr_off + t * (KT * 64) + (sub * 32 + 16) * 64, 8 * 64);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, d1, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (kt + 1 < ntiles) {
      tk.store(k_rm, k_tr, tid);
      tv.store(v_rm, nullptr, tid);
    }
    __syncthreads();
  }
  store_token_rows<D>(acc, g.dq + b * g.dq_bs + (size_t)q_idx * g.dq_ld + h * D, q_ok, half, 1.f);
}

// ------------------------------------------------------------------------------------------------------ backward: dK, dV
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv(const AttnArgs g) {
  using G = Geo<D>;
  constexpr int QT = 32;  // queries per staged tile
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *q_rm = lds;                             // [QT][ROWB]  A operand of S
  char *do_rm = lds + QT * G::ROWB;             // [QT][ROWB]  A operand of dP
  char *q_tr = lds + 2 * QT * G::ROWB;          // NT x [QT][64]  A operand (transposed) of dK^T
  char *do_tr = q_tr + G::NT * QT * 64;         // NT x [QT][64]  A operand (transposed) of dV^T
  float *ls = reinterpret_cast<float *>(do_tr + G::NT * QT * 64);  // [QT] lse, [QT] dsum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int bh = blockIdx.y, b = bh / g.H, h = bh - b * g.H;
  const int k_idx = blockIdx.x * 128 + wave * 32 + (lane & 31);
  const bool k_ok = k_idx < g.Nk;
  const uint16_t *qb = g.q + b * g.q_bs + h * D, *dob = g.d_o + b * g.do_bs + h * D;
  bf16x8 kf[G::NJ], vf[G::NJ];
  load_token_operand<D>(kf, g.k + b * g.k_bs + (size_t)k_idx * g.k_ld + h * D, k_ok, half);
  load_token_operand<D>(vf, g.v + b * g.v_bs + (size_t)k_idx * g.v_ld + h * D, k_ok, half);
  zero_pads<D, QT>(q_rm, q_tr, tid);
  zero_pads<D, QT>(do_rm, do_tr, tid);
  f32x16 dk[G::NT], dv[G::NT];
#pragma unroll
  for (int t = 0; t < G::NT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) dk[t][v] = dv[t][v] = 0.f;
  const float c = g.scale * LOG2E;
  const int gq = lane >> 4, sl = lane & 15;
  const uint32_t a_off = (uint32_t)((lane & 31) * G::ROWB + half * 16);
  const uint32_t tr_off = (uint32_t)((4 * half + (sl >> 2)) * 64 + (16 * (gq & 1) + 4 * (sl & 3)) * 2);
  Tile<D, QT> tq, td;
  float r_l = 0.f, r_d = 0.f;
  const int ntiles = (g.Nq + QT - 1) / QT;
  auto load_tile = [&](int qt) {
    tq.load(qb, g.q_ld, qt * QT, g.Nq, tid);
    td.load(dob, g.do_ld, qt * QT, g.Nq, tid);
    if (tid < QT) {
      const int qi = qt * QT + tid;
      r_l = qi < g.Nq ? g.lse[(size_t)bh * g.Nq + qi] : INFINITY;  // exp2(x - inf) = 0: an absent query contributes nothing
      r_d = qi < g.Nq ? g.dsum[(size_t)bh * g.Nq + qi] : 0.f;
    }
  };
  auto store_tile = [&]() {
    tq.store(q_rm, q_tr, tid);
    td.store(do_rm, do_tr, tid);
    if (tid < QT) { ls[tid] = r_l; ls[QT + tid] = r_d; }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  for (int qt = 0; qt < ntiles; ++qt) {
    if (qt + 1 < ntiles) load_tile(qt + 1);
    // S[q][key] and dP[q][key]: rows = the tile's 32 queries, column = this lane's key
    f32x16 s, dp;
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v] = dp[v] = 0.f;
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
      const bf16x8 aq = *reinterpret_cast<const bf16x8 *>(q_rm + a_off + j * 32);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kf[j], s, 0, 0, 0);
      const bf16x8 ad = *reinterpret_cast<const bf16x8 *>(do_rm + a_off + j * 32);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[j], dp, 0, 0, 0);
    }
    float p[16], ds[16];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 l4 = *reinterpret_cast<const float4 *>(ls + 8 * a + 4 * half);
      const float4 d4 = *reinterpret_cast<const float4 *>(ls + QT + 8 * a + 4 * half);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dvv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = 4 * a + e;
        p[v] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[v], c, -lv[e]));
        ds[v] = p[v] * (dp[v] - dvv[e]) * g.scale;
      }
    }
    const bf16x8 p0 = pack_operand(p), p1 = pack_operand(p + 8), d0 = pack_operand(ds), d1 = pack_operand(ds + 8);
#pragma unroll
    for (int t = 0; t < G::NT; ++t) {
      const bf16x8 o0 = tr_operand(do_tr, tr_off + t * (QT * 64), 8 * 64);
      dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o0, p0, dv[t], 0, 0, 0);
      const bf16x8 o1 = tr_operand(do_tr, tr_off + t * (QT * 64) + 16 * 64, 8 * 64);
      dv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, p1, dv[t], 0, 0, 0);
      const bf16x8 q0 = tr_operand(q_tr, tr_off + t * (QT * 64), 8 * 64);
      dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q0, d0, dk[t], 0, 0, 0);
      const bf16x8 q1 = tr_operand(q_tr, tr_off + t * (QT * 64) + 16 * 64, 8 * 64);
      dk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q1, d1, dk[t], 0, 0, 0);
    }
    __syncthreads();
    if (qt + 1 < ntiles) store_tile();
    __syncthreads();
  }
  store_token_rows<D>(dk, g.dk + b * g.dk_bs + (size_t)k_idx * g.dk_ld + h * D, k_ok, half, 1.f);
  store_token_rows<D>(dv, g.dv + b * g.dv_bs + (size_t)k_idx * g.dv_ld + h * D, k_ok, half, 1.f);
}

template <int D>
int launch_fwd(const AttnArgs &a, hipStream_t st) {
  using G = Geo<D>;
  const size_t lds = (size_t)KT * G::ROWB + (size_t)G::NT * KT * 64;
  hipLaunchKernelGGL(attn_fwd<D>, dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), lds, st, a);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
template <int D>
int launch_bwd(const AttnArgs &a, hipStream_t st) {
  using G = Geo<D>;
  const int64_t total = (int64_t)a.B * a.H * a.Nq;
  hipLaunchKernelGGL(attn_dsum, dim3(salun_grid_for(total, 256)), dim3(256), 0, st, a.o, a.d_o, a.dsum, a.o_bs, a.o_ld, a.do_bs,
                     a.do_ld, a.B, a.H, a.Nq, D);
  SALUN_LAUNCH_CHECK();
  const size_t lds_q = 2 * (size_t)KT * G::ROWB + (size_t)G::NT * KT * 64;
  static unsigned long long attr_q = 0, attr_kv = 0;  // one bit per device
  if (!((attr_q >> salun_device_bit()) & 1ull)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_dq<D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_q) != hipSuccess)
      return SALUN_EIO;
    attr_q |= 1ull << salun_device_bit();
  }
  hipLaunchKernelGGL(attn_bwd_dq<D>, dim3((a.Nq + 127) / 128, a.B * a.H), dim3(256), lds_q, st, a);
  SALUN_LAUNCH_CHECK();
  const size_t lds_kv = 2 * (size_t)32 * G::ROWB + 2 * (size_t)G::NT * 32 * 64 + 2 * 32 * sizeof(float);
  if (!((attr_kv >> salun_device_bit()) & 1ull)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_dkv<D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_kv) != hipSuccess)
      return SALUN_EIO;
    attr_kv |= 1ull << salun_device_bit();
  }
  hipLaunchKernelGGL(attn_bwd_dkv<D>, dim3((a.Nk + 127) / 128, a.B * a.H), dim3(256), lds_kv, st, a);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

bool attn_ok(int B, int H, int Nq, int Nk, int D) {
  return B >= 1 && H >= 1 && Nq >= 1 && Nk >= 1 && (D == 8 || D == 16 || D == 32 || D == 40 || D == 64 || D == 80 || D == 160);
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT int salun_attn_supported(int D) { return attn_ok(1, 1, 1, 1, D) ? 1 : 0; }

// o[b,q,h,:] = softmax_k(scale * q[b,q,h,:].k[b,k,h,:]) @ v[b,k,h,:];  lse (optional, [B*H][Nq]) is what the backward needs.
// Every tensor is a [B, tokens, H, D] view given by its batch stride `*_bs` and token stride `*_ld` (elements); the head
// stride is D and the channel stride 1.  Pointers and strides must keep every (token, head) row 16-byte aligned.
SALUN_EXPORT int salun_attn_forward(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *o, float *lse, int B,
                                    int H, int Nq, int Nk, int D, long long q_bs, int q_ld, long long k_bs, int k_ld,
                                    long long v_bs, int v_ld, long long o_bs, int o_ld, double scale, salun_stream_t stream) {
  if (!q || !k || !v || !o || !attn_ok(B, H, Nq, Nk, D)) return SALUN_EINVAL;
  if (!salun_aligned16(q) || !salun_aligned16(k) || !salun_aligned16(v) || !salun_aligned16(o)) return SALUN_EINVAL;
  if ((q_ld | k_ld | v_ld | o_ld) % 8 || (q_bs | k_bs | v_bs | o_bs) % 8) return SALUN_EINVAL;
  // the kernels take the running maximum on the RAW scores and multiply by scale*log2(e) afterwards: that is only
  // monotone for a positive, finite scale (every attention of the three models uses 1/sqrt(d))
  if (!(scale > 0.0) || !(scale < 1e30)) return SALUN_EINVAL;
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = (float)scale;
  hipStream_t st = salun_hip_stream(stream);
  switch (D) {
    case 8: return launch_fwd<8>(a, st);
    case 16: return launch_fwd<16>(a, st);
    case 32: return launch_fwd<32>(a, st);
    case 40: return launch_fwd<40>(a, st);
    case 64: return launch_fwd<64>(a, st);
    case 80: return launch_fwd<80>(a, st);
    default: return launch_fwd<160>(a, st);
  }
}

// dq, dk, dv (contiguous [B, tokens, H, D]) from q, k, v, the forward's o and lse, and d_o.  `dsum` is [B*H][Nq] scratch.
SALUN_EXPORT int salun_attn_backward(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *o,
                                     const uint16_t *d_o, const float *lse, uint16_t *dq, uint16_t *dk, uint16_t *dv,
                                     float *dsum, int B, int H, int Nq, int Nk, int D, long long q_bs, int q_ld, long long k_bs,
                                     int k_ld, long long v_bs, int v_ld, long long o_bs, int o_ld, long long do_bs, int do_ld,
                                     double scale, salun_stream_t stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !dsum || !attn_ok(B, H, Nq, Nk, D)) return SALUN_EINVAL;
  if (!salun_aligned16(q) || !salun_aligned16(k) || !salun_aligned16(v) || !salun_aligned16(o) || !salun_aligned16(d_o) ||
      !salun_aligned16(dq) || !salun_aligned16(dk) || !salun_aligned16(dv))
    return SALUN_EINVAL;
  if ((q_ld | k_ld | v_ld | o_ld | do_ld) % 8 || (q_bs | k_bs | v_bs | o_bs | do_bs) % 8) return SALUN_EINVAL;
  if (!(scale > 0.0) || !(scale < 1e30)) return SALUN_EINVAL;  // see salun_attn_forward
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.o = const_cast<uint16_t *>(o); a.d_o = d_o; a.lse = const_cast<float *>(lse); a.dsum = dsum;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs; a.do_bs = do_bs;
  a.q_ld = q_ld; a.k_ld = k_ld; a.v_ld = v_ld; a.o_ld = o_ld; a.do_ld = do_ld;
  a.dq_ld = a.dk_ld = a.dv_ld = H * D;
  a.dq_bs = (long long)Nq * H * D; a.dk_bs = a.dv_bs = (long long)Nk * H * D;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.scale = (float)scale;
  hipStream_t st = salun_hip_stream(stream);
  switch (D) {
    case 8: return launch_bwd<8>(a, st);
    case 16: return launch_bwd<16>(a, st);
    case 32: return launch_bwd<32>(a, st);
    case 40: return launch_bwd<40>(a, st);
    case 64: return launch_bwd<64>(a, st);
    case 80: return launch_bwd<80>(a, st);
    default: return launch_bwd<160>(a, st);
  }
}
