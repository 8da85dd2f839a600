// salun_topk.hip — K2: global top-k saliency mask by radix select (gfx950 / CDNA4).
//
// The reference ranks every weight with two full argsorts per threshold
// (Classification/generate_mask.py:57-64) only to compare the rank with k.  Here the
// k-th largest |acc| is located with three histogram passes over a monotone 31-bit
// integer key (11 + 10 + 10 bit digits, most significant first), all thresholds
// sharing every pass, and one pass writes the nk u8 masks.  Integer-only, hence
// bit-exact and independent of launch geometry.
//
//   key(x) = 0                       if x is NaN   (ranks after every number)
//          = (bits(x) & 0x7fffffff) + 1  otherwise (|x| as an ordered integer)
//   mask_j[i] = 1  iff  key_i > tau_j, or key_i == tau_j and i is among the first
//               r_j indices holding tau_j   (stable tie rule)
//
// LDS: per-workgroup histograms (u32, LDS atomics) flushed once with 64-bit global
// atomics; pass 1/2 histograms are kept per *group* of thresholds that share the
// already-fixed prefix, so 10 thresholds cost one read of the vector per pass.
//
// One threshold over a large vector (N >= 2^25: the DDPM and SD masks) takes the sampled
// single-pass route first — rank a 2^20-element sample, bracket the threshold, ONE pass
// over the vector (final mask outside the bracket, ~1 % candidates compacted), exact
// select among the candidates, fix-up — and falls back to the full scan on the device
// when the bracket misses; same masks bit for bit (see k_sample ... k_sampled_finish).
#include "salun_common.h"
#include <cmath>
#include <cstdlib>

namespace {

constexpr int MAXK = SALUN_MAX_THRESHOLDS;
constexpr int D0_BINS = 2048;  // key >> 20
constexpr int D1_BINS = 1024;  // (key >> 10) & 1023
constexpr int D2_BINS = 1024;  // key & 1023
constexpr int CHUNK_VEC = 4 * SALUN_BLOCK;  // float4 per chunk (4 sub-vectors per lane)
constexpr int CHUNK = CHUNK_VEC * 4;        // 4096 elements: the tie-ordering granule
constexpr int HIST_MAX_GRID = 512;

enum Mode : uint32_t { MODE_NONE = 0, MODE_ALL = 1, MODE_GE = 2, MODE_ORDERED = 3 };

typedef unsigned long long u64;

struct TopkState {
  u64 hist0[D0_BINS];
  u64 hist1[MAXK][D1_BINS];
  u64 hist2[MAXK][D2_BINS];
  long long k[MAXK];
  u64 rem[MAXK];         // how many still to take inside the currently selected bin
  u64 ceq[MAXK];         // population of key == tau
  uint32_t prefix0[MAXK];
  uint32_t prefix1[MAXK];  // 21-bit prefix (d0 << 10 | d1)
  uint32_t tau[MAXK];
  uint32_t mode[MAXK];
  uint32_t group0_of[MAXK];  // threshold -> pass-1 histogram slot
  uint32_t group1_of[MAXK];  // threshold -> pass-2 histogram slot
  uint32_t group0_prefix[MAXK];
  uint32_t group1_prefix[MAXK];
  uint32_t ngroups0, ngroups1;
  uint32_t any_ordered;
  uint32_t nk;
  uint32_t skip;           // set by the sampled path on success: the full-scan passes below return immediately
  uint32_t use_kdev;       // take k from kdev[] (computed on the device) instead of the launch argument
  long long kdev[MAXK];
  uint8_t lut0[D0_BINS];  // d0 -> pass-1 slot + 1 (0 = not a boundary bin)
};

struct KList {
  long long k[MAXK];
  int nk;
};
struct MaskPtrs {
  uint8_t *m[MAXK];
};

__device__ __forceinline__ uint32_t key_of(float x) {
  const uint32_t b = __float_as_uint(x) & 0x7FFFFFFFu;
  return (b > 0x7F800000u) ? 0u : b + 1u;
}
constexpr uint32_t KEY_SKIP = 0xFFFFFFFFu;  // out-of-range lane marker (never a real key)

// Keys of float4 #v; elements at or beyond n become KEY_SKIP.
template <bool ALIGNED>
__device__ __forceinline__ void load_keys(const float *__restrict__ acc, int64_t v, int64_t n, uint32_t k[4]) {
  const int64_t i = v << 2;
  if (ALIGNED && i + 3 < n) {
    const float4 x = reinterpret_cast<const float4 *>(acc)[v];
    k[0] = key_of(x.x); k[1] = key_of(x.y); k[2] = key_of(x.z); k[3] = key_of(x.w);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = (i + e < n) ? key_of(acc[i + e]) : KEY_SKIP;
  }
}

// ---------------------------------------------------------------- pass 0 histogram
template <bool ALIGNED>
__global__ __launch_bounds__(SALUN_BLOCK) void k_hist0(const float *__restrict__ acc, int64_t n, TopkState *st) {
  __shared__ uint32_t h[D0_BINS];
  if (st->skip) return;
  for (int i = threadIdx.x; i < D0_BINS; i += SALUN_BLOCK) h[i] = 0;
  __syncthreads();
  const int64_t nvec = (n + 3) >> 2;
  const int64_t nchunk = (nvec + CHUNK_VEC - 1) / CHUNK_VEC;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) load_keys<ALIGNED>(acc, v, n, k[u]);
      else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k[u][e] != KEY_SKIP) atomicAdd(&h[k[u][e] >> 20], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D0_BINS; i += SALUN_BLOCK)
    if (h[i]) atomicAdd(&st->hist0[i], (u64)h[i]);
}

// ------------------------------------------------------- pass 1 / pass 2 histograms
// LEVEL 1: bins = d1 of keys whose d0 is a boundary bin; LEVEL 2: bins = d2 of keys
// whose 21-bit prefix is a boundary prefix.  Dynamic LDS: lut0 (2 KiB) + nk * 1024 u32.
template <int LEVEL, bool ALIGNED>
__global__ __launch_bounds__(SALUN_BLOCK) void k_hist12(const float *__restrict__ acc, int64_t n, TopkState *st) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  if (st->skip) return;
  const uint32_t ngroups = (LEVEL == 1) ? st->ngroups0 : st->ngroups1;
  if (ngroups == 0) return;  // every threshold is trivial (k <= 0 or k >= n)
  uint8_t *lut = reinterpret_cast<uint8_t *>(lds);  // 2048 bytes
  uint32_t *h = lds + D0_BINS / 4;                  // ngroups (<= nk) * 1024 counters
  __shared__ uint32_t gprefix[MAXK];
  for (uint32_t i = threadIdx.x; i < ngroups * 1024u; i += SALUN_BLOCK) h[i] = 0;
  for (int i = threadIdx.x; i < D0_BINS; i += SALUN_BLOCK) lut[i] = st->lut0[i];
  if (threadIdx.x < MAXK) gprefix[threadIdx.x] = st->group1_prefix[threadIdx.x];
  __syncthreads();
  const int64_t nvec = (n + 3) >> 2;
  const int64_t nchunk = (nvec + CHUNK_VEC - 1) / CHUNK_VEC;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) load_keys<ALIGNED>(acc, v, n, k[u]);
      else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t key = k[u][e];
        if (key == KEY_SKIP) continue;
        const uint32_t g0 = lut[key >> 20];
        if (!g0) continue;
        if (LEVEL == 1) {
          atomicAdd(&h[(g0 - 1) * 1024u + ((key >> 10) & 1023u)], 1u);
        } else {
          const uint32_t pre = key >> 10;
          for (uint32_t g = 0; g < ngroups; ++g)
            if (gprefix[g] == pre) { atomicAdd(&h[g * 1024u + (key & 1023u)], 1u); break; }
        }
      }
  }
  __syncthreads();
  u64 *gh = (LEVEL == 1) ? &st->hist1[0][0] : &st->hist2[0][0];
  for (uint32_t i = threadIdx.x; i < ngroups * 1024u; i += SALUN_BLOCK)
    if (h[i]) atomicAdd(&gh[i], (u64)h[i]);
}

// -------------------------------------------------------------------- selection
// One 1024-thread workgroup, wave w serves threshold w.  Bins are walked from the
// top: lane l owns the l-th highest slice; a 64-lane exclusive scan of the slice
// totals finds the slice holding the k-th element, that lane walks its bins.
__device__ __forceinline__ u64 wave_excl_scan_u64(u64 v, int lane) {
  u64 inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u64 t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  return inc - v;
}

template <int LEVEL>
__global__ __launch_bounds__(1024) void k_select(TopkState *st, int64_t n, KList kl) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nk = kl.nk;
  constexpr int NBINS = (LEVEL == 0) ? D0_BINS : 1024;
  constexpr int PER_LANE = NBINS / 64;
  if (st->skip) return;
  __shared__ uint32_t s_bin[MAXK];
  __shared__ u64 s_rem[MAXK];
  __shared__ u64 s_cnt[MAXK];
  if (LEVEL == 0 && threadIdx.x < MAXK) {
    uint32_t mode = MODE_NONE;
    long long k = 0;
    if ((int)threadIdx.x < nk) {
      k = st->use_kdev ? st->kdev[threadIdx.x] : kl.k[threadIdx.x];
      if (k <= 0) { k = 0; mode = MODE_NONE; }
      else if (k > n) { k = n; mode = MODE_ALL; }  // k == n runs the select: its threshold (the minimum) is real
      else mode = MODE_GE;  // provisional: refined after the last level
    }
    st->k[threadIdx.x] = k;
    st->mode[threadIdx.x] = mode;
    st->rem[threadIdx.x] = (u64)k;
    if (threadIdx.x == 0) st->nk = nk;
  }
  __syncthreads();
  if (wave < nk) {
    const uint32_t mode = st->mode[wave];
    if (mode >= MODE_GE) {
      const u64 *hist = (LEVEL == 0) ? st->hist0
                        : (LEVEL == 1) ? st->hist1[st->group0_of[wave]]
                                       : st->hist2[st->group1_of[wave]];
      const u64 want = st->rem[wave];  // 1 <= want <= population of this histogram
      u64 mine = 0;
      for (int j = 0; j < PER_LANE; ++j) mine += hist[NBINS - 1 - (lane * PER_LANE + j)];
      const u64 before = wave_excl_scan_u64(mine, lane);
      if (before < want && want <= before + mine) {
        u64 cum = before;
        for (int j = 0; j < PER_LANE; ++j) {
          const int bin = NBINS - 1 - (lane * PER_LANE + j);
          const u64 c = hist[bin];
          if (want <= cum + c) {
            s_bin[wave] = (uint32_t)bin;
            s_rem[wave] = want - cum;
            s_cnt[wave] = c;
            break;
          }
          cum += c;
        }
      }
    }
  }
  __syncthreads();
  // Publish + group thresholds that fell into the same bin (they share the next histogram).
  if (threadIdx.x == 0) {
    uint32_t ng = 0, any_ordered = 0;
    for (int i = 0; i < nk; ++i) {
      if (st->mode[i] < MODE_GE) continue;
      st->rem[i] = s_rem[i];
      if (LEVEL == 0) {
        st->prefix0[i] = s_bin[i];
        uint32_t g = 0;
        for (; g < ng; ++g) if (st->group0_prefix[g] == s_bin[i]) break;
        if (g == ng) st->group0_prefix[ng++] = s_bin[i];
        st->group0_of[i] = g;
      } else if (LEVEL == 1) {
        const uint32_t pre = (st->prefix0[i] << 10) | s_bin[i];
        st->prefix1[i] = pre;
        uint32_t g = 0;
        for (; g < ng; ++g) if (st->group1_prefix[g] == pre) break;
        if (g == ng) st->group1_prefix[ng++] = pre;
        st->group1_of[i] = g;
      } else {
        st->tau[i] = (st->prefix1[i] << 10) | s_bin[i];
        st->ceq[i] = s_cnt[i];
        if (s_rem[i] != s_cnt[i]) { st->mode[i] = MODE_ORDERED; any_ordered = 1; }
      }
    }
    if (LEVEL == 0) st->ngroups0 = ng;
    if (LEVEL == 1) st->ngroups1 = ng;
    if (LEVEL == 2) st->any_ordered = any_ordered;
  }
  if (LEVEL == 0) {
    __syncthreads();
    for (int i = threadIdx.x; i < D0_BINS; i += 1024) st->lut0[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t ng = st->ngroups0;
      for (uint32_t g = 0; g < ng; ++g) st->lut0[st->group0_prefix[g]] = (uint8_t)(g + 1);
    }
  }
}

// ------------------------------------------------------- ordered ties (rare path)
// Only when some threshold splits a run of equal keys: per-chunk populations of
// key == tau, exclusive-scanned over chunks, give every chunk the number of equal
// keys that precede it in flat-index order.
__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const uint32_t r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return r;
}

template <bool ALIGNED>
__global__ __launch_bounds__(SALUN_BLOCK) void k_tie_count(const float *__restrict__ acc, int64_t n,
                                                           const TopkState *st, u64 *tie /*[nk][nchunk]*/,
                                                           int64_t nchunk) {
  if (st->skip || !st->any_ordered) return;
  __shared__ uint32_t lds4[4];
  const int nk = (int)st->nk;
  const int64_t nvec = (n + 3) >> 2;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) load_keys<ALIGNED>(acc, v, n, k[u]);
      else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
    }
    for (int j = 0; j < nk; ++j) {
      if (st->mode[j] != MODE_ORDERED) continue;
      const uint32_t tau = st->tau[j];
      uint32_t cnt = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) cnt += (k[u][e] == tau);
      const uint32_t tot = block_sum_u32(cnt, lds4);
      if (threadIdx.x == 0) tie[(int64_t)j * nchunk + c] = tot;
    }
  }
}

// One workgroup per threshold: in-place exclusive scan over chunks.
__global__ __launch_bounds__(SALUN_BLOCK) void k_tie_scan(const TopkState *st, u64 *tie, int64_t nchunk) {
  if (st->skip || !st->any_ordered) return;
  const int j = blockIdx.x;
  if (st->mode[j] != MODE_ORDERED) return;
  __shared__ u64 s_wave[4];
  __shared__ u64 s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  u64 *row = tie + (int64_t)j * nchunk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t base = 0; base < nchunk; base += SALUN_BLOCK) {
    const int64_t i = base + threadIdx.x;
    const u64 v = (i < nchunk) ? row[i] : 0;
    const u64 ex = wave_excl_scan_u64(v, lane);
    if (lane == 63) s_wave[wave] = ex + v;
    __syncthreads();
    u64 woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const u64 carry = s_carry;
    if (i < nchunk) row[i] = carry + woff + ex;
    __syncthreads();
    if (threadIdx.x == SALUN_BLOCK - 1) s_carry = carry + woff + ex + v;
    __syncthreads();
  }
}

// ------------------------------------------------------------------ mask write
// Block exclusive scan of one u32 per thread (256 threads); returns the exclusive
// prefix, *total gets the block total.
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *lds4, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return woff + inc - v;
}

template <bool ALIGNED, bool MALIGNED>
__global__ __launch_bounds__(SALUN_BLOCK) void k_write_masks(const float *__restrict__ acc, int64_t n,
                                                             const TopkState *st, const u64 *tie,
                                                             int64_t nchunk, MaskPtrs out) {
  __shared__ uint32_t s_thr[MAXK];   // key >= thr  => 1   (fast modes)
  __shared__ uint32_t s_mode[MAXK];
  __shared__ uint32_t s_tau[MAXK];
  __shared__ u64 s_budget[MAXK];
  __shared__ uint32_t lds4[4];
  if (st->skip) return;
  const int nk = (int)st->nk;
  if ((int)threadIdx.x < nk) {
    const uint32_t mode = st->mode[threadIdx.x];
    s_mode[threadIdx.x] = mode;
    s_tau[threadIdx.x] = st->tau[threadIdx.x];
    s_budget[threadIdx.x] = st->rem[threadIdx.x];
    // NONE: nothing passes (real keys <= 0x7F800001); ALL: everything passes;
    // GE: every key equal to tau is inside the budget; ORDERED: strictly greater passes here.
    s_thr[threadIdx.x] = (mode == MODE_NONE) ? 0xFFFFFFFEu
                         : (mode == MODE_ALL) ? 0u
                         : (mode == MODE_GE) ? st->tau[threadIdx.x]
                                             : st->tau[threadIdx.x] + 1u;
  }
  __syncthreads();
  const int64_t nvec = (n + 3) >> 2;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) load_keys<ALIGNED>(acc, v, n, k[u]);
      else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
    }
    for (int j = 0; j < nk; ++j) {
      const uint32_t thr = s_thr[j];
      uint32_t bits[4];  // 4 mask bytes per sub-vector
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bits[u] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          bits[u] |= (uint32_t)(k[u][e] != KEY_SKIP && k[u][e] >= thr) << (8 * e);
      }
      if (s_mode[j] == MODE_ORDERED) {  // workgroup-uniform branch
        const uint32_t tau = s_tau[j];
        u64 before = tie[(int64_t)j * nchunk + c];  // equal keys in earlier chunks
        const u64 budget = s_budget[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // in-chunk index order: sub-vector, lane, element
          uint32_t cnt = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) cnt += (k[u][e] == tau);
          uint32_t total;
          u64 pos = before + block_excl_scan_u32(cnt, lds4, &total);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (k[u][e] == tau) {
              if (pos < budget) bits[u] |= 1u << (8 * e);
              ++pos;
            }
          before += total;
        }
      }
      uint8_t *mj = out.m[j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
        if (v >= nvec) continue;
        const int64_t i = v << 2;
        if (MALIGNED && i + 3 < n) {
          reinterpret_cast<uint32_t *>(mj)[v] = bits[u];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e < n) mj[i + e] = (uint8_t)((bits[u] >> (8 * e)) & 1u);
        }
      }
    }
  }
}

__global__ void k_export_tau(const TopkState *st, int nk, float *out) {
  const int j = threadIdx.x;
  if (j >= nk) return;
  const uint32_t mode = st->mode[j];
  float v;
  if (mode == MODE_NONE) v = __uint_as_float(0x7F800000u);        // +inf: nothing selected
  else if (mode == MODE_ALL) v = -1.0f;                             // below every |x|
  else v = st->tau[j] ? __uint_as_float(st->tau[j] - 1u) : __uint_as_float(0x7FC00000u);
  out[j] = v;
}

// ------------------------------------------------------------- format converters
__global__ __launch_bounds__(SALUN_BLOCK) void k_u8_to_i64(const uint8_t *__restrict__ m, long long *__restrict__ out,
                                                           int64_t n, int aligned) {
  const int64_t nvec = aligned ? (n >> 2) : 0;
  for (int64_t v = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * SALUN_BLOCK) {
    const uint32_t b = reinterpret_cast<const uint32_t *>(m)[v];
    longlong2 lo, hi;
    lo.x = (b & 0xFFu) != 0; lo.y = (b & 0xFF00u) != 0;
    hi.x = (b & 0xFF0000u) != 0; hi.y = (b & 0xFF000000u) != 0;
    reinterpret_cast<longlong2 *>(out)[2 * v] = lo;
    reinterpret_cast<longlong2 *>(out)[2 * v + 1] = hi;
  }
  for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * SALUN_BLOCK)
    out[i] = m[i] != 0;
}

__global__ __launch_bounds__(SALUN_BLOCK) void k_i64_to_u8(const long long *__restrict__ m, uint8_t *__restrict__ out,
                                                           int64_t n, int aligned) {
  const int64_t nvec = aligned ? (n >> 2) : 0;
  for (int64_t v = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * SALUN_BLOCK) {
    const longlong2 lo = reinterpret_cast<const longlong2 *>(m)[2 * v];
    const longlong2 hi = reinterpret_cast<const longlong2 *>(m)[2 * v + 1];
    const uint32_t b = (uint32_t)(lo.x != 0) | ((uint32_t)(lo.y != 0) << 8) | ((uint32_t)(hi.x != 0) << 16) |
                       ((uint32_t)(hi.y != 0) << 24);
    reinterpret_cast<uint32_t *>(out)[v] = b;
  }
  for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * SALUN_BLOCK)
    out[i] = m[i] != 0;
}

__global__ __launch_bounds__(SALUN_BLOCK) void k_popcount_partial(const uint8_t *__restrict__ m, int64_t n,
                                                                  u64 *__restrict__ partial, int aligned) {
  __shared__ u64 lds[4];
  u64 s = 0;
  const int64_t nvec = aligned ? (n >> 4) : 0;  // 16 bytes per lane
  for (int64_t v = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * SALUN_BLOCK) {
    const uint4 b = reinterpret_cast<const uint4 *>(m)[v];
    // bytes are 0/1 by contract; tolerate any non-zero byte
    auto nz = [](uint32_t w) -> uint32_t {
      return ((w & 0xFFu) != 0) + ((w & 0xFF00u) != 0) + ((w & 0xFF0000u) != 0) + ((w & 0xFF000000u) != 0);
    };
    s += nz(b.x) + nz(b.y) + nz(b.z) + nz(b.w);
  }
  for (int64_t i = (nvec << 4) + (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * SALUN_BLOCK)
    s += m[i] != 0;
  s = salun_wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(SALUN_BLOCK) void k_sum_partials_i64(const u64 *__restrict__ partial, int count,
                                                                   long long *__restrict__ out) {
  __shared__ u64 lds[4];
  u64 s = 0;
  for (int i = threadIdx.x; i < count; i += SALUN_BLOCK) s += partial[i];
  s = salun_wave_sum_u64(s);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (long long)(lds[0] + lds[1] + lds[2] + lds[3]);
}

// =====================================================================================================
// Sampled single-pass path (one threshold, N >= SAMPLED_MIN_N): the three histogram passes above read the vector
// three times before the write pass reads it a fourth time.  For SD-sized vectors (3.4 GB) that is the whole cost.
// Here a strided sample of S = 2^20 values is ranked first (same radix select, 4 MB); its order statistics
// 8 sigma either side of the target rank bracket the true threshold:  lo <= tau <= hi  except with
// probability ~1e-15 for exchangeable data.  ONE pass over the vector then
//     writes mask = [key > hi]                      (final for everything outside the bracket)
//     counts c_gt = #{key > hi}
//     compacts the ~0.8 % of elements with lo <= key <= hi (|value| and flat index)
// and the exact threshold is the (k - c_gt)-th largest of the compacted values (radix select over ~N/126 elements),
// after which those candidates' mask bytes are fixed up.  The result is the same function of the input as the
// full-scan path (bit-exact masks).  Whenever the bracket misses (adversarial periodic data), the candidate buffer
// overflows or the threshold splits a run of equal keys, nothing is published and the full-scan passes run as
// before — they start with `if (st->skip) return`.
constexpr int64_t SAMPLED_MIN_N = int64_t(1) << 25;  // measured cross-over with the full scan: ~30 M elements (tools/topk_scale.py)
constexpr int SAMPLE_LOG2 = 20;

constexpr int TIE_CAP = 4096;   // tied candidates that can be ordered by index in one workgroup

struct SampCounters {
  u64 c_gt;          // elements strictly above the bracket
  u64 m;             // candidates kept
  uint32_t overflow; // some workgroup ran out of its slab of the candidate buffer
  uint32_t ok;
  uint32_t tie_n;    // candidates equal to the threshold (ordered-tie case only)
};
struct SampExtra {
  TopkState samp;    // select on the sample: threshold 0 = upper bracket, 1 = lower bracket
  TopkState cand;    // select on the candidates
  SampCounters cnt;
  uint32_t tie_idx[TIE_CAP];
};

__global__ __launch_bounds__(SALUN_BLOCK) void k_sample(const float *__restrict__ acc, int64_t stride, int64_t S,
                                                        float *__restrict__ out) {
  // one element per stride window at a hashed offset: a fixed offset would lock onto periodic structure of the
  // flat vector (e.g. always the centre tap of 3x3 kernels when the stride is a multiple of 9)
  const int64_t s = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x;
  if (s < S) out[s] = acc[s * stride + (int64_t)(salun_splitmix64((uint64_t)s) % (uint64_t)stride)];
}

// The one full pass.  Every workgroup appends its candidates to its own slab of the candidate buffer (LDS counter:
// no contended global atomic); unused slab entries keep their NaN fill, which ranks after every number.
template <bool ALIGNED, bool MALIGNED>
__global__ __launch_bounds__(SALUN_BLOCK) void k_sampled_main(const float *__restrict__ acc, int64_t n,
                                                              const TopkState *samp, int no_hi, int no_lo,
                                                              uint8_t *__restrict__ mask, float *__restrict__ candv,
                                                              uint32_t *__restrict__ candi, u64 slab_cap,
                                                              SampCounters *cnt) {
  __shared__ uint32_t s_count;
  __shared__ u64 s_w[4];
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const uint32_t hi_key = no_hi ? 0xFFFFFFFFu : samp->tau[0];
  const uint32_t lo_key = no_lo ? 0u : samp->tau[1];
  const int lane = threadIdx.x & 63;
  const int64_t nvec = (n + 3) >> 2;
  const int64_t nchunk = (nvec + CHUNK_VEC - 1) / CHUNK_VEC;
  float *myv = candv + (u64)blockIdx.x * slab_cap;
  uint32_t *myi = candi + (u64)blockIdx.x * slab_cap;
  uint32_t my_gt = 0;
  u64 blk_gt = 0;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) load_keys<ALIGNED>(acc, v, n, k[u]);
      else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
    }
    uint32_t ncand = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      uint32_t bits = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t key = k[u][e];
        const bool real = key != KEY_SKIP;
        const bool gt = real && key > hi_key;
        bits |= (uint32_t)gt << (8 * e);
        my_gt += gt;
        ncand += (real && !gt && key >= lo_key);
      }
      const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
      if (v < nvec) {
        const int64_t i = v << 2;
        if (MALIGNED && i + 3 < n) {
          reinterpret_cast<uint32_t *>(mask)[v] = bits;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (i + e < n) mask[i + e] = (uint8_t)((bits >> (8 * e)) & 1u);
        }
      }
    }
    // wave-aggregated append into the workgroup's slab
    uint32_t inc = ncand;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    const uint32_t wave_total = __shfl(inc, 63, 64);
    if (wave_total) {
      uint32_t base = 0;
      if (lane == 63) base = atomicAdd(&s_count, wave_total);
      base = __shfl(base, 63, 64);
      u64 pos = (u64)base + (inc - ncand);
      if (ncand) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t key = k[u][e];
            if (key != KEY_SKIP && key <= hi_key && key >= lo_key) {
              if (pos < slab_cap) {
                myv[pos] = key ? __uint_as_float(key - 1u) : __uint_as_float(0x7FC00000u);  // |x| (NaN for key 0)
                myi[pos] = (uint32_t)(((c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x) << 2) + e);
              }
              ++pos;
            }
          }
      }
    }
    if (my_gt > 0x7FFFFFFFu) { blk_gt += my_gt; my_gt = 0; }
  }
  blk_gt += my_gt;
  u64 v = salun_wave_sum_u64(blk_gt);
  if (lane == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    if (tot) atomicAdd(&cnt->c_gt, tot);
    const u64 mine = s_count;
    if (mine > slab_cap) cnt->overflow = 1;
    if (mine) atomicAdd(&cnt->m, mine < slab_cap ? mine : slab_cap);
  }
}

__global__ void k_sampled_prep(SampExtra *ex, long long k) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const u64 m = ex->cnt.m;
  const long long r = k - (long long)ex->cnt.c_gt;
  const bool ok = !ex->cnt.overflow && r >= 1 && (u64)r <= m;
  ex->cnt.ok = ok ? 1u : 0u;
  ex->cand.kdev[0] = ok ? r : 1;
  ex->cand.use_kdev = 1;
}

// candidates above the exact threshold -> 1; equal to it -> 1 (all ties inside the budget) or collected for ordering
__global__ __launch_bounds__(SALUN_BLOCK) void k_sampled_fixup(const float *__restrict__ candv,
                                                               const uint32_t *__restrict__ candi, SampExtra *ex,
                                                               u64 cap, uint8_t *__restrict__ mask) {
  if (!ex->cnt.ok) return;
  const uint32_t tau = ex->cand.tau[0];
  const uint32_t mode = ex->cand.mode[0];
  if (tau == 0 || mode < MODE_GE) return;  // NaN threshold: slab padding is indistinguishable -> full scan
  for (u64 i = (u64)blockIdx.x * SALUN_BLOCK + threadIdx.x; i < cap; i += (u64)gridDim.x * SALUN_BLOCK) {
    const uint32_t key = key_of(candv[i]);
    if (key > tau || (key == tau && mode == MODE_GE)) {
      mask[candi[i]] = 1;
    } else if (key == tau) {
      const uint32_t pos = atomicAdd(&ex->cnt.tie_n, 1u);
      if (pos < (uint32_t)TIE_CAP) ex->tie_idx[pos] = candi[i];
    }
  }
}

// one workgroup: order the tied candidates by flat index, admit the first `rem`, publish, switch the full scan off
__global__ __launch_bounds__(1024) void k_sampled_finish(SampExtra *ex, TopkState *main_state, long long k,
                                                         uint8_t *__restrict__ mask) {
  __shared__ uint32_t s_idx[TIE_CAP];
  if (!ex->cnt.ok) return;
  const uint32_t tau = ex->cand.tau[0];
  const uint32_t mode = ex->cand.mode[0];
  if (tau == 0 || mode < MODE_GE) return;
  if (mode == MODE_ORDERED) {
    const uint32_t T = ex->cnt.tie_n;
    if (T > (uint32_t)TIE_CAP || (u64)T != ex->cand.ceq[0]) return;  // too many ties for this path -> full scan
    const u64 budget = ex->cand.rem[0];
    for (uint32_t t = threadIdx.x; t < T; t += 1024) s_idx[t] = ex->tie_idx[t];
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < T; t += 1024) {
      const uint32_t mine = s_idx[t];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < T; ++j) rank += (s_idx[j] < mine);
      if ((u64)rank < budget) mask[mine] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // publish for salun_mask_topk_thresholds; the full-scan kernels see skip and return
    main_state->nk = 1;
    main_state->k[0] = k;
    main_state->mode[0] = MODE_GE;
    main_state->tau[0] = tau;
    main_state->skip = 1;
  }
}

inline int64_t sampled_min_n() {
  const char *e = getenv("SALUN_TOPK_SAMPLED_MIN");  // test hook: exercise the sampled path at small sizes
  if (e && *e) return (int64_t)atoll(e);
  return SAMPLED_MIN_N;
}
inline bool sampled_applies(int64_t n, int nk) {
  return nk == 1 && n >= sampled_min_n() && n >= 65536 && n < (int64_t(1) << 32);
}
inline int64_t sample_size(int64_t n) {
  int64_t S = int64_t(1) << SAMPLE_LOG2;
  while (S > 4096 && S * 16 > n) S >>= 1;
  return S;
}
inline u64 cand_capacity(int64_t n) { return (u64)(((n / 64 + 4095) / 4096) * 4096 + 4096); }
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

inline size_t state_bytes() { return (sizeof(TopkState) + 255) & ~size_t(255); }

// the three histogram + selection rounds over `data` (state zeroed by the caller): fills state->tau / mode / rem
inline int run_select(const float *data, int64_t n, const KList &kl, TopkState *state, hipStream_t st) {
  const bool aligned = salun_aligned16(data);
  const int grid = salun_grid_for(n, CHUNK);
  // Histogram kernels end with one global atomic per non-empty bin per workgroup, all workgroups hitting the
  // same few hundred addresses: keep that chain short (2 workgroups per CU) — the read side still has
  // 8 waves x 4 KiB in flight per CU.
  const int hgrid = grid < HIST_MAX_GRID ? grid : HIST_MAX_GRID;
  // dynamic LDS of the pass-1/2 histogram kernels: 2 KiB lut + one 4 KiB histogram per threshold
  const size_t lds_bytes = D0_BINS + sizeof(uint32_t) * (size_t)kl.nk * 1024;
  if (aligned) hipLaunchKernelGGL(k_hist0<true>, dim3(hgrid), dim3(SALUN_BLOCK), 0, st, data, n, state);
  else hipLaunchKernelGGL(k_hist0<false>, dim3(hgrid), dim3(SALUN_BLOCK), 0, st, data, n, state);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_select<0>, dim3(1), dim3(1024), 0, st, state, n, kl);
  SALUN_LAUNCH_CHECK();
  if (aligned) hipLaunchKernelGGL((k_hist12<1, true>), dim3(hgrid), dim3(SALUN_BLOCK), lds_bytes, st, data, n, state);
  else hipLaunchKernelGGL((k_hist12<1, false>), dim3(hgrid), dim3(SALUN_BLOCK), lds_bytes, st, data, n, state);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_select<1>, dim3(1), dim3(1024), 0, st, state, n, kl);
  SALUN_LAUNCH_CHECK();
  if (aligned) hipLaunchKernelGGL((k_hist12<2, true>), dim3(hgrid), dim3(SALUN_BLOCK), lds_bytes, st, data, n, state);
  else hipLaunchKernelGGL((k_hist12<2, false>), dim3(hgrid), dim3(SALUN_BLOCK), lds_bytes, st, data, n, state);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_select<2>, dim3(1), dim3(1024), 0, st, state, n, kl);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
inline int64_t chunks_of(int64_t n) { return (n + CHUNK - 1) / CHUNK; }

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT size_t salun_mask_topk_workspace_bytes(int64_t n, int nk) {
  if (n < 0 || nk < 1 || nk > MAXK) return 0;
  size_t b = align256(state_bytes() + sizeof(u64) * (size_t)nk * (size_t)(chunks_of(n) + 1));
  if (sampled_applies(n, nk))
    b += align256(sizeof(SampExtra)) + align256(sizeof(float) * (size_t)sample_size(n)) +
         2 * align256(sizeof(float) * (size_t)cand_capacity(n));
  return b;
}

SALUN_EXPORT int salun_mask_topk(const float *acc, int64_t n, const int64_t *ks, int nk,
                                 uint8_t *const *masks_out, void *ws, size_t ws_bytes,
                                 salun_stream_t stream) {
  if (n < 0 || nk < 1 || nk > MAXK || !ks || !masks_out || !ws || (n > 0 && !acc)) return SALUN_EINVAL;
  for (int j = 0; j < nk; ++j)
    if (n > 0 && !masks_out[j]) return SALUN_EINVAL;
  if (ws_bytes < salun_mask_topk_workspace_bytes(n, nk)) return SALUN_ENOSPC;
  if (n == 0) return SALUN_OK;
  hipStream_t st = salun_hip_stream(stream);
  TopkState *state = static_cast<TopkState *>(ws);
  u64 *tie = reinterpret_cast<u64 *>(static_cast<char *>(ws) + state_bytes());
  const int64_t nchunk = chunks_of(n);
  KList kl;
  MaskPtrs mp;
  kl.nk = nk;
  bool maligned = true;
  for (int j = 0; j < MAXK; ++j) {
    kl.k[j] = (j < nk) ? (long long)ks[j] : 0;
    mp.m[j] = (j < nk) ? masks_out[j] : nullptr;
    if (j < nk && !salun_aligned4(masks_out[j])) maligned = false;
  }
  const bool aligned = salun_aligned16(acc);
  if (hipMemsetAsync(state, 0, sizeof(TopkState), st) != hipSuccess) return SALUN_EIO;
  const int grid = salun_grid_for(n, CHUNK);

  if (sampled_applies(n, nk) && ks[0] > 0 && ks[0] < n) {
    // ---- sampled single-pass attempt (see the comment above k_sample); publishes and sets state->skip on success
    char *base = static_cast<char *>(ws) + align256(state_bytes() + sizeof(u64) * (size_t)nk * (size_t)(nchunk + 1));
    SampExtra *ex = reinterpret_cast<SampExtra *>(base);
    const int64_t S = sample_size(n);
    const u64 cap = cand_capacity(n);
    float *samp = reinterpret_cast<float *>(base + align256(sizeof(SampExtra)));
    float *candv = reinterpret_cast<float *>(reinterpret_cast<char *>(samp) + align256(sizeof(float) * (size_t)S));
    uint32_t *candi = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(candv) + align256(sizeof(float) * cap));
    if (hipMemsetAsync(ex, 0, sizeof(SampExtra), st) != hipSuccess) return SALUN_EIO;
    if (hipMemsetAsync(candv, 0xFF, sizeof(float) * cap, st) != hipSuccess) return SALUN_EIO;  // NaN: ranks last
    const double pfrac = (double)ks[0] / (double)n;
    const double sigma = std::sqrt((double)S * pfrac * (1.0 - pfrac));
    const int64_t margin = (int64_t)std::ceil(8.0 * sigma) + 64;
    const int64_t ks_s = (int64_t)std::llround(pfrac * (double)S);
    const int no_hi = ks_s - margin < 1, no_lo = ks_s + margin > S;
    KList k2;
    k2.nk = 2;
    for (int j = 0; j < MAXK; ++j) k2.k[j] = 0;
    k2.k[0] = no_hi ? 1 : ks_s - margin;   // descending rank of the upper bracket in the sample
    k2.k[1] = no_lo ? S : ks_s + margin;   // ... and of the lower bracket
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((S + SALUN_BLOCK - 1) / SALUN_BLOCK)), dim3(SALUN_BLOCK), 0, st, acc,
                       n / S, S, samp);
    SALUN_LAUNCH_CHECK();
    int rc = run_select(samp, S, k2, &ex->samp, st);
    if (rc != SALUN_OK) return rc;
    const int mgrid = grid < 2048 ? grid : 2048;
    const u64 slab_cap = cap / (u64)mgrid;
#define SALUN_SMAIN(A, M)                                                                                          \
  hipLaunchKernelGGL((k_sampled_main<A, M>), dim3(mgrid), dim3(SALUN_BLOCK), 0, st, acc, n, &ex->samp, no_hi, no_lo, \
                     mp.m[0], candv, candi, slab_cap, &ex->cnt)
    if (aligned && maligned) SALUN_SMAIN(true, true);
    else if (aligned) SALUN_SMAIN(true, false);
    else if (maligned) SALUN_SMAIN(false, true);
    else SALUN_SMAIN(false, false);
#undef SALUN_SMAIN
    SALUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sampled_prep, dim3(1), dim3(64), 0, st, ex, (long long)ks[0]);
    SALUN_LAUNCH_CHECK();
    KList k1;
    k1.nk = 1;
    for (int j = 0; j < MAXK; ++j) k1.k[j] = 0;
    k1.k[0] = 1;  // replaced on the device by cand.kdev[0]
    rc = run_select(candv, (int64_t)cap, k1, &ex->cand, st);
    if (rc != SALUN_OK) return rc;
    hipLaunchKernelGGL(k_sampled_fixup, dim3(1024), dim3(SALUN_BLOCK), 0, st, candv, candi, ex, cap, mp.m[0]);
    SALUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sampled_finish, dim3(1), dim3(1024), 0, st, ex, state, (long long)ks[0], mp.m[0]);
    SALUN_LAUNCH_CHECK();
  }

  {
    const int rc = run_select(acc, n, kl, state, st);  // returns at once on the device if the sampled path published
    if (rc != SALUN_OK) return rc;
  }
  // rare path, early-exits on the device when no threshold splits a run of ties
  if (aligned) hipLaunchKernelGGL(k_tie_count<true>, dim3(grid), dim3(SALUN_BLOCK), 0, st, acc, n, state, tie, nchunk);
  else hipLaunchKernelGGL(k_tie_count<false>, dim3(grid), dim3(SALUN_BLOCK), 0, st, acc, n, state, tie, nchunk);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_tie_scan, dim3(nk), dim3(SALUN_BLOCK), 0, st, state, tie, nchunk);
  SALUN_LAUNCH_CHECK();
#define SALUN_WRITE(A, M) \
  hipLaunchKernelGGL((k_write_masks<A, M>), dim3(grid), dim3(SALUN_BLOCK), 0, st, acc, n, state, tie, nchunk, mp)
  if (aligned && maligned) SALUN_WRITE(true, true);
  else if (aligned) SALUN_WRITE(true, false);
  else if (maligned) SALUN_WRITE(false, true);
  else SALUN_WRITE(false, false);
#undef SALUN_WRITE
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_mask_topk_thresholds(const void *ws, int nk, float *tau_out, salun_stream_t stream) {
  if (!ws || !tau_out || nk < 1 || nk > MAXK) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_export_tau, dim3(1), dim3(64), 0, salun_hip_stream(stream),
                     static_cast<const TopkState *>(ws), nk, tau_out);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_mask_u8_to_i64(const uint8_t *m, int64_t *out, int64_t n, salun_stream_t stream) {
  if (n < 0 || (n > 0 && (!m || !out))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  const int aligned = salun_aligned4(m) && salun_aligned16(out);
  hipLaunchKernelGGL(k_u8_to_i64, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), m, reinterpret_cast<long long *>(out), n, aligned);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_mask_i64_to_u8(const int64_t *m, uint8_t *out, int64_t n, salun_stream_t stream) {
  if (n < 0 || (n > 0 && (!m || !out))) return SALUN_EINVAL;
  if (n == 0) return SALUN_OK;
  const int aligned = salun_aligned16(m) && salun_aligned4(out);
  hipLaunchKernelGGL(k_i64_to_u8, dim3(salun_grid_for(n, SALUN_BLOCK * 4)), dim3(SALUN_BLOCK), 0,
                     salun_hip_stream(stream), reinterpret_cast<const long long *>(m), out, n, aligned);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_mask_popcount(const uint8_t *m, int64_t n, int64_t *count, void *ws, size_t ws_bytes,
                                     salun_stream_t stream) {
  if (n < 0 || !count || !ws || (n > 0 && !m)) return SALUN_EINVAL;
  if (ws_bytes < sizeof(u64) * 1024) return SALUN_ENOSPC;
  hipStream_t st = salun_hip_stream(stream);
  int grid = salun_grid_for(n, SALUN_BLOCK * 16);
  if (grid > 1024) grid = 1024;
  const int aligned = salun_aligned16(m);
  hipLaunchKernelGGL(k_popcount_partial, dim3(grid), dim3(SALUN_BLOCK), 0, st, m, n, static_cast<u64 *>(ws), aligned);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_sum_partials_i64, dim3(1), dim3(SALUN_BLOCK), 0, st, static_cast<const u64 *>(ws), grid,
                     reinterpret_cast<long long *>(count));
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
