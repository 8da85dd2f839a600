// salun_topk.hip — K2: global top-k saliency mask (gfx950 / CDNA4).
//
// The reference ranks every weight with two full argsorts per threshold
// (Classification/generate_mask.py:57-64) only to compare the rank with k.  Here the k-th largest |acc| is
// located by selection on a monotone 31-bit integer key, integer-only, hence bit-exact and independent of launch
// geometry:
//
//   key(x) = 0                           if x is NaN   (ranks after every number)
//          = (bits(x) & 0x7fffffff) + 1  otherwise     (|x| as an ordered integer; the abs is fused)
//   mask_j[i] = 1  iff  key_i > tau_j, or key_i == tau_j and i is among the first r_j indices holding tau_j
//               (stable tie rule)
//
// Two routes, the same function of the input:
//
// FAST (n >= 8192, 16-B aligned input; the route of every model-sized vector) — the vector is read ONCE:
//   k_bracket   one workgroup ranks a 16 K-element hashed sample with a two-level LDS histogram and brackets every
//               threshold,  lo_j <= tau_j <= hi_j,  6 sigma either side of the target rank (~4.7 % of the mass)
//   k_main      the one streaming pass: writes mask_j = [key > hi_j] (final outside the bracket), counts
//               c_gt_j = #{key > hi_j}, compacts the in-bracket candidates (key, flat index) into per-workgroup slabs
//   k_hist_a    histogram of the candidates over the bracket (LDS, 256..1024 bins per threshold)
//   k_resolve   every workgroup picks the bin holding rank k_j - c_gt_j; candidates above it get their mask byte,
//               candidates inside it go to a short list + a 4096-bin histogram of that bin
//   k_finish    one workgroup per threshold: picks the final bin (<= a few keys wide), ranks its residents
//               exactly (key descending, flat index ascending) and publishes tau_j
//   For n >= 2^27 the bracket comes from an exact selection (this same route, values only) on a 2^20-element sample
//   instead, which narrows it to ~0.6 % of the mass.
//   Anything unusual — a bracket that misses, a slab or list that overflows (heavy ties) — raises `fail` on the
//   device and the full scan below redoes the job; no host synchronisation anywhere.
//
// FULL SCAN (small or unaligned inputs, and the fallback): k_fullscan, ONE persistent launch of <= 2 workgroups per
//   CU: three histogram passes (11 + 10 + 10 bit digits, most significant first; thresholds that share a prefix share
//   a histogram), per-workgroup LDS histograms flushed with 64-bit global atomics, every workgroup repeating the tiny
//   selection step itself, grid barriers between passes (monotonic counter, agent-scope release/acquire, bounded
//   spin), per-chunk tie prefixes when a threshold splits a run of equal keys, one write pass.  When launched behind
//   the fast route it returns at once unless `fail` is set.
#include "salun_common.h"
#include <cmath>
#include <cstdlib>

namespace {

constexpr int MAXK = SALUN_MAX_THRESHOLDS;
constexpr int D0_BINS = 2048;  // key >> 20
constexpr int CHUNK_VEC = 4 * SALUN_BLOCK;  // float4 per chunk (4 sub-vectors per lane)
constexpr int CHUNK = CHUNK_VEC * 4;        // 4096 elements: streaming / tie-ordering granule

constexpr int64_t FAST_MIN_N = 8192;
constexpr int SAMPLE_MAX = 16384;             // k_bracket's sample (16 keys per thread)
constexpr int64_t TWO_LEVEL_MIN_N = int64_t(1) << 27;
constexpr int SAMPLE2_LOG2 = 20;              // outer sample of the two-level route
constexpr int HIST2_BINS = 4096;
constexpr int BINS_A_MAX = 1024;
constexpr int FINAL_CAP = 1024;               // final-bin residents ranked exactly in LDS
constexpr int MAIN_GRID = 1024;
constexpr double BRACKET_SIGMAS = 6.0;

enum Mode : uint32_t { MODE_NONE = 0, MODE_ALL = 1, MODE_GE = 2, MODE_ORDERED = 3 };

typedef unsigned long long u64;
typedef float vf4 __attribute__((ext_vector_type(4)));

// What salun_mask_topk_thresholds reads: first bytes of the workspace, written by whichever route finished the job.
struct TopkPub {
  uint32_t nk;
  uint32_t error;        // 1: a grid barrier of the full scan timed out (results invalid)
  uint32_t mode[MAXK];
  uint32_t tau[MAXK];    // key of the k-th element
  uint32_t route;        // 1 fast, 2 full scan (diagnostic)
  uint32_t pad;
};

struct FullState {
  u64 hist0[D0_BINS];
  u64 hist1[MAXK][1024];
  u64 hist2[MAXK][1024];
  uint32_t bar;          // grid-barrier counter
  uint32_t pad;
};

struct FastState {
  long long k[MAXK];     // clamped to [0, n]
  uint32_t mode[MAXK];   // MODE_NONE or MODE_GE (bracketed)
  uint32_t was_all[MAXK];  // k > n on entry: published as MODE_ALL
  uint32_t lo[MAXK], hi[MAXK], shiftA[MAXK];
  uint32_t mid[MAXK];    // k_main's guess of the threshold (centre of the bracket): it writes mask = [key > mid] and
                         // the later kernels touch only the candidates whose final bit differs from that guess
  u64 c_gt[MAXK];        // elements strictly above the bracket
  uint32_t n2[MAXK];     // residents of the chosen first-level bin
  uint32_t lo2[MAXK], hi2[MAXK], shift2[MAXK];
  u64 r2[MAXK];          // rank wanted inside [lo2, hi2]
  uint32_t fail;         // -> the full scan redoes the job
  uint32_t pad;
  uint32_t histA[MAXK][BINS_A_MAX];
  uint32_t hist2[MAXK][HIST2_BINS];
};

struct KList {
  long long k[MAXK];
  int nk;
};
struct MaskPtrs {
  uint8_t *m[MAXK];
};
// explicit target ranks (descending, 1-based) in a sample, for the two-level route
struct RankList {
  long long hi[MAXK], lo[MAXK];  // 0 = unbounded on that side
};

__device__ __forceinline__ uint32_t key_of(float x) {
  const uint32_t b = __float_as_uint(x) & 0x7FFFFFFFu;
  return (b > 0x7F800000u) ? 0u : b + 1u;
}
constexpr uint32_t KEY_MAX = 0x7F800001u;   // key of +-inf: no real key is larger
constexpr uint32_t KEY_SKIP = 0xFFFFFFFFu;  // out-of-range lane marker (never a real key)

__device__ __forceinline__ u64 ld_agent_u64(const u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent_u32(const uint32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 64-lane scans
__device__ __forceinline__ u64 wave_excl_scan_u64(u64 v, int lane) {
  u64 inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u64 t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  return inc - v;
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// =====================================================================================================
//                                              FAST ROUTE
// =====================================================================================================

// ---------------------------------------------------------------------------------- k_bracket
// One workgroup of 1024 threads.  Zeroes the fast state, ranks a hashed sample of S keys with a 2048-bin and a
// 256-bin LDS histogram (19 key bits: the bin edges, taken outward, only widen a bracket by ~0.05 % of the values)
// and writes lo/hi per threshold.
__device__ __forceinline__ uint32_t shift_for(uint32_t lo, uint32_t hi, int bins) {
  const uint32_t w = hi - lo;
  uint32_t s = 0;
  while ((w >> s) >= (uint32_t)bins) ++s;
  return s;
}

// Workgroup 0 of the sampling kernel clears the small head of the state and the publication block for this call
// (the histograms are zeroed by k_main's workgroups); k_bracket's workgroups then fill in one threshold each.
__device__ __forceinline__ void reset_head(FastState *fs, TopkPub *pub, FullState *full, int nk, int tid, int nthreads) {
  uint32_t *z = reinterpret_cast<uint32_t *>(fs);
  const int words_head = (int)(offsetof(FastState, histA) / 4);
  for (int i = tid; i < words_head; i += nthreads) z[i] = 0;
  if (tid == 0) { full->bar = 0; pub->nk = (uint32_t)nk; pub->error = 0; pub->route = 0; }
}

// The sample: one element per stride window at a hashed offset (a fixed offset would lock onto periodic structure of
// the flat vector, e.g. always the centre tap of 3x3 kernels).  Spread over S/1024 workgroups: 16 K random cache lines
// are more than one CU can pull in a few microseconds.  Writes the KEYS, coalesced.
__global__ __launch_bounds__(1024) void k_sample(const float *__restrict__ acc, int64_t n, int S,
                                                 uint32_t *__restrict__ keys, FastState *fs, TopkPub *pub,
                                                 FullState *full, int nk) {
  if (blockIdx.x == 0) reset_head(fs, pub, full, nk, threadIdx.x, 1024);
  const int64_t stride = n / S;  // < 2^31: the offset inside a window is a 32-bit multiply-high, not a 64-bit modulo
  const int64_t s = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  if (s < S) {
    const uint32_t off = __umulhi((uint32_t)salun_splitmix64((uint64_t)s), (uint32_t)stride);
    keys[s] = key_of(acc[s * stride + (int64_t)off]);
  }
}

constexpr int H0_COPIES = 8;  // lane-indexed copies of the first-level histogram: its hot bins would serialise the
                              // LDS atomics of a wave (a model's |gradients| sit in ~40 of the 2048 bins)

__global__ __launch_bounds__(1024) void k_bracket(const uint32_t *__restrict__ keys, int64_t n, KList kl, int S, int bins_a,
                                                  FastState *fs, TopkPub *pub, FullState *full) {
  __shared__ uint32_t h0c[H0_COPIES][D0_BINS];
  __shared__ uint32_t h1[2 * MAXK][256];
  __shared__ uint8_t lut[D0_BINS];
  __shared__ long long s_rank[2 * MAXK];   // descending rank in the sample; 0 = unbounded
  __shared__ uint32_t s_b0[2 * MAXK], s_rem[2 * MAXK], s_grp[2 * MAXK], s_b1[2 * MAXK];
  __shared__ uint32_t s_gprefix[2 * MAXK];
  uint32_t *h0 = h0c[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // One workgroup per threshold (every workgroup histograms the whole sample: 64 KB out of L2); a one-workgroup
  // launch serves all of them.  The head of the state was cleared by the sampling kernel.
  const int nk = gridDim.x > 1 ? 1 : kl.nk;
  const int j0 = gridDim.x > 1 ? (int)blockIdx.x : 0;
  for (int i = tid; i < H0_COPIES * D0_BINS; i += 1024) (&h0c[0][0])[i] = 0;
  for (int i = tid; i < D0_BINS; i += 1024) lut[i] = 0;
  for (int i = tid; i < 2 * nk * 256; i += 1024) (&h1[0][0])[i] = 0;
  // ---- target ranks
  if (tid < nk) {
    long long k = kl.k[j0 + tid];
    if (k > n) k = n;
    long long rhi = 0, rlo = 0;
    if (k > 0) {
      const double p = (double)k / (double)n;
      const double sigma = sqrt((double)S * p * (1.0 - p));
      const long long margin = (long long)ceil(BRACKET_SIGMAS * sigma) + 8;
      const long long rho = llround(p * (double)S);
      rhi = rho - margin;  // larger keys: smaller descending rank
      rlo = rho + margin;
      if (rhi < 1) rhi = 0;
      if (rlo > S) rlo = 0;
    }
    s_rank[2 * tid] = rhi;
    s_rank[2 * tid + 1] = rlo;
  }
  const int per = S >> 10;  // 4 .. 16 keys per thread
  uint32_t key[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) key[i] = (i < per) ? keys[i * 1024 + tid] : KEY_SKIP;
  __syncthreads();
  {
    uint32_t *mine = h0c[lane & (H0_COPIES - 1)];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key[i] != KEY_SKIP) atomicAdd(&mine[key[i] >> 20], 1u);
  }
  __syncthreads();
  for (int i = tid; i < D0_BINS; i += 1024) {
    uint32_t t = 0;
#pragma unroll
    for (int c = 0; c < H0_COPIES; ++c) t += h0c[c][i];
    h0[i] = t;  // copy 0 becomes the total (each thread touches only its own bins)
  }
  __syncthreads();
  // ---- level 0: wave w serves ranks w, w + 16 (bins walked from the top, lane l owns 32 bins)
  for (int q = wave; q < 2 * nk; q += 16) {
    const long long want = s_rank[q];
    if (want == 0) continue;
    uint32_t mine = 0;
    for (int j = 0; j < 32; ++j) mine += h0[D0_BINS - 1 - (lane * 32 + j)];
    const uint32_t incl = wave_incl_scan_u32(mine, lane);
    const uint32_t before = incl - mine;
    if ((long long)before < want && want <= (long long)incl) {
      uint32_t cum = before;
      for (int j = 0; j < 32; ++j) {
        const int bin = D0_BINS - 1 - (lane * 32 + j);
        const uint32_t c = h0[bin];
        if (want <= (long long)(cum + c)) { s_b0[q] = (uint32_t)bin; s_rem[q] = (uint32_t)(want - cum); break; }
        cum += c;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t ng = 0;
    for (int q = 0; q < 2 * nk; ++q) {
      if (s_rank[q] == 0) continue;
      uint32_t g = 0;
      for (; g < ng; ++g) if (s_gprefix[g] == s_b0[q]) break;
      if (g == ng) { s_gprefix[ng] = s_b0[q]; lut[s_b0[q]] = (uint8_t)(ng + 1); ++ng; }
      s_grp[q] = g;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (key[i] != KEY_SKIP) {
      const uint32_t g = lut[key[i] >> 20];
      if (g) atomicAdd(&h1[g - 1][(key[i] >> 12) & 255u], 1u);
    }
  __syncthreads();
  // ---- level 1: 256 bins, lane owns 4
  for (int q = wave; q < 2 * nk; q += 16) {
    if (s_rank[q] == 0) continue;
    const uint32_t *h = h1[s_grp[q]];
    const uint32_t want = s_rem[q];
    uint32_t mine = 0;
    for (int j = 0; j < 4; ++j) mine += h[255 - (lane * 4 + j)];
    const uint32_t incl = wave_incl_scan_u32(mine, lane);
    const uint32_t before = incl - mine;
    if (before < want && want <= incl) {
      uint32_t cum = before;
      for (int j = 0; j < 4; ++j) {
        const int bin = 255 - (lane * 4 + j);
        const uint32_t c = h[bin];
        if (want <= cum + c) { s_b1[q] = (uint32_t)bin; break; }
        cum += c;
      }
    }
  }
  __syncthreads();
  if (tid < nk) {
    const int j = j0 + tid;
    long long k = kl.k[j];
    if (k > n) k = n;
    uint32_t lo, hi, mode;
    if (k <= 0) {
      k = 0; mode = MODE_NONE; lo = hi = KEY_SKIP;  // nothing is above, nothing is inside
    } else {
      mode = MODE_GE;
      hi = s_rank[2 * tid] ? ((s_b0[2 * tid] << 20) | (s_b1[2 * tid] << 12) | 0xFFFu) : KEY_MAX;
      lo = s_rank[2 * tid + 1] ? ((s_b0[2 * tid + 1] << 20) | (s_b1[2 * tid + 1] << 12)) : 0u;
      if (hi > KEY_MAX) hi = KEY_MAX;
    }
    fs->k[j] = k;
    fs->mode[j] = mode;
    fs->was_all[j] = kl.k[j] > n ? 1u : 0u;
    fs->lo[j] = lo;
    fs->hi[j] = hi;
    fs->mid[j] = (mode == MODE_GE) ? lo + (hi - lo) / 2 : KEY_SKIP;
    fs->shiftA[j] = (mode == MODE_GE) ? shift_for(lo, hi, bins_a) : 0;
  }
}

// Two-level route (n >= 2^27).  The 2^20-element sample is bracketed by its own 16 K sub-sample (k_bracket), streamed
// once by k_main (values only) and histogrammed by k_hist_a; this kernel then reads, for the two target ranks of
// every threshold (2j = upper, 2j+1 = lower), the histogram bin holding that rank and takes the bin's OUTER edge as
// the bracket of the full vector: 1024 bins over ~4.7 % of the sample's mass widen a bracket by < 0.01 % of the mass,
// against its own width of ~0.6 %.  A rank outside its sample bracket raises `fail` (the full scan takes over).
__global__ __launch_bounds__(1024) void k_bracket_from_hist(const FastState *inner, int inner_bins, int64_t n, KList kl,
                                                            RankList rl, int bins_a, FastState *fs, TopkPub *pub,
                                                            FullState *full) {
  __shared__ uint32_t s_edge[2 * MAXK];
  __shared__ uint32_t s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = kl.nk;
  uint32_t *z = reinterpret_cast<uint32_t *>(fs);
  const int words_head = (int)(offsetof(FastState, histA) / 4);
  for (int i = tid; i < words_head; i += 1024) z[i] = 0;
  if (tid == 0) { full->bar = 0; pub->nk = (uint32_t)nk; pub->error = 0; pub->route = 0; s_bad = inner->fail; }
  __syncthreads();
  for (int q = wave; q < 2 * nk; q += 16) {
    const bool upper = (q & 1) == 0;
    const long long want_rank = upper ? rl.hi[q >> 1] : rl.lo[q >> 1];
    if (want_rank == 0 || inner->mode[q] != MODE_GE) continue;  // unbounded side / trivial threshold
    const long long r = inner->k[q] - (long long)inner->c_gt[q];
    const uint32_t *h = inner->histA[q];
    const int per = inner_bins / 64;
    u64 mine = 0;
    for (int i = 0; i < per; ++i) mine += h[inner_bins - 1 - (lane * per + i)];
    const u64 before = wave_excl_scan_u64(mine, lane);
    const u64 total = __shfl(before + mine, 63, 64);
    if (r < 1 || (u64)r > total) { if (lane == 0) s_bad = 1; continue; }
    if (before < (u64)r && (u64)r <= before + mine) {
      u64 cum = before;
      for (int i = 0; i < per; ++i) {
        const int bin = inner_bins - 1 - (lane * per + i);
        const u64 c = h[bin];
        if ((u64)r <= cum + c) {
          const uint32_t sh = inner->shiftA[q];
          const uint32_t lo_edge = inner->lo[q] + ((uint32_t)bin << sh);
          uint32_t hi_edge = lo_edge + ((1u << sh) - 1u);
          if (hi_edge > inner->hi[q] || hi_edge < lo_edge) hi_edge = inner->hi[q];
          s_edge[q] = upper ? hi_edge : lo_edge;
          break;
        }
        cum += c;
      }
    }
  }
  __syncthreads();
  if (tid < nk) {
    long long k = kl.k[tid];
    if (k > n) k = n;
    uint32_t lo, hi, mode;
    if (k <= 0) {
      k = 0; mode = MODE_NONE; lo = hi = KEY_SKIP;
    } else {
      mode = MODE_GE;
      hi = rl.hi[tid] ? s_edge[2 * tid] : KEY_MAX;
      lo = rl.lo[tid] ? s_edge[2 * tid + 1] : 0u;
      if (hi > KEY_MAX) hi = KEY_MAX;
      if (lo > hi) lo = hi;  // cannot happen for valid sample brackets; keeps the arithmetic below in range
    }
    fs->k[tid] = k;
    fs->mode[tid] = mode;
    fs->was_all[tid] = kl.k[tid] > n ? 1u : 0u;
    fs->lo[tid] = lo;
    fs->hi[tid] = hi;
    fs->mid[tid] = (mode == MODE_GE) ? lo + (hi - lo) / 2 : KEY_SKIP;
    fs->shiftA[tid] = (mode == MODE_GE) ? shift_for(lo, hi, bins_a) : 0;
  }
  __syncthreads();
  if (tid == 0 && s_bad) fs->fail = 1;
}

// The 2^20-element sample of the two-level route (values, for k_main) and, in the same launch, its 16 K-element
// sub-sample (keys, for k_bracket): window w of `sub` consecutive samples contributes the one at a hashed offset.
__global__ __launch_bounds__(SALUN_BLOCK) void k_gather_sample(const float *__restrict__ acc, int64_t stride, int64_t S,
                                                               float *__restrict__ out, int sub,
                                                               uint32_t *__restrict__ keys, FastState *fs, TopkPub *pub,
                                                               FullState *full, int nk) {
  if (blockIdx.x == 0) reset_head(fs, pub, full, nk, threadIdx.x, SALUN_BLOCK);
  const int64_t s = (int64_t)blockIdx.x * SALUN_BLOCK + threadIdx.x;
  if (s >= S) return;
  const float v = acc[s * stride + (int64_t)__umulhi((uint32_t)salun_splitmix64((uint64_t)s ^ 0x5bd1e995ull),
                                                     (uint32_t)stride)];
  out[s] = v;
  const int64_t w = s / sub;
  if ((int64_t)__umulhi((uint32_t)salun_splitmix64((uint64_t)w), (uint32_t)sub) == s - w * sub) keys[w] = key_of(v);
}

// -------------------------------------------------------------------------------------- k_main
// The one streaming pass.  Chunk c = 4096 elements; lane t, sub-vector u touches float4 #(c*1024 + u*256 + t), so a
// wave instruction covers a contiguous 1 KiB and the mask goes out as one dword per float4.  Candidates are
// compacted slot by slot: the ballot of "lane holds a candidate in slot (u, e)" gives every lane its offset
// (mbcnt) with no scan, one LDS atomic per wave and chunk reserves the slab range, empty slots are skipped by a
// scalar branch.  No same-address global atomics anywhere (each costs ~12 ns at the memory side and they serialise):
// per-workgroup counts go to plain rows that the next kernel sums.
template <int NK, bool VO>
__global__ __launch_bounds__(SALUN_BLOCK) void k_main(const float *__restrict__ acc, int64_t n, FastState *fs,
                                                      MaskPtrs mp, uint2 *__restrict__ slabs,
                                                      uint32_t *__restrict__ slab_cnt, uint32_t *__restrict__ wg_gt,
                                                      uint32_t cap, int nk_real, int bins_a) {
  __shared__ uint32_t s_cnt[NK];
  __shared__ uint32_t s_gt[4][NK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NK) s_cnt[tid] = 0;
  {  // this workgroup's share of zeroing the histograms k_hist_a / k_resolve accumulate into
    const int wa = nk_real * BINS_A_MAX, w2 = nk_real * HIST2_BINS;
    uint32_t *za = &fs->histA[0][0], *z2 = &fs->hist2[0][0];
    for (int i = blockIdx.x * SALUN_BLOCK + tid; i < wa + w2; i += gridDim.x * SALUN_BLOCK) {
      if (i < wa) za[i] = 0; else z2[i - wa] = 0;
    }
  }
  uint32_t hi[NK], lo[NK], mid[NK], gtc[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    hi[j] = (j < nk_real) ? fs->hi[j] : KEY_SKIP;
    lo[j] = (j < nk_real) ? fs->lo[j] : KEY_SKIP;
    mid[j] = (j < nk_real) ? fs->mid[j] : KEY_SKIP;
    gtc[j] = 0;  // wave-uniform count of keys above the bracket
  }
  __syncthreads();
  const int64_t nfull = n / CHUNK;
  for (int64_t c = blockIdx.x; c < nfull; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // read exactly once: non-temporal
      const vf4 x = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(acc) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
      k[u][0] = key_of(x.x); k[u][1] = key_of(x.y); k[u][2] = key_of(x.z); k[u][3] = key_of(x.w);
    }
    const uint32_t idx0 = (uint32_t)((c * CHUNK_VEC + tid) << 2);  // flat index of slot (u, e): idx0 + u*1024 + e
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      if (j >= nk_real) continue;  // NK is the instantiated size >= nk_real (uniform)
      const uint32_t w = hi[j] - lo[j];
      unsigned long long bal[16];
      uint32_t total = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bits |= (uint32_t)(k[u][e] > mid[j]) << (8 * e);
          bal[u * 4 + e] = __builtin_amdgcn_ballot_w64((k[u][e] - lo[j]) <= w);
          total += (uint32_t)__builtin_popcountll(bal[u * 4 + e]);
          gtc[j] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(k[u][e] > hi[j]));
        }
        if (!VO)
          __builtin_nontemporal_store(bits, reinterpret_cast<uint32_t *>(mp.m[j]) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
      }
      if (total) {  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt[j], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        uint2 *slab = slabs + ((size_t)blockIdx.x * (size_t)nk_real + (size_t)j) * cap;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned long long b = bal[u * 4 + e];
            if (b) {  // scalar branch: most slots hold no candidate when the bracket is narrow
              if ((b >> lane) & 1ull) {
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                if (pos < cap) slab[pos] = make_uint2(k[u][e], idx0 + (uint32_t)(u * 1024 + e));
              }
              base += (uint32_t)__builtin_popcountll(b);
            }
          }
      }
    }
  }
  // ragged tail (n % 4096 elements): one workgroup, element-wise
  uint32_t tail_gt[NK];
#pragma unroll
  for (int j = 0; j < NK; ++j) tail_gt[j] = 0;
  if ((int64_t)blockIdx.x == nfull % (int64_t)gridDim.x) {
    for (int64_t i = nfull * CHUNK + tid; i < n; i += SALUN_BLOCK) {
      const uint32_t key = key_of(acc[i]);
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (j >= nk_real) continue;
        tail_gt[j] += key > hi[j];
        if (!VO) mp.m[j][i] = (uint8_t)(key > mid[j]);
        if ((key - lo[j]) <= (hi[j] - lo[j])) {
          const uint32_t pos = atomicAdd(&s_cnt[j], 1u);
          if (pos < cap)
            slabs[((size_t)blockIdx.x * (size_t)nk_real + (size_t)j) * cap + pos] = make_uint2(key, (uint32_t)i);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    uint32_t v = tail_gt[j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) s_gt[wave][j] = v + gtc[j];
  }
  __syncthreads();
  if (tid < nk_real) {
    wg_gt[(size_t)blockIdx.x * nk_real + tid] = s_gt[0][tid] + s_gt[1][tid] + s_gt[2][tid] + s_gt[3][tid];
    const uint32_t mine = s_cnt[tid];
    slab_cnt[(size_t)blockIdx.x * nk_real + tid] = mine < cap ? mine : cap;
    if (mine > cap) fs->fail = 1;
  }
}

// ------------------------------------------------------------------------------------ k_hist_a
// Histogram of the compacted candidates over their bracket, one wave per slab row (no serial chain of dependent
// loads inside a workgroup).  Dynamic LDS: nk * bins_a counters.  Workgroups 0..nk-1 also sum the per-workgroup
// "above the bracket" counts of k_main into c_gt.
__global__ __launch_bounds__(1024) void k_hist_a(FastState *fs, const uint2 *__restrict__ slabs,
                                                 const uint32_t *__restrict__ slab_cnt,
                                                 const uint32_t *__restrict__ wg_gt, uint32_t cap, int main_grid, int nk,
                                                 int bins_a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ u64 s_red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (fs->fail) return;  // a slab overflowed in k_main (heavy ties): the full scan redoes the job
  for (int i = tid; i < nk * bins_a; i += 1024) lds[i] = 0;
  __syncthreads();
  const int rows = main_grid * nk;
  const int waves = gridDim.x * 16;
  const int spr = waves > rows ? waves / rows : 1;  // waves sharing one row when there are more waves than rows
  for (int w = blockIdx.x * 16 + wave; w < rows * spr; w += waves) {
    const int r = w / spr, part = w % spr;
    const int j = r % nk;
    const uint32_t cnt = slab_cnt[r];
    const uint32_t lo = fs->lo[j], sh = fs->shiftA[j];
    const uint2 *slab = slabs + (size_t)r * cap;
    uint32_t *h = lds + j * bins_a;
    for (uint32_t i = part * 64 + lane; i < cnt; i += 64 * spr) atomicAdd(&h[(slab[i].x - lo) >> sh], 1u);
  }
  __syncthreads();
  for (int i = tid; i < nk * bins_a; i += 1024)
    if (lds[i]) atomicAdd(&fs->histA[i / bins_a][i % bins_a], lds[i]);
  if ((int)blockIdx.x < nk) {
    const int j = blockIdx.x;
    u64 v = 0;
    for (int b = tid; b < main_grid; b += 1024) v += wg_gt[(size_t)b * nk + j];
    v = salun_wave_sum_u64(v);
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    if (tid == 0) {
      u64 t = 0;
      for (int w = 0; w < 16; ++w) t += s_red[w];
      fs->c_gt[j] = t;
    }
  }
}

// ------------------------------------------------------------------------------------ k_resolve
// Every workgroup repeats the tiny first-level selection (wave w serves threshold w), then its waves walk slab rows
// (one wave per row): candidates above the chosen bin are selected for good, candidates inside it are counted in a
// 4096-bin histogram of that bin (global atomics spread over 4096 addresses) and staged in LDS; each workgroup owns a
// private segment of the threshold's short list and a plain count, so nothing is appended through a shared counter.
constexpr int STAGE_CAP = 1024;

template <bool VO>
__global__ __launch_bounds__(1024) void k_resolve(FastState *fs, const uint2 *__restrict__ slabs,
                                                  const uint32_t *__restrict__ slab_cnt, uint32_t cap, int main_grid,
                                                  int nk, int bins_a, int nseg, uint2 *__restrict__ list2 /*[nk][nseg][STAGE_CAP]*/,
                                                  uint32_t *__restrict__ seg_cnt /*[nk][nseg]*/, MaskPtrs mp) {
  __shared__ uint32_t s_lo2, s_hi2, s_sh2, s_ok;
  __shared__ uint2 s_stage[STAGE_CAP];
  __shared__ uint32_t s_h2[HIST2_BINS];  // this workgroup's share of the 4096-bin histogram: a run of equal keys
                                         // would otherwise serialise ~12 ns global atomics on ONE address
  __shared__ uint32_t s_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x % nk, g = blockIdx.x / nk;  // workgroup g of the nseg that serve threshold j
  if (fs->fail) return;  // already decided (slab overflow): the full scan redoes the job
  if (tid == 0) { s_n = 0; s_ok = 0; }
  for (int i = tid; i < HIST2_BINS; i += 1024) s_h2[i] = 0;
  __syncthreads();
  if (wave == 0 && fs->mode[j] == MODE_GE) {
    const long long r = fs->k[j] - (long long)fs->c_gt[j];  // rank wanted among the candidates, 1-based descending
    const uint32_t *h = fs->histA[j];
    const int per = bins_a / 64;
    u64 mine = 0;
    for (int i = 0; i < per; ++i) mine += h[bins_a - 1 - (lane * per + i)];
    const u64 before = wave_excl_scan_u64(mine, lane);
    const u64 total = __shfl(before + mine, 63, 64);
    if (r >= 1 && (u64)r <= total) {
      if (before < (u64)r && (u64)r <= before + mine) {
        u64 cum = before;
        for (int i = 0; i < per; ++i) {
          const int bin = bins_a - 1 - (lane * per + i);
          const u64 c = h[bin];
          if ((u64)r <= cum + c) {
            const uint32_t sh = fs->shiftA[j];
            const uint32_t lo2 = fs->lo[j] + ((uint32_t)bin << sh);
            uint32_t hi2 = lo2 + ((1u << sh) - 1u);
            if (hi2 > fs->hi[j] || hi2 < lo2) hi2 = fs->hi[j];
            s_lo2 = lo2; s_hi2 = hi2; s_sh2 = sh > 12 ? sh - 12 : 0; s_ok = 1;
            if (g == 0) {
              fs->lo2[j] = lo2; fs->hi2[j] = hi2; fs->shift2[j] = sh > 12 ? sh - 12 : 0;
              fs->r2[j] = (u64)r - cum;
            }
            break;
          }
          cum += c;
        }
      }
    } else if (g == 0 && lane == 0) {
      fs->fail = 1;  // the bracket missed the threshold
    }
  }
  __syncthreads();
  if (s_ok) {
    const uint32_t lo2 = s_lo2, hi2 = s_hi2, sh2 = s_sh2, mid = fs->mid[j];
    (void)mid;
    uint8_t *mask = mp.m[j];
    const int waves = nseg * 16;
    const int spr = waves > main_grid ? waves / main_grid : 1;  // waves sharing one slab row
    for (int w = g * 16 + wave; w < main_grid * spr; w += waves) {
      const int b = w / spr, part = w % spr;
      const size_t r = (size_t)b * nk + j;
      const uint32_t cnt = slab_cnt[r];
      const uint2 *slab = slabs + r * cap;
      for (uint32_t i = part * 64 + lane; i < cnt; i += 64 * spr) {
        const uint2 e = slab[i];
        if (e.x > hi2) { if (!VO && e.x <= mid) mask[e.y] = 1; }       // selected for good; k_main guessed 0
        else if (e.x < lo2) { if (!VO && e.x > mid) mask[e.y] = 0; }   // rejected for good; k_main guessed 1
        else {
          atomicAdd(&s_h2[(e.x - lo2) >> sh2], 1u);
          const uint32_t p = atomicAdd(&s_n, 1u);
          if (p < (uint32_t)STAGE_CAP) s_stage[p] = e;
        }
      }
    }
  }
  __syncthreads();
  if (s_n)
    for (int i = tid; i < HIST2_BINS; i += 1024)
      if (s_h2[i]) atomicAdd(&fs->hist2[j][i], s_h2[i]);
  const uint32_t m = s_n;
  const uint32_t keep = m < (uint32_t)STAGE_CAP ? m : (uint32_t)STAGE_CAP;
  uint2 *seg = list2 + ((size_t)j * nseg + g) * STAGE_CAP;
  for (uint32_t q = tid; q < keep; q += 1024) seg[q] = s_stage[q];
  if (tid == 0) {
    seg_cnt[(size_t)j * nseg + g] = keep;
    if (m > (uint32_t)STAGE_CAP) fs->fail = 1;  // more residents than this workgroup's segment holds (heavy ties)
  }
}

// ------------------------------------------------------------------------------------- k_finish
// One workgroup (1024 threads) per threshold.
__device__ __forceinline__ uint32_t block1024_excl_scan(uint32_t v, uint32_t *s_w /*16*/, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan_u32(v, lane);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t woff = 0, tot = 0;
  for (int w = 0; w < 16; ++w) { if (w < wave) woff += s_w[w]; tot += s_w[w]; }
  *total = tot;
  __syncthreads();
  return woff + incl - v;
}

constexpr int MAX_SEGS = 1024;  // k_resolve grid bound

// entry #f of the concatenated segments: segment by binary search over the exclusive offsets
__device__ __forceinline__ uint2 list_entry(const uint2 *__restrict__ lst, const uint32_t *s_off, int nseg, uint32_t f) {
  int a = 0, b = nseg - 1;
  while (a < b) {
    const int mid = (a + b + 1) >> 1;
    if (s_off[mid] <= f) a = mid; else b = mid - 1;
  }
  return lst[(size_t)a * STAGE_CAP + (f - s_off[a])];
}

template <bool VO>
__global__ __launch_bounds__(1024) void k_finish(FastState *fs, TopkPub *pub, const uint2 *__restrict__ list2,
                                                 const uint32_t *__restrict__ seg_cnt, int nseg, MaskPtrs mp) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_off[MAX_SEGS];
  __shared__ uint32_t s_key[FINAL_CAP], s_idx[FINAL_CAP];
  __shared__ uint32_t s_b2, s_r3, s_c3, s_found, s_n, s_tau, s_hist[2048], s_sel, s_rem;
  const int j = blockIdx.x, tid = threadIdx.x;
  if (fs->fail) return;  // the full scan publishes
  const uint32_t mode = fs->mode[j];
  if (mode != MODE_GE) {
    if (tid == 0) { pub->mode[j] = MODE_NONE; pub->tau[j] = 0; pub->route = 1; }
    return;
  }
  const u64 r2 = fs->r2[j];
  const uint32_t lo2 = fs->lo2[j], hi2 = fs->hi2[j], sh2 = fs->shift2[j];
  if (tid == 0) { s_found = 0; s_n = 0; s_tau = 0; }
  uint32_t n2;
  {  // offsets of the per-workgroup segments of the short list
    const uint32_t c = (tid < nseg) ? seg_cnt[(size_t)j * nseg + tid] : 0;
    const uint32_t ex = block1024_excl_scan(c, s_w, &n2);
    if (tid < nseg) s_off[tid] = ex;
  }
  __syncthreads();
  // ---- final bin: 4096 bins walked from the top, thread t owns bins 4095-4t .. 4095-4t-3
  {
    const uint32_t *h = fs->hist2[j];
    uint32_t c[4], mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = h[HIST2_BINS - 1 - (tid * 4 + i)]; mine += c[i]; }
    uint32_t total;
    const uint32_t before = block1024_excl_scan(mine, s_w, &total);
    if ((u64)total != (u64)n2 || r2 < 1 || r2 > (u64)total) {
      if (tid == 0) fs->fail = 1;
      return;
    }
    if ((u64)before < r2 && r2 <= (u64)before + mine) {
      uint32_t cum = before;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!s_found && r2 <= (u64)cum + c[i]) {
          s_b2 = HIST2_BINS - 1 - (tid * 4 + i);
          s_r3 = (uint32_t)(r2 - cum);
          s_c3 = c[i];
          s_found = 1;
        }
        cum += c[i];
      }
    }
  }
  __syncthreads();
  const uint32_t lo3 = lo2 + (s_b2 << sh2);
  uint32_t hi3 = lo3 + ((1u << sh2) - 1u);
  if (hi3 > hi2 || hi3 < lo3) hi3 = hi2;
  const uint32_t r3 = s_r3, c3 = s_c3;
  const uint2 *lst = list2 + (size_t)j * nseg * STAGE_CAP;
  uint8_t *mask = mp.m[j];
  const uint32_t mid = fs->mid[j];  // k_main wrote [key > mid]
  const uint32_t all_mode = fs->was_all[j] ? MODE_ALL : MODE_GE;
  if (c3 <= (uint32_t)FINAL_CAP) {
    for (uint32_t f = tid; f < n2; f += 1024) {
      const uint2 e = list_entry(lst, s_off, nseg, f);
      if (e.x > hi3) { if (!VO && e.x <= mid) mask[e.y] = 1; }
      else if (e.x < lo3) { if (!VO && e.x > mid) mask[e.y] = 0; }
      else {
        const uint32_t p = atomicAdd(&s_n, 1u);
        if (p < (uint32_t)FINAL_CAP) { s_key[p] = e.x; s_idx[p] = e.y; }
      }
    }
    __syncthreads();
    const uint32_t L = s_n;
    if (L != c3) { if (tid == 0) fs->fail = 1; return; }
    for (uint32_t p = tid; p < L; p += 1024) {
      const uint32_t kp = s_key[p], ip = s_idx[p];
      uint32_t rank = 0;
      for (uint32_t q = 0; q < L; ++q) rank += (s_key[q] > kp) || (s_key[q] == kp && s_idx[q] < ip);
      if (!VO && (rank < r3) != (kp > mid)) mask[ip] = (uint8_t)(rank < r3);
      if (rank == r3 - 1) s_tau = kp;
    }
    __syncthreads();
    if (tid == 0) { pub->mode[j] = all_mode; pub->tau[j] = s_tau; pub->route = 1; }
    return;
  }
  if (sh2 != 0) {  // a crowded final bin that still spans several keys: not worth a fourth level here
    if (tid == 0) fs->fail = 1;
    return;
  }
  // ---- more than FINAL_CAP entries tie at the single key lo3: admit the r3 lowest flat indices.  Radix select
  // on the index (11 + 11 + 10 bits, most significant first, ascending) over the short list.
  uint32_t prefix = 0, want = r3;
  for (int level = 0; level < 3; ++level) {
    const int shift = (level == 0) ? 21 : (level == 1) ? 10 : 0;
    const int nb = (level == 2) ? 1024 : 2048;
    for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t f = tid; f < n2; f += 1024) {
      const uint2 e = list_entry(lst, s_off, nseg, f);
      if (e.x != lo3) continue;
      const bool match = (level == 0) || (level == 1 ? (e.y >> 21) == prefix : (e.y >> 10) == prefix);
      if (match) atomicAdd(&s_hist[(e.y >> shift) & (uint32_t)(nb - 1)], 1u);
    }
    __syncthreads();
    const uint32_t a = s_hist[2 * tid], b = s_hist[2 * tid + 1];  // ascending bins (bins >= nb stay zero)
    uint32_t total;
    const uint32_t before = block1024_excl_scan(a + b, s_w, &total);
    if (before < want && want <= before + a + b) {
      if (want <= before + a) { s_sel = 2 * tid; s_rem = want - before; }
      else { s_sel = 2 * tid + 1; s_rem = want - before - a; }
    }
    __syncthreads();
    prefix = (level == 0) ? s_sel : (level == 1) ? ((prefix << 11) | s_sel) : ((prefix << 10) | s_sel);
    want = s_rem;
    __syncthreads();
  }
  const uint32_t idx_star = prefix;  // the r3-th lowest index among the ties
  if (!VO)
    for (uint32_t f = tid; f < n2; f += 1024) {
      const uint2 e = list_entry(lst, s_off, nseg, f);
      const bool fin = e.x > hi3 || (e.x == lo3 && e.y <= idx_star);
      if (fin != (e.x > mid)) mask[e.y] = (uint8_t)fin;
    }
  if (tid == 0) { pub->mode[j] = all_mode; pub->tau[j] = lo3; pub->route = 1; }
}

// =====================================================================================================
//                                              FULL SCAN
// =====================================================================================================
struct Sel {  // selection state, one copy per workgroup (every workgroup repeats the selection)
  long long k[MAXK];
  u64 rem[MAXK];
  u64 ceq[MAXK];
  uint32_t prefix0[MAXK], prefix1[MAXK], tau[MAXK], mode[MAXK];
  uint32_t group0_of[MAXK], group1_of[MAXK], group0_prefix[MAXK], group1_prefix[MAXK];
  uint32_t ngroups0, ngroups1, any_ordered, nk;
  uint32_t bin[MAXK];
  u64 brem[MAXK], bcnt[MAXK];
};

__device__ __forceinline__ void load_keys(const float *__restrict__ acc, int64_t v, int64_t n, bool aligned,
                                          uint32_t k[4]) {
  const int64_t i = v << 2;
  if (aligned && i + 3 < n) {
    const float4 x = reinterpret_cast<const float4 *>(acc)[v];
    k[0] = key_of(x.x); k[1] = key_of(x.y); k[2] = key_of(x.z); k[3] = key_of(x.w);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = (i + e < n) ? key_of(acc[i + e]) : KEY_SKIP;
  }
}
__device__ __forceinline__ void load_chunk(const float *__restrict__ acc, int64_t c, int64_t nvec, int64_t n,
                                           bool aligned, uint32_t k[4][4]) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + threadIdx.x;
    if (v < nvec) load_keys(acc, v, n, aligned, k[u]);
    else k[u][0] = k[u][1] = k[u][2] = k[u][3] = KEY_SKIP;
  }
}

// Grid barrier: monotonic counter, lane-0 agent release before arriving, relaxed polling with s_sleep, agent
// acquire after (MI355X_MICROARCH.md "barrier-counter").  The spin is bounded: on a time-out `*err` is raised and
// the caller bails out.
__device__ __forceinline__ bool grid_barrier(uint32_t *counter, uint32_t *epoch, uint32_t *err) {
  __shared__ uint32_t s_bad;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *epoch += 1;
    const uint32_t target = *epoch * gridDim.x;
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0, bad = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 21) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
    }
    if (bad) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_bad = bad;
  }
  __syncthreads();
  return s_bad == 0;
}

// One level of the selection, repeated by every workgroup: wave w serves thresholds w, w+4, ...; bins are walked from
// the top, lane l owns the l-th highest slice; a 64-lane exclusive scan finds the slice holding the wanted element.
template <int LEVEL>
__device__ __forceinline__ void select_level(const FullState *full, Sel *S, int64_t n, const KList &kl) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int NBINS = (LEVEL == 0) ? D0_BINS : 1024;
  constexpr int PER_LANE = NBINS / 64;
  if (LEVEL == 0 && threadIdx.x < MAXK) {
    uint32_t mode = MODE_NONE;
    long long k = 0;
    if ((int)threadIdx.x < kl.nk) {
      k = kl.k[threadIdx.x];
      if (k <= 0) { k = 0; mode = MODE_NONE; }
      else if (k > n) { k = n; mode = MODE_ALL; }  // k == n runs the select: its threshold (the minimum) is real
      else mode = MODE_GE;  // provisional: refined after the last level
    }
    S->k[threadIdx.x] = k;
    S->mode[threadIdx.x] = mode;
    S->rem[threadIdx.x] = (u64)k;
    if (threadIdx.x == 0) S->nk = (uint32_t)kl.nk;
  }
  __syncthreads();
  for (int j = wave; j < kl.nk; j += 4) {
    if (S->mode[j] < MODE_GE) continue;
    const u64 *hist = (LEVEL == 0) ? full->hist0 : (LEVEL == 1) ? full->hist1[S->group0_of[j]]
                                                                 : full->hist2[S->group1_of[j]];
    const u64 want = S->rem[j];  // 1 <= want <= population of this histogram
    u64 mine = 0;
    for (int i = 0; i < PER_LANE; ++i) mine += ld_agent_u64(&hist[NBINS - 1 - (lane * PER_LANE + i)]);
    const u64 before = wave_excl_scan_u64(mine, lane);
    if (before < want && want <= before + mine) {
      u64 cum = before;
      for (int i = 0; i < PER_LANE; ++i) {
        const int bin = NBINS - 1 - (lane * PER_LANE + i);
        const u64 c = ld_agent_u64(&hist[bin]);
        if (want <= cum + c) { S->bin[j] = (uint32_t)bin; S->brem[j] = want - cum; S->bcnt[j] = c; break; }
        cum += c;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // group thresholds that fell into the same bin (they share the next histogram)
    uint32_t ng = 0, any_ordered = 0;
    for (int i = 0; i < kl.nk; ++i) {
      if (S->mode[i] < MODE_GE) continue;
      S->rem[i] = S->brem[i];
      if (LEVEL == 0) {
        S->prefix0[i] = S->bin[i];
        uint32_t g = 0;
        for (; g < ng; ++g) if (S->group0_prefix[g] == S->bin[i]) break;
        if (g == ng) S->group0_prefix[ng++] = S->bin[i];
        S->group0_of[i] = g;
      } else if (LEVEL == 1) {
        const uint32_t pre = (S->prefix0[i] << 10) | S->bin[i];
        S->prefix1[i] = pre;
        uint32_t g = 0;
        for (; g < ng; ++g) if (S->group1_prefix[g] == pre) break;
        if (g == ng) S->group1_prefix[ng++] = pre;
        S->group1_of[i] = g;
      } else {
        S->tau[i] = (S->prefix1[i] << 10) | S->bin[i];
        S->ceq[i] = S->bcnt[i];
        if (S->brem[i] != S->bcnt[i]) { S->mode[i] = MODE_ORDERED; any_ordered = 1; }
      }
    }
    if (LEVEL == 0) S->ngroups0 = ng;
    if (LEVEL == 1) S->ngroups1 = ng;
    if (LEVEL == 2) S->any_ordered = any_ordered;
  }
  __syncthreads();
}

__device__ __forceinline__ uint32_t block_sum_u32(uint32_t v, uint32_t *lds4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  const uint32_t r = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return r;
}
// Block exclusive scan of one u32 per thread (256 threads); *total gets the block total.
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t *lds4, uint32_t *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t inc = wave_incl_scan_u32(v, lane);
  if (lane == 63) lds4[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return woff + inc - v;
}

// Dynamic LDS: 2 KiB lut + nk * 1024 counters.
__global__ __launch_bounds__(SALUN_BLOCK) void k_fullscan(const float *__restrict__ acc, int64_t n, KList kl, MaskPtrs mp,
                                                          TopkPub *pub, FullState *full, const FastState *fs,
                                                          u64 *tie /*[nk][nchunk]*/, int always, int aligned,
                                                          int maligned, int values_only) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ Sel S;
  __shared__ uint32_t lds4[4];
  __shared__ uint32_t s_epoch;
  __shared__ uint32_t s_thr[MAXK];
  if (!always && !ld_agent_u32(&fs->fail)) return;
  const int tid = threadIdx.x;
  const int nk = kl.nk;
  uint8_t *lut = reinterpret_cast<uint8_t *>(lds);  // 2048 bytes
  uint32_t *h = lds + D0_BINS / 4;                  // up to nk * 1024 counters (pass 0 uses the first 2048)
  if (tid == 0) s_epoch = 0;
  const int64_t nvec = (n + 3) >> 2;
  const int64_t nchunk = (nvec + CHUNK_VEC - 1) / CHUNK_VEC;
  // ---- phase 0: zero the global histograms
  {
    u64 *z = reinterpret_cast<u64 *>(full);
    const int64_t words = (int64_t)(offsetof(FullState, bar) / 8);
    for (int64_t i = (int64_t)blockIdx.x * SALUN_BLOCK + tid; i < words; i += (int64_t)gridDim.x * SALUN_BLOCK) z[i] = 0;
  }
  if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
  // ---- phase 1: digit 0
  for (int i = tid; i < D0_BINS; i += SALUN_BLOCK) h[i] = 0;
  __syncthreads();
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
    load_chunk(acc, c, nvec, n, aligned, k);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k[u][e] != KEY_SKIP) atomicAdd(&h[k[u][e] >> 20], 1u);
  }
  __syncthreads();
  for (int i = tid; i < D0_BINS; i += SALUN_BLOCK)
    if (h[i]) atomicAdd(&full->hist0[i], (u64)h[i]);
  if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
  select_level<0>(full, &S, n, kl);
  // ---- phase 2: digit 1 of keys whose digit 0 is a boundary bin
  for (int i = tid; i < D0_BINS; i += SALUN_BLOCK) lut[i] = 0;
  __syncthreads();
  if (tid == 0)
    for (uint32_t g = 0; g < S.ngroups0; ++g) lut[S.group0_prefix[g]] = (uint8_t)(g + 1);
  for (uint32_t i = tid; i < S.ngroups0 * 1024u; i += SALUN_BLOCK) h[i] = 0;
  __syncthreads();
  if (S.ngroups0) {
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      uint32_t k[4][4];
      load_chunk(acc, c, nvec, n, aligned, k);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t key = k[u][e];
          if (key == KEY_SKIP) continue;
          const uint32_t g0 = lut[key >> 20];
          if (g0) atomicAdd(&h[(g0 - 1) * 1024u + ((key >> 10) & 1023u)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < S.ngroups0 * 1024u; i += SALUN_BLOCK)
      if (h[i]) atomicAdd(&(&full->hist1[0][0])[i], (u64)h[i]);
  }
  if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
  select_level<1>(full, &S, n, kl);
  // ---- phase 3: digit 2 of keys whose 21-bit prefix is a boundary prefix
  for (uint32_t i = tid; i < S.ngroups1 * 1024u; i += SALUN_BLOCK) h[i] = 0;
  __syncthreads();
  if (S.ngroups1) {
    const uint32_t ng = S.ngroups1;
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      uint32_t k[4][4];
      load_chunk(acc, c, nvec, n, aligned, k);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t key = k[u][e];
          if (key == KEY_SKIP || !lut[key >> 20]) continue;
          const uint32_t pre = key >> 10;
          for (uint32_t g = 0; g < ng; ++g)
            if (S.group1_prefix[g] == pre) { atomicAdd(&h[g * 1024u + (key & 1023u)], 1u); break; }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < ng * 1024u; i += SALUN_BLOCK)
      if (h[i]) atomicAdd(&(&full->hist2[0][0])[i], (u64)h[i]);
  }
  if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
  select_level<2>(full, &S, n, kl);
  // ---- phase 4 (rare): a threshold splits a run of equal keys — per-chunk tie populations, scanned over chunks
  if (S.any_ordered) {
    for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
      uint32_t k[4][4];
      load_chunk(acc, c, nvec, n, aligned, k);
      for (int j = 0; j < nk; ++j) {
        if (S.mode[j] != MODE_ORDERED) continue;
        const uint32_t tau = S.tau[j];
        uint32_t cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) cnt += (k[u][e] == tau);
        const uint32_t tot = block_sum_u32(cnt, lds4);
        if (tid == 0) tie[(int64_t)j * nchunk + c] = tot;
      }
    }
    if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
    for (int j = blockIdx.x; j < nk; j += gridDim.x) {  // in-place exclusive scan over chunks, one workgroup per row
      if (S.mode[j] != MODE_ORDERED) continue;
      __shared__ u64 s_wave[4];
      __shared__ u64 s_carry;
      if (tid == 0) s_carry = 0;
      __syncthreads();
      u64 *row = tie + (int64_t)j * nchunk;
      const int lane = tid & 63, wave = tid >> 6;
      for (int64_t base = 0; base < nchunk; base += SALUN_BLOCK) {
        const int64_t i = base + tid;
        const u64 v = (i < nchunk) ? ld_agent_u64(&row[i]) : 0;
        const u64 ex = wave_excl_scan_u64(v, lane);
        if (lane == 63) s_wave[wave] = ex + v;
        __syncthreads();
        u64 woff = 0;
        for (int w = 0; w < wave; ++w) woff += s_wave[w];
        const u64 carry = s_carry;
        if (i < nchunk) row[i] = carry + woff + ex;
        __syncthreads();
        if (tid == SALUN_BLOCK - 1) s_carry = carry + woff + ex + v;
        __syncthreads();
      }
    }
    if (!grid_barrier(&full->bar, &s_epoch, &pub->error)) return;
  }
  // ---- publish
  if (blockIdx.x == 0 && tid < MAXK) {
    if (tid < nk) { pub->mode[tid] = S.mode[tid]; pub->tau[tid] = S.tau[tid]; }
    if (tid == 0) { pub->nk = (uint32_t)nk; pub->route = 2; }
  }
  if (values_only) return;
  // ---- phase 5: the masks
  if (tid < nk) {
    const uint32_t mode = S.mode[tid];
    // NONE: nothing passes (real keys <= 0x7F800001); ALL: everything passes;
    // GE: every key equal to tau is inside the budget; ORDERED: strictly greater passes here.
    s_thr[tid] = (mode == MODE_NONE) ? 0xFFFFFFFEu : (mode == MODE_ALL) ? 0u
                 : (mode == MODE_GE) ? S.tau[tid] : S.tau[tid] + 1u;
  }
  __syncthreads();
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
    load_chunk(acc, c, nvec, n, aligned, k);
    for (int j = 0; j < nk; ++j) {
      const uint32_t thr = s_thr[j];
      uint32_t bits[4];  // 4 mask bytes per sub-vector
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bits[u] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) bits[u] |= (uint32_t)(k[u][e] != KEY_SKIP && k[u][e] >= thr) << (8 * e);
      }
      if (S.mode[j] == MODE_ORDERED) {  // workgroup-uniform branch
        const uint32_t tau = S.tau[j];
        u64 before = ld_agent_u64(&tie[(int64_t)j * nchunk + c]);  // equal keys in earlier chunks
        const u64 budget = S.rem[j];
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
      uint8_t *mj = mp.m[j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = c * CHUNK_VEC + u * SALUN_BLOCK + tid;
        if (v >= nvec) continue;
        const int64_t i = v << 2;
        if (maligned && i + 3 < n) {
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

__global__ void k_export_tau(const TopkPub *pub, int nk, float *out) {
  const int j = threadIdx.x;
  if (j >= nk) return;
  const uint32_t mode = pub->mode[j];
  float v;
  if (pub->error) v = __uint_as_float(0x7FC00000u);                 // the select failed: NaN poisons every consumer
  else if (mode == MODE_NONE) v = __uint_as_float(0x7F800000u);        // +inf: nothing selected
  else if (mode == MODE_ALL) v = -1.0f;                             // below every |x|
  else v = pub->tau[j] ? __uint_as_float(pub->tau[j] - 1u) : __uint_as_float(0x7FC00000u);
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
//                                              host side
// =====================================================================================================
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }
inline int64_t chunks_of(int64_t n) { return (n + CHUNK - 1) / CHUNK; }

inline int bins_a_for(int nk) { return nk <= 2 ? 1024 : nk <= 4 ? 512 : 256; }
inline int sample_size(int64_t n) {
  int S = SAMPLE_MAX;
  while (S > 1024 && (int64_t)S * 2 > n) S >>= 1;
  return S;
}
inline int main_grid_for(int64_t n) {
  const int64_t nfull = n / CHUNK;
  return (int)(nfull < MAIN_GRID ? (nfull < 1 ? 1 : nfull) : MAIN_GRID);
}
// worst-case (p = 1/2) share of the values inside one bracket, for sizing: 2 * margin / S plus the bin-edge slack
inline double bracket_fraction(int64_t S) {
  const double margin = std::ceil(BRACKET_SIGMAS * 0.5 * std::sqrt((double)S)) + 8.0;
  return 2.0 * margin / (double)S + 0.002;
}

// Sizes of one fast-route instance over `n` elements with `nk` thresholds whose brackets hold a fraction `f` each.
struct FastLayout {
  int grid;          // k_main workgroups
  int ga;            // k_hist_a workgroups (1024 threads, one wave per slab row)
  int gr;            // k_resolve workgroups per threshold = short-list segments per threshold
  uint32_t cap;      // slab entries per (workgroup, threshold)
  size_t off_fs, off_keys, off_cnt, off_gt, off_slabs, off_list2, off_seg, bytes;
};
inline FastLayout fast_layout(int64_t n, int nk, double f) {
  FastLayout L;
  L.grid = main_grid_for(n);
  const double per_slab = f * (double)n / (double)L.grid;
  L.cap = (uint32_t)(1.5 * per_slab) + 256u;
  const int rows = L.grid * nk;
  // LDS atomics per workgroup vs the depth of the flush's atomic chains (one global atomic per bin per workgroup)
  const double cands = f * (double)n * (double)nk;
  int ga = (int)(cands / 8192.0);
  if (ga < 32) ga = 32;
  if (ga > 256) ga = 256;
  if (ga < nk) ga = nk;  // workgroups 0..nk-1 also reduce the c_gt rows
  L.ga = ga;
  // k_resolve workgroups per threshold (= short-list segments per threshold): one wave per slab row, more when the
  // rows are long (large n)
  int gpj = (L.grid + 15) / 16;
  if (per_slab > 1024.0) gpj *= 4;
  if (gpj > 256 / nk) gpj = 256 / nk;
  if (gpj < 1) gpj = 1;
  L.gr = gpj;
  size_t b = 0;
  L.off_fs = b;    b += align256(sizeof(FastState));
  L.off_keys = b;  b += align256(sizeof(uint32_t) * (size_t)SAMPLE_MAX);
  L.off_cnt = b;   b += align256(sizeof(uint32_t) * (size_t)rows);
  L.off_gt = b;    b += align256(sizeof(uint32_t) * (size_t)rows);
  L.off_slabs = b; b += align256(sizeof(uint2) * (size_t)rows * (size_t)L.cap);
  L.off_list2 = b; b += align256(sizeof(uint2) * (size_t)nk * (size_t)L.gr * (size_t)STAGE_CAP);
  L.off_seg = b;   b += align256(sizeof(uint32_t) * (size_t)nk * (size_t)L.gr);
  L.bytes = b;
  return L;
}

inline bool fast_applies(int64_t n) { return n >= FAST_MIN_N && n < (int64_t(1) << 32); }
inline bool two_level_applies(int64_t n, int nk) { return n >= TWO_LEVEL_MIN_N && 2 * nk <= MAXK; }

// Workspace: [TopkPub | FullState | tie rows | outer fast instance | (two-level) sample + inner pub/full + inner fast instance]
struct WsLayout {
  size_t off_pub, off_full, off_tie, off_fast, off_sample, off_ipub, off_ifull, off_ifast, bytes;
  FastLayout outer, inner;
  bool fast, two_level;
};
inline WsLayout ws_layout(int64_t n, int nk) {
  WsLayout W;
  size_t b = 0;
  W.off_pub = b;  b += align256(sizeof(TopkPub));
  W.off_full = b; b += align256(sizeof(FullState));
  W.off_tie = b;  b += align256(sizeof(u64) * (size_t)nk * (size_t)(chunks_of(n) + 1));
  W.fast = fast_applies(n);
  W.two_level = W.fast && two_level_applies(n, nk);
  W.off_fast = W.off_sample = W.off_ipub = W.off_ifull = W.off_ifast = b;
  if (W.fast) {
    const int64_t S2 = int64_t(1) << SAMPLE2_LOG2;
    W.outer = fast_layout(n, nk, W.two_level ? bracket_fraction(S2) : bracket_fraction(sample_size(n)));
    W.off_fast = b; b += W.outer.bytes;
    if (W.two_level) {
      W.off_sample = b; b += align256(sizeof(float) * (size_t)S2);
      W.off_ipub = b;   b += align256(sizeof(TopkPub));
      W.off_ifull = b;  b += align256(sizeof(FullState));
      W.inner = fast_layout(S2, 2 * nk, bracket_fraction(sample_size(S2)));
      W.off_ifast = b;  b += W.inner.bytes;
    }
  }
  W.bytes = b;
  return W;
}

int g_cu_count = 0;
inline int fullscan_grid(int64_t n) {
  if (g_cu_count == 0) {
    // 2 KiB + 16 x 4 KiB of dynamic LDS at the maximum threshold count: above the 64 KiB default
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fullscan), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      g_cu_count = cus;
    else
      g_cu_count = 64;
  }
  int64_t g = 2 * (int64_t)g_cu_count;  // resident with room to spare: the grid barrier needs every workgroup running
  if (g > 512) g = 512;
  const int64_t nc = chunks_of(n);
  if (g > nc) g = nc < 1 ? 1 : nc;
  return (int)g;
}

template <bool VO>
inline void launch_main(int nk, int grid, hipStream_t st, const float *acc, int64_t n, FastState *fs, const MaskPtrs &mp,
                        uint2 *slabs, uint32_t *slab_cnt, uint32_t *wg_gt, uint32_t cap, int bins_a) {
#define SALUN_MAIN(NKT)                                                                                             \
  hipLaunchKernelGGL((k_main<NKT, VO>), dim3(grid), dim3(SALUN_BLOCK), 0, st, acc, n, fs, mp, slabs, slab_cnt, wg_gt, \
                     cap, nk, bins_a)
  if (nk <= 1) SALUN_MAIN(1);
  else if (nk <= 2) SALUN_MAIN(2);
  else if (nk <= 3) SALUN_MAIN(3);
  else if (nk <= 4) SALUN_MAIN(4);
  else if (nk <= 6) SALUN_MAIN(6);
  else if (nk <= 8) SALUN_MAIN(8);
  else if (nk <= 10) SALUN_MAIN(10);
  else if (nk <= 12) SALUN_MAIN(12);
  else SALUN_MAIN(16);
#undef SALUN_MAIN
}

// k_main .. k_finish + the fallback launch of one fast instance whose FastState already holds the brackets.
inline int run_fast_tail(const float *acc, int64_t n, const KList &kl, const MaskPtrs &mp, bool values_only, char *fast_base,
                         const FastLayout &L, TopkPub *pub, FullState *full, u64 *tie, bool aligned, bool maligned,
                         hipStream_t st) {
  const int nk = kl.nk;
  FastState *fs = reinterpret_cast<FastState *>(fast_base + L.off_fs);
  uint32_t *slab_cnt = reinterpret_cast<uint32_t *>(fast_base + L.off_cnt);
  uint32_t *wg_gt = reinterpret_cast<uint32_t *>(fast_base + L.off_gt);
  uint2 *slabs = reinterpret_cast<uint2 *>(fast_base + L.off_slabs);
  uint2 *list2 = reinterpret_cast<uint2 *>(fast_base + L.off_list2);
  uint32_t *seg_cnt = reinterpret_cast<uint32_t *>(fast_base + L.off_seg);
  const int bins_a = bins_a_for(nk);
  if (values_only) launch_main<true>(nk, L.grid, st, acc, n, fs, mp, slabs, slab_cnt, wg_gt, L.cap, bins_a);
  else launch_main<false>(nk, L.grid, st, acc, n, fs, mp, slabs, slab_cnt, wg_gt, L.cap, bins_a);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_hist_a, dim3(L.ga), dim3(1024), sizeof(uint32_t) * (size_t)nk * bins_a, st, fs, slabs, slab_cnt,
                     wg_gt, L.cap, L.grid, nk, bins_a);
  SALUN_LAUNCH_CHECK();
  if (values_only)
    hipLaunchKernelGGL(k_resolve<true>, dim3(L.gr * nk), dim3(1024), 0, st, fs, slabs, slab_cnt, L.cap, L.grid, nk, bins_a,
                       L.gr, list2, seg_cnt, mp);
  else
    hipLaunchKernelGGL(k_resolve<false>, dim3(L.gr * nk), dim3(1024), 0, st, fs, slabs, slab_cnt, L.cap, L.grid, nk,
                       bins_a, L.gr, list2, seg_cnt, mp);
  SALUN_LAUNCH_CHECK();
  if (values_only) hipLaunchKernelGGL(k_finish<true>, dim3(nk), dim3(1024), 0, st, fs, pub, list2, seg_cnt, L.gr, mp);
  else hipLaunchKernelGGL(k_finish<false>, dim3(nk), dim3(1024), 0, st, fs, pub, list2, seg_cnt, L.gr, mp);
  SALUN_LAUNCH_CHECK();
  const size_t lds_bytes = D0_BINS + sizeof(uint32_t) * (size_t)(nk < 2 ? 2 : nk) * 1024;
  hipLaunchKernelGGL(k_fullscan, dim3(fullscan_grid(n)), dim3(SALUN_BLOCK), lds_bytes, st, acc, n, kl, mp, pub, full, fs,
                     tie, 0, (int)aligned, (int)maligned, (int)values_only);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

}  // namespace

// ================================================================== C-ABI =======
SALUN_EXPORT size_t salun_mask_topk_workspace_bytes(int64_t n, int nk) {
  if (n < 0 || nk < 1 || nk > MAXK) return 0;
  return ws_layout(n, nk).bytes;
}

SALUN_EXPORT int salun_mask_topk_ex(const float *acc, int64_t n, const int64_t *ks, int nk, uint8_t *const *masks_out,
                                    void *ws, size_t ws_bytes, unsigned flags, salun_stream_t stream) {
  const bool values_only = (flags & SALUN_TOPK_VALUES_ONLY) != 0;
  if (n < 0 || nk < 1 || nk > MAXK || !ks || !ws || (n > 0 && !acc)) return SALUN_EINVAL;
  if (!values_only) {
    if (!masks_out) return SALUN_EINVAL;
    for (int j = 0; j < nk; ++j)
      if (n > 0 && !masks_out[j]) return SALUN_EINVAL;
  }
  const WsLayout W = ws_layout(n, nk);
  if (ws_bytes < W.bytes) return SALUN_ENOSPC;
  hipStream_t st = salun_hip_stream(stream);
  char *base = static_cast<char *>(ws);
  TopkPub *pub = reinterpret_cast<TopkPub *>(base + W.off_pub);
  FullState *full = reinterpret_cast<FullState *>(base + W.off_full);
  u64 *tie = reinterpret_cast<u64 *>(base + W.off_tie);
  KList kl;
  MaskPtrs mp;
  kl.nk = nk;
  bool maligned = true;
  for (int j = 0; j < MAXK; ++j) {
    kl.k[j] = (j < nk) ? (long long)ks[j] : 0;
    mp.m[j] = (j < nk && !values_only) ? masks_out[j] : nullptr;
    if (j < nk && !values_only && !salun_aligned4(masks_out[j])) maligned = false;
  }
  if (n == 0) {  // nothing to rank: publish "nothing selected"
    if (hipMemsetAsync(pub, 0, sizeof(TopkPub), st) != hipSuccess) return SALUN_EIO;
    return SALUN_OK;
  }
  const bool aligned = salun_aligned16(acc);
  const bool fast = W.fast && aligned && maligned && !(flags & SALUN_TOPK_FORCE_FULL_SCAN);
  if (!fast) {
    // full scan directly: one memset (publication block + barrier word) and the persistent kernel
    if (hipMemsetAsync(pub, 0, sizeof(TopkPub), st) != hipSuccess) return SALUN_EIO;
    if (hipMemsetAsync(&full->bar, 0, 8, st) != hipSuccess) return SALUN_EIO;
    const size_t lds_bytes = D0_BINS + sizeof(uint32_t) * (size_t)(nk < 2 ? 2 : nk) * 1024;
    // The kernel synchronises its workgroups with a software grid barrier, which needs all of them resident at once:
    // a COOPERATIVE launch makes the runtime guarantee that (it waits for room instead of letting side-stream /
    // RCCL kernels or another tenant hold back part of the grid).  If the runtime refuses the cooperative launch the
    // plain launch below still works — its barrier is bounded and a time-out is reported (pub->error -> NaN
    // thresholds, salun_mask_topk_status), never silent.
    const FastState *no_fs = nullptr;
    int always = 1, al = (int)aligned, mal = (int)maligned, vo = (int)values_only;
    void *args[] = {(void *)&acc, (void *)&n, (void *)&kl, (void *)&mp, (void *)&pub, (void *)&full, (void *)&no_fs,
                    (void *)&tie, (void *)&always, (void *)&al, (void *)&mal, (void *)&vo};
    const int grid = fullscan_grid(n);
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_fullscan), dim3(grid), dim3(SALUN_BLOCK), args,
                                   (unsigned)lds_bytes, st) != hipSuccess) {
      (void)hipGetLastError();
      hipLaunchKernelGGL(k_fullscan, dim3(grid), dim3(SALUN_BLOCK), lds_bytes, st, acc, n, kl, mp, pub, full, no_fs, tie,
                         1, al, mal, vo);
    }
    SALUN_LAUNCH_CHECK();
    return SALUN_OK;
  }
  const int bins_a = bins_a_for(nk);
  char *fast_base = base + W.off_fast;
  FastState *fs = reinterpret_cast<FastState *>(fast_base + W.outer.off_fs);
  const bool two_level = W.two_level;
  if (!two_level) {
    const int S = sample_size(n);
    uint32_t *skeys = reinterpret_cast<uint32_t *>(fast_base + W.outer.off_keys);
    hipLaunchKernelGGL(k_sample, dim3(S / 1024), dim3(1024), 0, st, acc, n, S, skeys, fs, pub, full, nk);
    SALUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bracket, dim3(nk), dim3(1024), 0, st, skeys, n, kl, S, bins_a, fs, pub, full);
    SALUN_LAUNCH_CHECK();
  } else {
    // ---- brackets = exact order statistics of a 2^20-element sample (this same route on the sample, values only)
    const int64_t S2 = int64_t(1) << SAMPLE2_LOG2;
    float *samp = reinterpret_cast<float *>(base + W.off_sample);
    TopkPub *ipub = reinterpret_cast<TopkPub *>(base + W.off_ipub);
    FullState *ifull = reinterpret_cast<FullState *>(base + W.off_ifull);
    char *ifast = base + W.off_ifast;
    FastState *ifs = reinterpret_cast<FastState *>(ifast + W.inner.off_fs);
    KList k2;
    RankList rl;
    k2.nk = 2 * nk;
    for (int j = 0; j < MAXK; ++j) { k2.k[j] = 0; rl.hi[j] = rl.lo[j] = 0; }
    for (int j = 0; j < nk; ++j) {
      long long k = ks[j] > n ? n : ks[j];
      if (k <= 0) { k2.k[2 * j] = k2.k[2 * j + 1] = 0; continue; }
      const double p = (double)k / (double)n;
      const double sigma = std::sqrt((double)S2 * p * (1.0 - p));
      const long long margin = (long long)std::ceil(BRACKET_SIGMAS * sigma) + 8;
      const long long rho = std::llround(p * (double)S2);
      long long rhi = rho - margin, rlo = rho + margin;
      if (rhi < 1) rhi = 0;
      if (rlo > S2) rlo = 0;
      rl.hi[j] = rhi; rl.lo[j] = rlo;
      k2.k[2 * j] = rhi ? rhi : 1;       // unbounded sides are not read back
      k2.k[2 * j + 1] = rlo ? rlo : 1;
    }
    const int S1 = sample_size(S2);
    uint32_t *ikeys = reinterpret_cast<uint32_t *>(ifast + W.inner.off_keys);
    hipLaunchKernelGGL(k_gather_sample, dim3((unsigned)(S2 / SALUN_BLOCK)), dim3(SALUN_BLOCK), 0, st, acc, n / S2, S2, samp,
                       (int)(S2 / S1), ikeys, ifs, ipub, ifull, 2 * nk);
    SALUN_LAUNCH_CHECK();
    const int ibins = bins_a_for(2 * nk);
    hipLaunchKernelGGL(k_bracket, dim3(2 * nk), dim3(1024), 0, st, ikeys, S2, k2, S1, ibins, ifs, ipub, ifull);
    SALUN_LAUNCH_CHECK();
    {  // the sample's streaming pass + candidate histogram; its resolution stops there (k_bracket_from_hist)
      const FastLayout &L = W.inner;
      uint32_t *slab_cnt = reinterpret_cast<uint32_t *>(ifast + L.off_cnt);
      uint32_t *wg_gt = reinterpret_cast<uint32_t *>(ifast + L.off_gt);
      uint2 *slabs = reinterpret_cast<uint2 *>(ifast + L.off_slabs);
      MaskPtrs none;
      for (int j = 0; j < MAXK; ++j) none.m[j] = nullptr;
      launch_main<true>(2 * nk, L.grid, st, samp, S2, ifs, none, slabs, slab_cnt, wg_gt, L.cap, ibins);
      SALUN_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_hist_a, dim3(L.ga), dim3(1024), sizeof(uint32_t) * (size_t)(2 * nk) * ibins, st, ifs, slabs,
                         slab_cnt, wg_gt, L.cap, L.grid, 2 * nk, ibins);
      SALUN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_bracket_from_hist, dim3(1), dim3(1024), 0, st, ifs, ibins, n, kl, rl, bins_a, fs, pub, full);
    SALUN_LAUNCH_CHECK();
  }
  return run_fast_tail(acc, n, kl, mp, values_only, fast_base, W.outer, pub, full, tie, aligned, maligned, st);
}

SALUN_EXPORT int salun_mask_topk(const float *acc, int64_t n, const int64_t *ks, int nk,
                                 uint8_t *const *masks_out, void *ws, size_t ws_bytes,
                                 salun_stream_t stream) {
  return salun_mask_topk_ex(acc, n, ks, nk, masks_out, ws, ws_bytes, 0u, stream);
}

SALUN_EXPORT int salun_mask_topk_thresholds(const void *ws, int nk, float *tau_out, salun_stream_t stream) {
  if (!ws || !tau_out || nk < 1 || nk > MAXK) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_export_tau, dim3(1), dim3(64), 0, salun_hip_stream(stream),
                     static_cast<const TopkPub *>(ws), nk, tau_out);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_mask_topk_status(const void *ws, int *route_out, int *error_out, salun_stream_t stream) {
  if (!ws || !route_out || !error_out) return SALUN_EINVAL;
  TopkPub host;
  hipStream_t st = salun_hip_stream(stream);
  if (hipMemcpyAsync(&host, ws, sizeof(TopkPub), hipMemcpyDeviceToHost, st) != hipSuccess) return SALUN_EIO;
  if (hipStreamSynchronize(st) != hipSuccess) return SALUN_EIO;
  *route_out = (int)host.route;
  *error_out = (int)host.error;
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
