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
//   k_sample    gathers a 16 K-element hashed sample (keys)
//   k_bracket   one workgroup per threshold ranks the sample with a two-level LDS histogram and brackets the
//               threshold,  lo_j <= tau_j <= hi_j,  5 sigma either side of the target rank (~4 % of the mass);
//               k <= 0 / k >= n need no bracket (nothing / everything selected; k == n publishes the minimum)
//   k_main      the one streaming pass: writes mask_j = [key > mid_j] (final outside the bracket), counts
//               c_gt_j = #{key > hi_j}, compacts the in-bracket candidates (key, flat index) into per-workgroup slabs
//               (what a slab cannot take — a layer whose magnitudes sit at the threshold — goes to the threshold's
//               shared spill row) and leaves a 128-bin u16 histogram row of them per (workgroup, threshold); exact
//               zeros are counted, never compacted (a real accumulator holds several per cent of them)
//   k_resolve   every workgroup sums the rows, picks the bin holding rank k_j - c_gt_j; candidates above it get their
//               mask byte, candidates inside it go to a short list + a 4096-bin histogram of that bin
//   k_finish    one workgroup per 64 short-list segments (<= 8 per threshold): picks the final bin (<= a few keys
//               wide), settles its share of the list, hands the final bin's residents over write-through; the last
//               ticket ranks them exactly (key descending, flat index ascending) and publishes tau_j
//   A threshold that lands in the block of exact zeros (tau = |0|) is finished by the tie pass of the launch behind
//   k_finish: per-chunk zero counts, a scan, and the first k_j - #{nonzero} zeros by flat index get their byte.
//   For n >= 2^27 the bracket comes from an exact selection (this same route, values only) on a 2^20-element sample
//   instead, which narrows it to ~0.5 % of the mass.
//   Anything unusual — a bracket that misses, a slab or list that overflows (heavy ties at a non-zero key), a
//   threshold among NaNs — raises `fail` on the device and the full scan below redoes the job; no host
//   synchronisation anywhere.
//
// FULL SCAN (small or unaligned inputs, and the fallback): k_fullscan, ONE persistent launch of <= 2 workgroups per
//   CU: three histogram passes (11 + 10 + 10 bit digits, most significant first; thresholds that share a prefix share
//   a histogram), per-workgroup LDS histograms flushed with 64-bit global atomics, every workgroup repeating the tiny
//   selection step itself, grid barriers between passes (monotonic counter, agent-scope release/acquire, bounded
//   spin), per-chunk tie prefixes when a threshold splits a run of equal keys, one write pass.  When launched behind
//   the fast route it returns at once unless `fail` (or the zero-tie flag) is set.
#include "salun_common.h"
#include <cmath>
#include <cstdlib>

// Timing switches (0 in the product build; profiles/r05_topk_experiments.txt was measured with run-time versions of
// them — results are WRONG with any bit set, they only show where a kernel's time goes):
//   k_main    8 no compaction   16 no mask stores   32 no software prefetch of the next chunk
//   k_resolve 1 stop after the row sums   2 ... after the selection   4 no flush of the second-level histogram
//   k_finish 64 stop after the prologue loads   128 ... after the two scans
#ifndef SALUN_TOPK_EXP
#define SALUN_TOPK_EXP 0
#endif

namespace {

constexpr int MAXK = SALUN_MAX_THRESHOLDS;
constexpr int TOPK_EXP = SALUN_TOPK_EXP;
constexpr int D0_BINS = 2048;  // key >> 20
constexpr int CHUNK_VEC = 4 * SALUN_BLOCK;  // float4 per chunk (4 sub-vectors per lane)
constexpr int CHUNK = CHUNK_VEC * 4;        // 4096 elements: streaming / tie-ordering granule

constexpr int64_t FAST_MIN_N = 8192;
constexpr int SAMPLE_MAX = 16384;             // k_bracket's sample (16 keys per thread)
constexpr int64_t TWO_LEVEL_MIN_N = int64_t(1) << 27;
constexpr int SAMPLE2_LOG2 = 20;              // outer sample of the two-level route
constexpr int HIST2_BINS = 4096;
constexpr int BINS_A = 128;                   // first-level bins over a bracket (u16 rows written by k_main)
constexpr int FINAL_CAP = 1024;               // final-bin residents ranked exactly in LDS
constexpr int MAIN_GRID = 1024;
constexpr double BRACKET_SIGMAS = 5.0;        // a miss (3e-7 per side) costs one full scan, never a wrong mask

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
  uint32_t mode[MAXK];   // MODE_NONE, MODE_GE (bracketed) or MODE_ALL (k >= n: every element, no select)
  uint32_t was_all[MAXK];  // k > n on entry: published as MODE_ALL
  uint32_t lo[MAXK], hi[MAXK], shiftA[MAXK];  // candidate bracket (lo >= 2: never the zero / NaN key), its bin shift
  uint32_t mid[MAXK];    // k_main's guess of the threshold (centre of the bracket): it writes mask = [key > mid] and
                         // the later kernels touch only the candidates whose final bit differs from that guess
  uint32_t gthr[MAXK];   // the same guess as a >= bound (mid + 1; 0: every key passes, KEY_SKIP: none does)
  uint32_t zero_in[MAXK];  // the sample's bracket reached down to the zero key
  uint32_t lo2[MAXK], hi2[MAXK], shift2[MAXK];
  u64 r2[MAXK];          // rank wanted inside [lo2, hi2]
  u64 zt_budget[MAXK];   // > 0: the threshold is the zero key; this many zeros (lowest flat index first) are selected
  uint32_t zt_any;       // some zt_budget is set -> the tie pass behind k_finish runs
  uint32_t spill_cnt[MAXK];  // entries in the threshold's shared spill row (candidates a full slab could not take)
  uint32_t fin_n[MAXK];      // k_finish: residents of the final bin gathered so far (its workgroups append)
  uint32_t fin_ticket[MAXK]; // k_finish: workgroups of the threshold that are done gathering
  uint32_t fail;         // -> the full scan redoes the job
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
constexpr uint32_t ZERO_KEY = 1u;           // key of +-0
constexpr uint32_t MIN_CAND_KEY = 2u;       // smallest key the fast route compacts (zeros are counted, NaNs fall back)
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

__device__ __forceinline__ uint32_t shift_for(uint32_t lo, uint32_t hi, int bins) {
  const uint32_t w = hi - lo;
  uint32_t s = 0;
  while ((w >> s) >= (uint32_t)bins) ++s;
  return s;
}

// Workgroup 0 of the sampling kernel clears the small head of the state and the publication block for this call
// (the second-level histograms are zeroed by k_main's workgroups); k_bracket's workgroups then fill in one
// threshold each.
__device__ __forceinline__ void reset_head(FastState *fs, TopkPub *pub, FullState *full, int nk, int tid, int nthreads) {
  uint32_t *z = reinterpret_cast<uint32_t *>(fs);
  const int words_head = (int)(offsetof(FastState, hist2) / 4);
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

// Exclusive scan of one u32 per thread over a 1024-thread workgroup; *total gets the workgroup total.
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

// What a bracket [lo, hi] of keys becomes in the state: the candidate range never holds the zero key or the NaN key
// (k_main counts the exact zeros instead of compacting them: a model's accumulator can hold several per cent of them,
// which no slab could take), `zero_in` remembers that the bracket reached down to them.
__device__ __forceinline__ void publish_bracket(FastState *fs, int j, long long k, bool was_all, uint32_t mode, uint32_t lo,
                                                uint32_t hi) {
  uint32_t zero_in = 0, mid = KEY_SKIP, gthr = KEY_SKIP, sh = 0;
  if (mode == MODE_GE) {
    if (hi > KEY_MAX) hi = KEY_MAX;
    zero_in = lo <= ZERO_KEY ? 1u : 0u;
    if (lo < MIN_CAND_KEY) lo = MIN_CAND_KEY;
    if (hi < lo) hi = lo;
    mid = lo + (hi - lo) / 2;
    gthr = mid + 1u;
    sh = shift_for(lo, hi, BINS_A);
  } else {
    lo = hi = KEY_SKIP;                       // nothing is inside, nothing is above
    gthr = (mode == MODE_ALL) ? 0u : KEY_SKIP;  // every key passes / no key passes
  }
  fs->k[j] = k;
  fs->mode[j] = mode;
  fs->was_all[j] = was_all ? 1u : 0u;
  fs->lo[j] = lo;
  fs->hi[j] = hi;
  fs->mid[j] = mid;
  fs->gthr[j] = gthr;
  fs->zero_in[j] = zero_in;
  fs->shiftA[j] = sh;
}

// ---------------------------------------------------------------------------------- k_bracket
// One workgroup of 1024 threads per threshold: ranks the S sampled keys with a 2048-bin and a 256-bin LDS histogram
// (19 key bits: the bin edges, taken outward, only widen a bracket by ~0.05 % of the values) and brackets the
// threshold between the sample's order statistics `BRACKET_SIGMAS` standard deviations either side of its rank.
// k <= 0 and k >= n need no bracket: nothing / everything is selected (k == n publishes the minimum as its threshold).
constexpr int H0_COPIES = 8;  // lane-indexed copies of the first-level histogram: its hot bins would serialise the
                              // LDS atomics of a wave (a model's |gradients| sit in ~40 of the 2048 bins)

__global__ __launch_bounds__(1024) void k_bracket(const uint32_t *__restrict__ keys, int64_t n, KList kl, int S,
                                                  FastState *fs) {
  __shared__ uint32_t h0c[H0_COPIES][D0_BINS];
  __shared__ uint32_t h1[2][256];
  __shared__ uint32_t s_w[16], s_wt[8];
  __shared__ uint32_t s_rank[2];   // descending rank in the sample; 0 = unbounded on that side
  __shared__ uint32_t s_b0[2], s_rem[2], s_b1[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x;
  const long long kin = kl.k[j];
  if (kin <= 0 || kin >= (long long)n) {  // workgroup-uniform
    if (tid == 0) publish_bracket(fs, j, kin <= 0 ? 0 : (long long)n, kin > (long long)n, kin <= 0 ? MODE_NONE : MODE_ALL,
                                  KEY_SKIP, KEY_SKIP);
    return;
  }
  const int per = S >> 10;  // 1 .. 16 keys per thread
  uint32_t key[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) key[i] = (i < per) ? keys[i * 1024 + tid] : KEY_SKIP;
  for (int i = tid; i < H0_COPIES * D0_BINS; i += 1024) (&h0c[0][0])[i] = 0;
  if (tid < 512) (&h1[0][0])[tid] = 0;
  if (tid == 0) {
    const double p = (double)kin / (double)n;
    const double sigma = sqrt((double)S * p * (1.0 - p));
    const long long margin = (long long)ceil(BRACKET_SIGMAS * sigma) + 8;
    const long long rho = llround(p * (double)S);
    long long rhi = rho - margin, rlo = rho + margin;  // larger keys: smaller descending rank
    if (rhi < 1) rhi = 0;
    if (rlo > S) rlo = 0;
    s_rank[0] = (uint32_t)rhi;
    s_rank[1] = (uint32_t)rlo;
  }
  __syncthreads();
  {
    uint32_t *mine = h0c[lane & (H0_COPIES - 1)];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (key[i] != KEY_SKIP) atomicAdd(&mine[key[i] >> 20], 1u);
  }
  __syncthreads();
  {  // ---- level 0: bins walked from the top, thread t owns bins 2047-2t and 2046-2t
    const int bA = D0_BINS - 1 - 2 * tid, bB = bA - 1;
    uint32_t cA = 0, cB = 0;
#pragma unroll
    for (int c = 0; c < H0_COPIES; ++c) { cA += h0c[c][bA]; cB += h0c[c][bB]; }
    uint32_t total;
    const uint32_t before = block1024_excl_scan(cA + cB, s_w, &total);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint32_t want = s_rank[q];
      if (want != 0 && before < want && want <= before + cA + cB) {
        if (want <= before + cA) { s_b0[q] = (uint32_t)bA; s_rem[q] = want - before; }
        else { s_b0[q] = (uint32_t)bB; s_rem[q] = want - before - cA; }
      }
    }
  }
  __syncthreads();
  const uint32_t b0a = s_rank[0] ? s_b0[0] : KEY_SKIP, b0b = s_rank[1] ? s_b0[1] : KEY_SKIP;
  const bool same = (b0a == b0b);
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (key[i] != KEY_SKIP) {
      const uint32_t top = key[i] >> 20;
      if (top == b0a) atomicAdd(&h1[0][(key[i] >> 12) & 255u], 1u);
      else if (top == b0b) atomicAdd(&h1[1][(key[i] >> 12) & 255u], 1u);
    }
  __syncthreads();
  // ---- level 1: 256 bins from the top, one per lane; waves 0-3 serve the upper rank, waves 4-7 the lower
  uint32_t c1 = 0, incl1 = 0;
  const int q1 = wave >> 2;
  if (wave < 8) {
    const uint32_t *h = h1[(q1 == 1 && !same) ? 1 : 0];
    c1 = h[255 - ((wave & 3) * 64 + lane)];
    incl1 = wave_incl_scan_u32(c1, lane);
    if (lane == 63) s_wt[wave] = incl1;
  }
  __syncthreads();
  if (wave < 8 && s_rank[q1] != 0) {
    uint32_t off = 0;
    for (int w = q1 * 4; w < wave; ++w) off += s_wt[w];
    const uint32_t before = off + incl1 - c1, want = s_rem[q1];
    if (before < want && want <= before + c1) s_b1[q1] = (uint32_t)(255 - ((wave & 3) * 64 + lane));
  }
  __syncthreads();
  if (tid == 0) {
    const uint32_t hi = s_rank[0] ? ((s_b0[0] << 20) | (s_b1[0] << 12) | 0xFFFu) : KEY_MAX;
    const uint32_t lo = s_rank[1] ? ((s_b0[1] << 20) | (s_b1[1] << 12)) : 0u;
    publish_bracket(fs, j, kin, false, MODE_GE, lo, hi);
  }
}

// ---- per-workgroup level-A rows (written by k_main): [workgroup][threshold][BINS_A] u16 ------------------------------
// One WAVE sums the rows of threshold j over `grid` workgroups: lane l owns bins 2l and 2l+1 (one dword per row).
__device__ __forceinline__ void wave_sum_rows(const uint16_t *__restrict__ rows, int grid, int nk, int j, int lane,
                                              uint32_t *c_even, uint32_t *c_odd) {
  const uint32_t *base = reinterpret_cast<const uint32_t *>(rows);
  uint32_t a = 0, b = 0;
  int g = 0;
  for (; g + 32 <= grid; g += 32) {  // 32 independent loads in flight per lane
    uint32_t v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = base[((size_t)(g + i) * nk + j) * (BINS_A / 2) + lane];
#pragma unroll
    for (int i = 0; i < 32; ++i) { a += v[i] & 0xFFFFu; b += v[i] >> 16; }
  }
  for (; g < grid; ++g) {
    const uint32_t v = base[((size_t)g * nk + j) * (BINS_A / 2) + lane];
    a += v & 0xFFFFu;
    b += v >> 16;
  }
  *c_even = a;
  *c_odd = b;
}

// Two-level route (n >= 2^27).  The 2^20-element sample is bracketed by its own 16 K sub-sample (k_bracket) and
// streamed once by k_main (values only), which leaves per-workgroup histogram rows of each bracket; this kernel reads,
// for the two target ranks of every threshold (2j = upper, 2j+1 = lower), the bin holding that rank and takes the bin's
// OUTER edge as the bracket of the full vector: 128 bins over ~4 % of the sample's mass widen a bracket by < 0.04 % of
// the mass, against its own width of ~0.5 %.  A rank that falls among the sample's exact zeros puts the edge on the
// zero key; a rank outside its sample bracket raises `fail` (the full scan takes over).  One wave per inner threshold.
__global__ __launch_bounds__(1024) void k_bracket_from_hist(const FastState *inner, const uint16_t *__restrict__ irows,
                                                            const uint32_t *__restrict__ iwg_gt,
                                                            const uint32_t *__restrict__ iwg_zero, int igrid, int64_t n,
                                                            KList kl, RankList rl, FastState *fs, TopkPub *pub,
                                                            FullState *full) {
  __shared__ uint32_t s_edge[2 * MAXK];
  __shared__ uint32_t s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = kl.nk;
  reset_head(fs, pub, full, nk, tid, 1024);
  if (tid == 0) s_bad = inner->fail;
  __syncthreads();
  const int q = wave;  // 16 waves, 2 * nk <= 16 inner thresholds
  if (q < 2 * nk) {
    const bool upper = (q & 1) == 0;
    const long long want_rank = upper ? rl.hi[q >> 1] : rl.lo[q >> 1];
    if (want_rank != 0 && inner->mode[q] == MODE_GE) {  // else: unbounded side / trivial threshold
      u64 gt = 0, zz = 0;
      for (int g = lane; g < igrid; g += 64) { gt += iwg_gt[(size_t)g * (2 * nk) + q]; zz += iwg_zero[g]; }
      for (int off = 32; off > 0; off >>= 1) { gt += __shfl_xor(gt, off, 64); zz += __shfl_xor(zz, off, 64); }
      uint32_t ce, co;
      wave_sum_rows(irows, igrid, 2 * nk, q, lane, &ce, &co);
      // descending walk: lane l's pair sits at descending positions 2(63-l) (odd bin first)
      const int rl_ = 63 - lane;  // lane rl_ owns the pair this lane needs in descending order
      const uint32_t de = __shfl(ce, rl_, 64), dodd = __shfl(co, rl_, 64);  // bins 2*rl_ (even), 2*rl_+1 (odd)
      const u64 mine = (u64)de + (u64)dodd;
      const u64 before = wave_excl_scan_u64(mine, lane);
      const u64 total = __shfl(before + mine, 63, 64);
      const long long r = inner->k[q] - (long long)gt;
      if (r >= 1 && (u64)r <= total) {
        if (before < (u64)r && (u64)r <= before + mine) {
          const int bin = ((u64)r <= before + dodd) ? 2 * rl_ + 1 : 2 * rl_;
          const uint32_t sh = inner->shiftA[q];
          const uint32_t lo_edge = inner->lo[q] + ((uint32_t)bin << sh);
          uint32_t hi_edge = lo_edge + ((1u << sh) - 1u);
          if (hi_edge > inner->hi[q] || hi_edge < lo_edge) hi_edge = inner->hi[q];
          s_edge[q] = upper ? hi_edge : lo_edge;
        }
      } else if (r >= 1 && (u64)r > total && inner->zero_in[q] && (u64)r - total <= zz) {
        if (lane == 0) s_edge[q] = ZERO_KEY;  // the rank sits among the sample's exact zeros
      } else if (lane == 0) {
        s_bad = 1;
      }
    }
  }
  __syncthreads();
  if (tid < nk) {
    const long long kin = kl.k[tid];
    if (kin <= 0) publish_bracket(fs, tid, 0, false, MODE_NONE, KEY_SKIP, KEY_SKIP);
    else if (kin >= (long long)n) publish_bracket(fs, tid, (long long)n, kin > (long long)n, MODE_ALL, KEY_SKIP, KEY_SKIP);
    else {
      uint32_t hi = rl.hi[tid] ? s_edge[2 * tid] : KEY_MAX;
      const uint32_t lo = rl.lo[tid] ? s_edge[2 * tid + 1] : 0u;
      if (hi < lo) hi = lo;  // cannot happen for valid sample brackets; keeps the arithmetic in range
      publish_bracket(fs, tid, kin, false, MODE_GE, lo, hi);
    }
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
// wave instruction covers a contiguous 1 KiB and the mask goes out as one dword per float4.  Per threshold it writes
// the guess mask = [key >= gthr] (final outside the bracket), counts the keys above the bracket, compacts the
// in-bracket candidates (key, flat index) into this workgroup's slab and leaves a 128-bin histogram of them as one u16
// row per (workgroup, threshold): the next kernel sums rows instead of re-reading the candidates.
// The compaction is the expensive part (profiles/r05_topk_experiments.txt: 11.5 us without it, 18.2 with it at N18, one
// threshold; 46 vs 135 us at ten) — vector-ALU instructions, not bytes — so it comes in two forms:
//   * BALLOT (one or two thresholds, or overlapping brackets): slot by slot, the ballot of "lane holds a candidate in
//     slot (u, e)" gives every lane its offset (mbcnt) with no scan, one LDS atomic per wave and chunk reserves the
//     slab range, empty slots are skipped by a scalar branch;
//   * ONE PASS (three or more thresholds with pairwise disjoint brackets — the reference's ten ratios): an element is a
//     candidate of at most one threshold, the threshold loop only records WHICH (one select per element and
//     threshold), and a single pass over the 16 slots appends each candidate to its threshold's slab through a
//     returning LDS atomic: 16 compaction blocks per chunk instead of 16 x nk.
// A real accumulator is not i.i.d. along the flat index: a layer whose magnitudes sit at a threshold puts several times
// the average share of candidates into the workgroups that stream it — what a slab cannot take goes to the
// threshold's shared SPILL row (global atomics, only when a slab is full).  Exact zeros are counted, never compacted;
// the smallest key is tracked for k == n.  `store` = 0 (the two-level route's sample pass): count only, no slabs.
constexpr uint32_t JSEL_NONE = 0xFFu;

template <int NK, bool VO, bool STORE>
__global__ __launch_bounds__(SALUN_BLOCK) void k_main(const float *__restrict__ acc, int64_t n, FastState *fs,
                                                      MaskPtrs mp, uint2 *__restrict__ slabs,
                                                      uint32_t *__restrict__ slab_cnt, uint32_t *__restrict__ wg_gt,
                                                      uint16_t *__restrict__ rows, uint32_t *__restrict__ wg_zero,
                                                      uint32_t *__restrict__ wg_min, uint2 *__restrict__ spill,
                                                      uint32_t cap, uint32_t spill_cap, int nk_real) {
  constexpr bool MULTI = NK > 2;
  constexpr bool store = STORE;
  __shared__ uint32_t s_hist[NK][BINS_A];
  __shared__ uint32_t s_cnt[NK];
  __shared__ uint32_t s_gt[4][NK];
  __shared__ uint32_t s_zero[4], s_min[4];
  __shared__ uint32_t s_lo[NK], s_sh[NK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t nfull = n / CHUNK;
  // the first chunk's loads leave before anything else: they do not depend on the brackets
  vf4 x[4];
  int64_t c = blockIdx.x;
  if (c < nfull) {
#pragma unroll
    for (int u = 0; u < 4; ++u)  // read exactly once: non-temporal
      x[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(acc) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
  }
  for (int i = tid; i < NK * BINS_A; i += SALUN_BLOCK) (&s_hist[0][0])[i] = 0;
  if (tid < NK) s_cnt[tid] = 0;
  {  // this workgroup's share of zeroing the second-level histograms k_resolve accumulates into
    const int w2 = nk_real * HIST2_BINS;
    uint32_t *z2 = &fs->hist2[0][0];
    for (int i = blockIdx.x * SALUN_BLOCK + tid; i < w2; i += gridDim.x * SALUN_BLOCK) z2[i] = 0;
  }
  uint32_t hi[NK], lo[NK], gthr[NK], sh[NK], gtc[NK];
  bool need_zero = false, need_min = false;  // uniform: some bracket reaches the zero key / some k >= n
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    hi[j] = (j < nk_real) ? fs->hi[j] : KEY_SKIP;
    lo[j] = (j < nk_real) ? fs->lo[j] : KEY_SKIP;
    gthr[j] = (j < nk_real) ? fs->gthr[j] : KEY_SKIP;
    sh[j] = (j < nk_real) ? fs->shiftA[j] : 0;
    gtc[j] = 0;  // wave-uniform count of keys above the bracket
    if (j < nk_real) {
      need_zero = need_zero || fs->zero_in[j] != 0;
      need_min = need_min || gthr[j] == 0u;
    }
  }
  bool one_pass = MULTI;  // uniform: every pair of (real) brackets is disjoint
  if (MULTI) {
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      if (tid == 0) { s_lo[j] = lo[j]; s_sh[j] = sh[j]; }
#pragma unroll
      for (int i = 0; i < j; ++i)
        if (lo[j] != KEY_SKIP && lo[i] != KEY_SKIP && !(hi[j] < lo[i] || hi[i] < lo[j])) one_pass = false;
    }
  }
  uint32_t zc = 0, mn = KEY_SKIP;  // per-lane: exact zeros seen, smallest key seen
  uint2 *const slab0 = slabs + (size_t)blockIdx.x * (size_t)nk_real * cap;  // this workgroup's nk_real slabs
  __syncthreads();
  for (; c < nfull; c += gridDim.x) {
    uint32_t k[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      k[u][0] = key_of(x[u].x); k[u][1] = key_of(x[u].y); k[u][2] = key_of(x[u].z); k[u][3] = key_of(x[u].w);
    }
    if (TOPK_EXP & 32) {  // (timing experiment: no software prefetch — load this chunk now)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(acc) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        k[u][0] = key_of(x[u].x); k[u][1] = key_of(x[u].y); k[u][2] = key_of(x[u].z); k[u][3] = key_of(x[u].w);
      }
    }
    const int64_t cn = c + gridDim.x;
    if (cn < nfull && !(TOPK_EXP & 32)) {  // the next chunk's loads fly while this one is classified
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4 *>(acc) + cn * CHUNK_VEC + u * SALUN_BLOCK + tid);
    }
    if (need_zero) {  // exact zeros are only ever consulted when a bracket reaches down to them
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) zc += (uint32_t)(k[u][e] == ZERO_KEY);
    }
    if (need_min) {   // ... and the smallest key only by a k == n threshold
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) mn = k[u][e] < mn ? k[u][e] : mn;
    }
    const uint32_t idx0 = (uint32_t)((c * CHUNK_VEC + tid) << 2);  // flat index of slot (u, e): idx0 + u*1024 + e
    if (MULTI && one_pass) {
      uint32_t jsel[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) jsel[u][e] = JSEL_NONE;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (j >= nk_real) continue;
        const uint32_t w = hi[j] - lo[j];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint32_t bits = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bits |= (uint32_t)(k[u][e] >= gthr[j]) << (8 * e);
            jsel[u][e] = ((k[u][e] - lo[j]) <= w) ? (uint32_t)j : jsel[u][e];
            gtc[j] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(k[u][e] > hi[j]));
          }
          if (!VO && !(TOPK_EXP & 16))
            __builtin_nontemporal_store(bits, reinterpret_cast<uint32_t *>(mp.m[j]) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
        }
      }
      if (!(TOPK_EXP & 8)) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t js = jsel[u][e];
            if (js != JSEL_NONE) {
              const uint32_t key = k[u][e];
              const uint32_t pos = atomicAdd(&s_cnt[js], 1u);
              atomicAdd(&(&s_hist[0][0])[js * BINS_A + ((key - s_lo[js]) >> s_sh[js])], 1u);
              if (store) {
                const uint2 ent = make_uint2(key, idx0 + (uint32_t)(u * 1024 + e));
                if (pos < cap) {
                  *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(slab0) + (size_t)((js * cap + pos) << 3)) = ent;
                } else {
                  const uint32_t sp_pos = atomicAdd(&fs->spill_cnt[js], 1u);
                  if (sp_pos < spill_cap) spill[(size_t)js * spill_cap + sp_pos] = ent;
                }
              }
            }
          }
      }
      continue;
    }
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
          bits |= (uint32_t)(k[u][e] >= gthr[j]) << (8 * e);
          bal[u * 4 + e] = __builtin_amdgcn_ballot_w64((k[u][e] - lo[j]) <= w);
          total += (uint32_t)__builtin_popcountll(bal[u * 4 + e]);
          gtc[j] += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(k[u][e] > hi[j]));
        }
        if (!VO && !(TOPK_EXP & 16))
          __builtin_nontemporal_store(bits, reinterpret_cast<uint32_t *>(mp.m[j]) + c * CHUNK_VEC + u * SALUN_BLOCK + tid);
      }
      if (total && !(TOPK_EXP & 8)) {  // wave-uniform
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_cnt[j], total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        char *const slab_b = reinterpret_cast<char *>(slab0 + (size_t)j * cap);
        if (!STORE || base + total <= cap) {
          // ---- the common case: the whole reservation fits the slab
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned long long b = bal[u * 4 + e];
              if (b) {  // scalar branch: most slots hold no candidate when the bracket is narrow
                const uint32_t d = k[u][e] - lo[j];
                if (d <= w) {  // (the lane's own predicate again: cheaper than shifting the 64-bit ballot)
                  atomicAdd(&s_hist[j][d >> sh[j]], 1u);
                  if (STORE) {
                    const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                    *reinterpret_cast<uint2 *>(slab_b + (size_t)(pos << 3)) =
                        make_uint2(k[u][e], idx0 + (uint32_t)(u * 1024 + e));
                  }
                }
                base += (uint32_t)__builtin_popcountll(b);
              }
            }
        } else {
          // ---- rare: positions [base, base + total) run past `cap`; those go to the shared spill row
          const uint32_t first_over = base > cap ? base : cap;
          uint32_t sbase = 0;
          if (lane == 0) sbase = atomicAdd(&fs->spill_cnt[j], base + total - first_over);
          sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase);
          for (int se = 0; se < 16; ++se) {  // (not unrolled: cold code)
            const unsigned long long b = bal[se];
            if (!b) continue;
            const uint32_t key = k[se >> 2][se & 3];
            const uint32_t d = key - lo[j];
            if (d <= w) {
              const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
              atomicAdd(&s_hist[j][d >> sh[j]], 1u);
              const uint2 ent = make_uint2(key, idx0 + (uint32_t)((se >> 2) * 1024 + (se & 3)));
              if (pos < cap) *reinterpret_cast<uint2 *>(slab_b + (size_t)(pos << 3)) = ent;
              else {
                const uint32_t sp_pos = sbase + (pos - first_over);
                if (sp_pos < spill_cap) spill[(size_t)j * spill_cap + sp_pos] = ent;
              }
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
      zc += (uint32_t)(key == ZERO_KEY);
      mn = key < mn ? key : mn;
#pragma unroll
      for (int j = 0; j < NK; ++j) {
        if (j >= nk_real) continue;
        tail_gt[j] += key > hi[j];
        if (!VO) mp.m[j][i] = (uint8_t)(key >= gthr[j]);
        if ((key - lo[j]) <= (hi[j] - lo[j])) {
          const uint32_t pos = atomicAdd(&s_cnt[j], 1u);
          atomicAdd(&s_hist[j][(key - lo[j]) >> sh[j]], 1u);
          if (store) {
            if (pos < cap) slab0[(size_t)j * cap + pos] = make_uint2(key, (uint32_t)i);
            else {
              const uint32_t sp_pos = atomicAdd(&fs->spill_cnt[j], 1u);
              if (sp_pos < spill_cap) spill[(size_t)j * spill_cap + sp_pos] = make_uint2(key, (uint32_t)i);
            }
          }
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
  for (int off = 32; off > 0; off >>= 1) {
    zc += __shfl_down(zc, off, 64);
    const uint32_t o = __shfl_down(mn, off, 64);
    mn = o < mn ? o : mn;
  }
  if (lane == 0) { s_zero[wave] = zc; s_min[wave] = mn; }
  __syncthreads();
  {  // this workgroup's histogram rows, two bins per dword
    uint32_t *out = reinterpret_cast<uint32_t *>(rows) + (size_t)blockIdx.x * (size_t)nk_real * (BINS_A / 2);
    const uint32_t *h = &s_hist[0][0];
    for (int i = tid; i < nk_real * (BINS_A / 2); i += SALUN_BLOCK) {
      const uint32_t a = h[2 * i], b = h[2 * i + 1];
      if ((a | b) > 0xFFFFu) fs->fail = 1;  // a u16 row counter would wrap (> 65535 candidates of one workgroup in one bin)
      out[i] = (a & 0xFFFFu) | (b << 16);
    }
  }
  if (tid < nk_real) {
    wg_gt[(size_t)blockIdx.x * nk_real + tid] = s_gt[0][tid] + s_gt[1][tid] + s_gt[2][tid] + s_gt[3][tid];
    const uint32_t mine = s_cnt[tid];
    slab_cnt[(size_t)blockIdx.x * nk_real + tid] = mine < cap ? mine : cap;
  }
  if (tid == 0) {
    wg_zero[blockIdx.x] = s_zero[0] + s_zero[1] + s_zero[2] + s_zero[3];
    uint32_t m = s_min[0];
    for (int w = 1; w < 4; ++w) m = s_min[w] < m ? s_min[w] : m;
    wg_min[blockIdx.x] = m;
  }
}

// ------------------------------------------------------------------------------------ k_resolve
// nseg workgroups of 1024 threads per threshold.  Every workgroup sums k_main's rows for its threshold (the bracket's
// 128-bin histogram, the count above the bracket, the exact zeros: 256 KB of L2 reads) and repeats the tiny first-level
// selection; then its waves walk the slab rows: candidates above the chosen bin are selected for good, candidates
// inside it are counted in a 4096-bin histogram of that bin (LDS per workgroup, then global atomics spread over 4096
// addresses) and appended to this workgroup's private segment of the threshold's short list, so nothing is appended
// through a shared counter.  The kernel is a chain of memory round trips, so everything it reads is requested up
// front — the state, the sixteen row pieces of each thread, and, speculatively, the first slab entries of each wave
// (a slab row is allocated to `cap` whatever its count): one round trip, then arithmetic.
// A rank that runs past every candidate into the block of exact zeros makes the threshold the ZERO key: every
// candidate is selected here and the tie pass behind k_finish admits the first `zt_budget` zeros by flat index.
template <bool VO>
__global__ __launch_bounds__(1024) void k_resolve(FastState *fs, const uint2 *__restrict__ slabs,
                                                  const uint32_t *__restrict__ slab_cnt,
                                                  const uint32_t *__restrict__ wg_gt, const uint16_t *__restrict__ rows,
                                                  const uint32_t *__restrict__ wg_zero, const uint2 *__restrict__ spill,
                                                  uint32_t cap, uint32_t spill_cap, int main_grid, int nk, int nseg,
                                                  uint32_t seg_cap, uint2 *__restrict__ list2 /*[nk][nseg][seg_cap]*/,
                                                  uint32_t *__restrict__ seg_cnt /*[nk][nseg]*/, MaskPtrs mp) {
  __shared__ uint32_t s_part[16][BINS_A];
  __shared__ uint32_t s_hist[BINS_A];
  __shared__ u64 s_gtw[16], s_zw[16];
  __shared__ uint32_t s_lo2, s_hi2, s_sh2, s_ok, s_n;
  __shared__ uint32_t s_h2[HIST2_BINS];  // this workgroup's share of the 4096-bin histogram: a run of equal keys
                                         // would otherwise serialise ~12 ns global atomics on ONE address
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = blockIdx.x % nk, g = blockIdx.x / nk;  // workgroup g of the nseg that serve threshold j
  // ---- one round trip: state, row pieces, first slab entries
  const uint32_t failed = fs->fail, mode = fs->mode[j];
  const long long kj = fs->k[j];
  const uint32_t lo_j = fs->lo[j], hi_j = fs->hi[j], sh_j = fs->shiftA[j], mid = fs->mid[j], zero_in = fs->zero_in[j];
  const uint32_t spill_n = fs->spill_cnt[j];
  const int cidx = tid & 15, rsub = tid >> 4;  // 16 threads per row (16 B each), 64 rows per sweep, <= 16 sweeps
  uint4 rv[16];
  {
    const uint4 *base = reinterpret_cast<const uint4 *>(rows);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = rsub + 64 * i;
      rv[i] = (b < main_grid) ? base[((size_t)b * nk + j) * (BINS_A / 8) + cidx] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  u64 gt = 0, zz = 0;
  for (int b = tid; b < main_grid; b += 1024) { gt += wg_gt[(size_t)b * nk + j]; zz += wg_zero[b]; }
  const int waves = nseg * 16;
  const int spr = waves > main_grid ? waves / main_grid : 1;  // waves sharing one slab row
  const uint32_t step = 64u * (uint32_t)spr;
  // wave id over this threshold's workgroups, interleaved: consecutive slab rows (a layer whose magnitudes sit at the
  // threshold fills a RUN of rows) go to different workgroups, hence to different short-list segments
  const int w0 = wave * nseg + g;
  const bool have0 = w0 < main_grid * spr;
  uint32_t cnt0 = 0, i00 = 0;
  uint2 e0[4];
  if (have0) {
    const int b = w0 / spr, part = w0 % spr;
    const size_t r = (size_t)b * nk + j;
    cnt0 = slab_cnt[r];
    const uint2 *slab = slabs + r * cap;
    i00 = (uint32_t)part * 64u + (uint32_t)lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t i = i00 + (uint32_t)q * step;
      e0[q] = (i < cap) ? slab[i] : make_uint2(KEY_SKIP, 0u);  // beyond the count: stale bytes, masked below
    }
  }
  {  // ---- sum the rows (before any exit test: every load above is consumed or pinned here, so the compiler keeps
     // them in ONE batch ahead of the first wait instead of sinking them behind a branch on `failed`)
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a[0] += rv[i].x & 0xFFFFu; a[1] += rv[i].x >> 16; a[2] += rv[i].y & 0xFFFFu; a[3] += rv[i].y >> 16;
      a[4] += rv[i].z & 0xFFFFu; a[5] += rv[i].z >> 16; a[6] += rv[i].w & 0xFFFFu; a[7] += rv[i].w >> 16;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[i] += __shfl_xor(a[i], 16, 64);
      a[i] += __shfl_xor(a[i], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s_part[wave][cidx * 8 + i] = a[i];
    }
    gt = salun_wave_sum_u64(gt);
    zz = salun_wave_sum_u64(zz);
    if (lane == 0) { s_gtw[wave] = gt; s_zw[wave] = zz; }
    asm volatile("" ::"v"(cnt0), "v"(e0[0].x), "v"(e0[1].x), "v"(e0[2].x), "v"(e0[3].x), "v"(e0[0].y), "v"(e0[1].y),
                 "v"(e0[2].y), "v"(e0[3].y));
  }
  if (failed) return;  // already decided (a u16 row counter wrapped in k_main): the full scan redoes the job
  if (mode != MODE_GE) return;  // nothing / everything selected: no candidates, k_finish publishes
  if (TOPK_EXP & 1) return;  // (timing experiment: loads + row sums only)
  if (tid == 0) { s_n = 0; s_ok = 0; }
  for (int i = tid; i < HIST2_BINS; i += 1024) s_h2[i] = 0;
  __syncthreads();
  if (tid < BINS_A) {
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += s_part[w][tid];
    s_hist[tid] = t;
  }
  __syncthreads();
  if (wave == 0) {  // ---- first-level selection: 128 bins from the top, lane l owns bins 127-2l and 126-2l
    u64 cgt = 0, zeros = 0;
    for (int w = 0; w < 16; ++w) { cgt += s_gtw[w]; zeros += s_zw[w]; }
    const long long r = kj - (long long)cgt;  // rank wanted among the candidates, 1-based descending
    const int bA = BINS_A - 1 - 2 * lane, bB = bA - 1;
    const u64 cA = s_hist[bA], cB = s_hist[bB];
    const u64 before = wave_excl_scan_u64(cA + cB, lane);
    const u64 total = __shfl(before + cA + cB, 63, 64);
    if (spill_n > spill_cap) {
      if (g == 0 && lane == 0) fs->fail = 1;  // more spilled candidates than the spill row holds (heavy ties)
    } else if (r >= 1 && (u64)r <= total) {
      if (before < (u64)r && (u64)r <= before + cA + cB) {
        const bool first = (u64)r <= before + cA;
        const int bin = first ? bA : bB;
        const u64 cum = first ? before : before + cA;
        const uint32_t lo2 = lo_j + ((uint32_t)bin << sh_j);
        uint32_t hi2 = lo2 + ((1u << sh_j) - 1u);
        if (hi2 > hi_j || hi2 < lo2) hi2 = hi_j;
        s_lo2 = lo2; s_hi2 = hi2; s_sh2 = sh_j > 12 ? sh_j - 12 : 0; s_ok = 1;
        if (g == 0) {
          fs->lo2[j] = lo2; fs->hi2[j] = hi2; fs->shift2[j] = sh_j > 12 ? sh_j - 12 : 0;
          fs->r2[j] = (u64)r - cum;
        }
      }
    } else if (r >= 1 && zero_in && (u64)r - total <= zeros) {
      if (lane == 0) {  // the k-th largest is an exact zero: every candidate is above it
        s_lo2 = ZERO_KEY; s_hi2 = ZERO_KEY; s_sh2 = 0; s_ok = 1;
        if (g == 0) {
          fs->zt_budget[j] = (u64)r - total;
          __hip_atomic_store(&fs->zt_any, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else if (g == 0 && lane == 0) {
      fs->fail = 1;  // the bracket missed the threshold (or it lies among NaNs)
    }
  }
  __syncthreads();
  uint2 *seg = list2 + ((size_t)j * nseg + g) * seg_cap;
  if (TOPK_EXP & 2) return;  // (timing experiment: ... + selection)
  if (s_ok) {
    const uint32_t lo2 = s_lo2, hi2 = s_hi2, sh2 = s_sh2;
    uint8_t *mask = mp.m[j];
    auto classify = [&](const uint2 e) {
      if (e.x > hi2) { if (!VO && e.x <= mid) mask[e.y] = 1; }       // selected for good; k_main guessed 0
      else if (e.x < lo2) { if (!VO && e.x > mid) mask[e.y] = 0; }   // rejected for good; k_main guessed 1
      else {
        atomicAdd(&s_h2[(e.x - lo2) >> sh2], 1u);
        const uint32_t p = atomicAdd(&s_n, 1u);
        if (p < seg_cap) seg[p] = e;
      }
    };
    if (have0) {  // the first row of this wave: its first four entries per lane are already here
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (i00 + (uint32_t)q * step < cnt0) classify(e0[q]);
      const int b = w0 / spr;
      const uint2 *slab = slabs + ((size_t)b * nk + j) * cap;
      for (uint32_t i0 = i00 + 4u * step; i0 < cnt0; i0 += 4u * step) {
        uint2 e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t i = i0 + (uint32_t)q * step;
          e[q] = (i < cnt0) ? slab[i] : make_uint2(KEY_SKIP, 0u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (e[q].x != KEY_SKIP) classify(e[q]);
      }
    }
    for (int w = w0 + waves; w < main_grid * spr; w += waves) {
      const int b = w / spr, part = w % spr;
      const size_t r = (size_t)b * nk + j;
      const uint32_t cnt = slab_cnt[r];
      const uint2 *slab = slabs + r * cap;
      for (uint32_t i0 = (uint32_t)part * 64u + (uint32_t)lane; i0 < cnt; i0 += 4u * step) {
        uint2 e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t i = i0 + (uint32_t)q * step;
          e[q] = (i < cnt) ? slab[i] : make_uint2(KEY_SKIP, 0u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (e[q].x != KEY_SKIP) classify(e[q]);
      }
    }
    // the shared spill row of this threshold, dealt over every wave of its workgroups
    const uint2 *sp = spill + (size_t)j * spill_cap;
    for (uint32_t i = (uint32_t)w0 * 64u + (uint32_t)lane; i < spill_n; i += (uint32_t)waves * 64u) classify(sp[i]);
  }
  __syncthreads();
  const uint32_t m = s_n;
  if (m && !(TOPK_EXP & 4))
    for (int i = tid; i < HIST2_BINS; i += 1024)
      if (s_h2[i]) atomicAdd(&fs->hist2[j][i], s_h2[i]);
  if (tid == 0) {
    seg_cnt[(size_t)j * nseg + g] = m < seg_cap ? m : seg_cap;
    if (m > seg_cap) fs->fail = 1;  // more residents than this workgroup's segment holds (heavy ties)
  }
}

// ------------------------------------------------------------------------------------- k_finish
// One workgroup (1024 threads) per threshold.
constexpr int MAX_SEGS = 1024;  // k_resolve grid bound

// entry #f of the concatenated segments: segment by binary search over the exclusive offsets
__device__ __forceinline__ const uint2 *list_entry_ptr(const uint2 *__restrict__ lst, const uint32_t *s_off, int nseg,
                                                       uint32_t seg_cap, uint32_t f) {
  int a = 0, b = nseg - 1;
  while (a < b) {
    const int mid = (a + b + 1) >> 1;
    if (s_off[mid] <= f) a = mid; else b = mid - 1;
  }
  return lst + (size_t)a * seg_cap + (f - s_off[a]);
}

// F workgroups of 1024 threads per threshold (F = 1 for short lists).  Each repeats the two small scans (segment
// offsets, second-level histogram -> final bin), walks ITS share of the short-list segments — entries outside the final
// bin are final and only need their mask byte when k_main's guess was wrong — and appends the few residents of the final
// bin to one global list with write-through (sc1) stores; the workgroup that takes the last ticket ranks them exactly
// (key descending, flat index ascending) and publishes tau_j.  Hand-off form (MI355X_MICROARCH.md, "valid forms"): sc1
// payload -> s_waitcnt vmcnt(0) -> device-scope ticket; the last arriver reads the payload with sc1 loads — no fence,
// so no write-back of the masks k_main left dirty in the L2s.
template <bool VO>
__global__ __launch_bounds__(1024) void k_finish(FastState *fs, TopkPub *pub, const uint2 *__restrict__ list2,
                                                 const uint32_t *__restrict__ seg_cnt, int nseg, uint32_t seg_cap,
                                                 const uint32_t *__restrict__ wg_min, int main_grid, u64 *fin_list, int nk,
                                                 MaskPtrs mp) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_off[MAX_SEGS], s_cl[MAX_SEGS];
  __shared__ uint32_t s_key[FINAL_CAP], s_idx[FINAL_CAP];
  __shared__ uint32_t s_b2, s_r3, s_c3, s_found, s_tau, s_hist[2048], s_sel, s_rem, s_last, s_n, s_base;
  const int j = blockIdx.x % nk, f = blockIdx.x / nk, F = gridDim.x / nk;
  const int tid = threadIdx.x;
  // ---- one round trip for everything the prologue needs
  const uint32_t failed = fs->fail, mode = fs->mode[j], was_all = fs->was_all[j];
  const u64 ztb = fs->zt_budget[j], r2 = fs->r2[j];
  const uint32_t lo2 = fs->lo2[j], hi2 = fs->hi2[j], sh2 = fs->shift2[j], mid = fs->mid[j];  // k_main wrote [key > mid]
  const uint32_t segc = (tid < nseg) ? seg_cnt[(size_t)j * nseg + tid] : 0;
  uint32_t hc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) hc[i] = fs->hist2[j][HIST2_BINS - 1 - (tid * 4 + i)];
  uint32_t wmin = KEY_SKIP;
  for (int b = tid; b < main_grid; b += 1024) { const uint32_t v = wg_min[b]; wmin = v < wmin ? v : wmin; }
  if (failed) return;  // the full scan publishes
  if (mode == MODE_NONE) {
    if (tid == 0 && f == 0) { pub->mode[j] = MODE_NONE; pub->tau[j] = 0; pub->route = 1; }
    return;
  }
  if (mode == MODE_ALL) {  // k >= n: every element is selected; for k == n the threshold is the smallest key
    if (f != 0) return;
    uint32_t m = wmin;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(m, off, 64); m = o < m ? o : m; }
    if ((tid & 63) == 0) s_w[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 16; ++w) m = s_w[w] < m ? s_w[w] : m;
      pub->mode[j] = was_all ? MODE_ALL : MODE_GE;
      pub->tau[j] = m;
      pub->route = 1;
    }
    return;
  }
  if (ztb != 0) {  // the threshold is the zero key; the tie pass writes the admitted zeros
    if (tid == 0 && f == 0) { pub->mode[j] = MODE_GE; pub->tau[j] = ZERO_KEY; pub->route = 1; }
    return;
  }
  if (TOPK_EXP & 64) return;  // (timing experiment: prologue loads only)
  if (tid == 0) { s_found = 0; s_tau = 0; s_last = 0; s_n = 0; s_base = 0; }
  uint32_t n2;
  {  // offsets of the per-workgroup segments of the short list
    const uint32_t ex = block1024_excl_scan(segc, s_w, &n2);
    if (tid < nseg) { s_off[tid] = ex; s_cl[tid] = segc; }
  }
  __syncthreads();
  // ---- final bin: 4096 bins walked from the top, thread t owns bins 4095-4t .. 4095-4t-3
  {
    const uint32_t mine = hc[0] + hc[1] + hc[2] + hc[3];
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
        if (!s_found && r2 <= (u64)cum + hc[i]) {
          s_b2 = HIST2_BINS - 1 - (tid * 4 + i);
          s_r3 = (uint32_t)(r2 - cum);
          s_c3 = hc[i];
          s_found = 1;
        }
        cum += hc[i];
      }
    }
  }
  __syncthreads();
  if (TOPK_EXP & 128) return;  // (timing experiment: ... + the two scans)
  uint32_t lo3 = lo2 + (s_b2 << sh2);
  uint32_t hi3 = lo3 + ((1u << sh2) - 1u);
  if (hi3 > hi2 || hi3 < lo3) hi3 = hi2;
  uint32_t r3 = s_r3;
  const uint32_t c3 = s_c3;
  const uint2 *lst = list2 + (size_t)j * nseg * seg_cap;
  uint8_t *mask = mp.m[j];
  if (c3 <= (uint32_t)FINAL_CAP) {
    // one wave per segment, eight segments (two entries per lane each) in flight per wave: coalesced loads, no search
    // for the segment of an entry; workgroup f of F takes segments f, f + F, f + 2F, ...
    u64 *fin = fin_list + (size_t)j * FINAL_CAP;
    auto settle = [&](const uint2 e) {
      if (e.x > hi3) { if (!VO && e.x <= mid) mask[e.y] = 1; }
      else if (e.x < lo3) { if (!VO && e.x > mid) mask[e.y] = 0; }
      else {
        const uint32_t p = atomicAdd(&s_n, 1u);  // this workgroup's residents of the final bin (a handful)
        if (p < (uint32_t)FINAL_CAP) { s_key[p] = e.x; s_idx[p] = e.y; }
      }
    };
    const int wave = tid >> 6, lane = tid & 63;
    const int nmine = (nseg - f + F - 1) / F;  // segments of this workgroup: f + F * i, i < nmine
    for (int i0 = wave; i0 < nmine; i0 += 128) {
      uint2 e[8][2];
      uint32_t cn[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = i0 + 16 * q;
        const int sg = f + F * i;
        cn[q] = (i < nmine) ? s_cl[sg] : 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const uint32_t li = (uint32_t)lane + 64u * (uint32_t)h;
          e[q][h] = (li < cn[q]) ? lst[(size_t)sg * seg_cap + li] : make_uint2(KEY_SKIP, 0u);
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (e[q][h].x != KEY_SKIP) settle(e[q][h]);
        for (uint32_t li = 128u + (uint32_t)lane; li < cn[q]; li += 64u)
          settle(lst[(size_t)(f + F * (i0 + 16 * q)) * seg_cap + li]);
      }
    }
    __syncthreads();
    uint32_t L = s_n;
    if (F > 1) {
      // ---- hand this workgroup's residents to whichever workgroup of the threshold finishes last
      if (L > (uint32_t)FINAL_CAP) L = FINAL_CAP;  // (cannot happen: c3 <= FINAL_CAP is the total)
      if (tid == 0) s_base = L ? atomicAdd(&fs->fin_n[j], L) : 0u;
      __syncthreads();
      const uint32_t base = s_base;
      for (uint32_t p = tid; p < L; p += 1024)
        if (base + p < (uint32_t)FINAL_CAP)
          __hip_atomic_store(&fin[base + p], ((u64)s_idx[p] << 32) | (u64)s_key[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores have left
      __syncthreads();
      if (tid == 0) {
        // acquire-release at agent scope: the write-through stores + drained vmcnt above and the sc1 loads
        // below are the hand-off that works on gfx950 (cdna_hip_programming.md G16, form R1); the ordering on the
        // ticket is what the HIP memory model needs to say the same thing — ~2 us on a call of 80 us or more
        const uint32_t t = __hip_atomic_fetch_add(&fs->fin_ticket[j], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (uint32_t)(F - 1)) ? 1u : 0u;
      }
      __syncthreads();
      if (!s_last) return;
      // ---- the last workgroup of this threshold: every resident of the final bin is in the list
      L = __hip_atomic_load(&fs->fin_n[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (L != c3) { if (tid == 0) fs->fail = 1; return; }
      for (uint32_t p = tid; p < L; p += 1024) {
        const u64 v = __hip_atomic_load(&fin[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_key[p] = (uint32_t)v;
        s_idx[p] = (uint32_t)(v >> 32);
      }
    } else if (L != c3) {
      if (tid == 0) fs->fail = 1;
      return;
    }
    __syncthreads();
    for (uint32_t p = tid; p < L; p += 1024) {
      const uint32_t kp = s_key[p], ip = s_idx[p];
      uint32_t rank = 0;
      for (uint32_t q = 0; q < L; ++q) rank += (s_key[q] > kp) || (s_key[q] == kp && s_idx[q] < ip);
      if (!VO && (rank < r3) != (kp > mid)) mask[ip] = (uint8_t)(rank < r3);
      if (rank == r3 - 1) s_tau = kp;
    }
    __syncthreads();
    if (tid == 0) { pub->mode[j] = MODE_GE; pub->tau[j] = s_tau; pub->route = 1; }
    return;
  }
  if (f != 0) return;  // a crowded final bin (ties): one workgroup walks the whole list below
  if (sh2 != 0) {
    // ---- a crowded final bin that still spans 2^sh2 keys (a run of ties inside it): one more histogram, one counter
    // per key, narrows it to the single key holding the rank
    if (sh2 > 11) { if (tid == 0) fs->fail = 1; return; }  // (wider than the counters below: never for a real bracket)
    for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t ff = tid; ff < n2; ff += 1024) {
      const uint2 e = *list_entry_ptr(lst, s_off, nseg, seg_cap, ff);
      if (e.x >= lo3 && e.x <= hi3) atomicAdd(&s_hist[e.x - lo3], 1u);
    }
    __syncthreads();
    const uint32_t a = s_hist[2047 - 2 * tid], b = s_hist[2046 - 2 * tid];  // keys from the top (bins past the width are empty)
    uint32_t total;
    const uint32_t before = block1024_excl_scan(a + b, s_w, &total);
    if (before < r3 && r3 <= before + a + b) {
      if (r3 <= before + a) { s_sel = 2047 - 2 * tid; s_rem = r3 - before; }
      else { s_sel = 2046 - 2 * tid; s_rem = r3 - before - a; }
    }
    __syncthreads();
    lo3 = hi3 = lo3 + s_sel;
    r3 = s_rem;
    __syncthreads();
  }
  // ---- entries tie at the single key lo3: admit the r3 lowest flat indices.  Radix select on the index
  // (11 + 11 + 10 bits, most significant first, ascending) over the short list.
  uint32_t prefix = 0, want = r3;
  for (int level = 0; level < 3; ++level) {
    const int shift = (level == 0) ? 21 : (level == 1) ? 10 : 0;
    const int nb = (level == 2) ? 1024 : 2048;
    for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t ff = tid; ff < n2; ff += 1024) {
      const uint2 e = *list_entry_ptr(lst, s_off, nseg, seg_cap, ff);
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
    for (uint32_t ff = tid; ff < n2; ff += 1024) {
      const uint2 e = *list_entry_ptr(lst, s_off, nseg, seg_cap, ff);
      const bool fin = e.x > hi3 || (e.x == lo3 && e.y <= idx_star);
      if (fin != (e.x > mid)) mask[e.y] = (uint8_t)fin;
    }
  if (tid == 0) { pub->mode[j] = MODE_GE; pub->tau[j] = lo3; pub->route = 1; }
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

// In-place exclusive scan of one row of per-chunk counts by ONE workgroup (256 threads).
__device__ __forceinline__ void scan_row_inplace(u64 *row, int64_t nchunk) {
  __shared__ u64 s_wave[4];
  __shared__ u64 s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
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

// The tie pass of the fast route (launched behind k_finish inside k_fullscan's grid): thresholds whose k-th largest
// element is an exact zero have every non-zero candidate selected already; here the zeros are numbered in flat-index
// order (per-chunk populations, one scan, one pass over the chunks that still hold budget) and the first zt_budget[j]
// of them get their mask byte.  One row of counts serves every such threshold: they all tie at the same key.
__device__ __forceinline__ void zero_tie_pass(const float *__restrict__ acc, int64_t n, int nk, const MaskPtrs &mp,
                                              TopkPub *pub, FullState *full, const FastState *fs, u64 *tie, bool aligned,
                                              uint32_t *lds4, uint32_t *s_epoch) {
  __shared__ u64 s_budget[MAXK];
  __shared__ u64 s_maxb;
  const int tid = threadIdx.x;
  const int64_t nvec = (n + 3) >> 2;
  const int64_t nchunk = (nvec + CHUNK_VEC - 1) / CHUNK_VEC;
  if (tid == 0) {
    u64 mx = 0;
    for (int j = 0; j < MAXK; ++j) {
      const u64 b = (j < nk) ? fs->zt_budget[j] : 0;
      s_budget[j] = b;
      mx = b > mx ? b : mx;
    }
    s_maxb = mx;
  }
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    uint32_t k[4][4];
    load_chunk(acc, c, nvec, n, aligned, k);
    uint32_t cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) cnt += (k[u][e] == ZERO_KEY);
    const uint32_t tot = block_sum_u32(cnt, lds4);
    if (tid == 0) tie[c] = tot;
  }
  if (!grid_barrier(&full->bar, s_epoch, &pub->error)) return;
  if (blockIdx.x == 0) scan_row_inplace(tie, nchunk);
  if (!grid_barrier(&full->bar, s_epoch, &pub->error)) return;
  const u64 maxb = s_maxb;
  for (int64_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
    u64 before = ld_agent_u64(&tie[c]);  // zeros in earlier chunks
    if (before >= maxb) continue;        // every budget is spent before this chunk (workgroup-uniform)
    uint32_t k[4][4];
    load_chunk(acc, c, nvec, n, aligned, k);
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // in-chunk index order: sub-vector, lane, element
      uint32_t cnt = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) cnt += (k[u][e] == ZERO_KEY);
      uint32_t total;
      u64 pos = before + block_excl_scan_u32(cnt, lds4, &total);
      const int64_t i0 = (c * CHUNK_VEC + u * SALUN_BLOCK + tid) << 2;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k[u][e] == ZERO_KEY) {
          for (int j = 0; j < nk; ++j)
            if (pos < s_budget[j]) mp.m[j][i0 + e] = 1;
          ++pos;
        }
      before += total;
    }
  }
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
  const int tid = threadIdx.x;
  const int nk = kl.nk;
  if (!always) {
    const uint32_t failed = ld_agent_u32(&fs->fail);
    if (!failed) {
      if (!ld_agent_u32(&fs->zt_any) || values_only) return;  // the fast route published everything
      if (tid == 0) s_epoch = 0;
      __syncthreads();
      zero_tie_pass(acc, n, nk, mp, pub, full, fs, tie, aligned != 0, lds4, &s_epoch);
      return;
    }
  }
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
      scan_row_inplace(tie + (int64_t)j * nchunk, nchunk);
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
  int gr;            // k_resolve workgroups per threshold = short-list segments per threshold
  uint32_t cap;      // slab entries per (workgroup, threshold)
  uint32_t spill_cap;  // entries of one threshold's shared spill row
  uint32_t seg_cap;  // short-list entries per k_resolve workgroup
  int gf;            // k_finish workgroups per threshold
  size_t off_fs, off_keys, off_cnt, off_gt, off_rows, off_zero, off_min, off_slabs, off_spill, off_list2, off_seg, off_fin, bytes;
};
inline FastLayout fast_layout(int64_t n, int nk, double f) {
  FastLayout L;
  L.grid = main_grid_for(n);
  const double cands = f * (double)n;  // per threshold
  const double per_slab = cands / (double)L.grid;
  // a slab takes twice the mean share; what a workgroup meets beyond that (a layer whose magnitudes sit at the
  // threshold: a real accumulator is not i.i.d. along the flat index) goes to the threshold's shared spill row
  L.cap = (uint32_t)(2.0 * per_slab) + 256u;
  L.spill_cap = (uint32_t)(0.5 * cands) + 131072u;
  const int rows = L.grid * nk;
  // k_resolve workgroups per threshold (= short-list segments per threshold): enough waves that a lane meets at most
  // four slab entries (one unrolled round of loads), at most 512 workgroups (2 per CU) in all
  int gpj = (int)(cands / 4096.0) + 1;
  if (gpj > 512 / nk) gpj = 512 / nk;
  if (gpj < 1) gpj = 1;
  L.gr = gpj;
  // residents of the chosen first-level bin, spread over the segments.  Their mean is cands / BINS_A / gpj, but a layer
  // whose magnitudes sit at the threshold can put a large share of ALL candidates into that one bin: room for half
  // of them (k_finish then walks a long list — slower, never wrong)
  L.seg_cap = (uint32_t)(0.5 * cands / (double)gpj) + 1024u;
  size_t b = 0;
  L.off_fs = b;    b += align256(sizeof(FastState));
  L.off_keys = b;  b += align256(sizeof(uint32_t) * (size_t)SAMPLE_MAX);
  L.off_cnt = b;   b += align256(sizeof(uint32_t) * (size_t)rows);
  L.off_gt = b;    b += align256(sizeof(uint32_t) * (size_t)rows);
  L.off_rows = b;  b += align256(sizeof(uint16_t) * (size_t)rows * (size_t)BINS_A);
  L.off_zero = b;  b += align256(sizeof(uint32_t) * (size_t)L.grid);
  L.off_min = b;   b += align256(sizeof(uint32_t) * (size_t)L.grid);
  L.off_slabs = b; b += align256(sizeof(uint2) * (size_t)rows * (size_t)L.cap);
  L.off_spill = b; b += align256(sizeof(uint2) * (size_t)nk * (size_t)L.spill_cap);
  L.off_list2 = b; b += align256(sizeof(uint2) * (size_t)nk * (size_t)L.gr * (size_t)L.seg_cap);
  L.off_seg = b;   b += align256(sizeof(uint32_t) * (size_t)nk * (size_t)L.gr);
  L.off_fin = b;   b += align256(sizeof(u64) * (size_t)nk * (size_t)FINAL_CAP);
  L.gf = (gpj + 63) / 64;  // one k_finish workgroup per 64 segments, at most 8 (and 128 in all)
  if (L.gf > 8) L.gf = 8;
  if (L.gf > 128 / nk) L.gf = 128 / nk;
  if (L.gf < 1) L.gf = 1;
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
unsigned long long g_fullscan_attr = 0;  // one bit per device: the dynamic-LDS opt-in is per device
inline int fullscan_grid(int64_t n) {
  if (salun_once_needed(&g_fullscan_attr)) {
    // 2 KiB + 16 x 4 KiB of dynamic LDS at the maximum threshold count: above the 64 KiB default
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_fullscan), hipFuncAttributeMaxDynamicSharedMemorySize,
                              96 * 1024);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
      g_cu_count = cus;   // (devices of one node are the same part)
    else if (g_cu_count == 0)
      g_cu_count = 64;
    salun_once_mark(&g_fullscan_attr);
  }
  int64_t g = 2 * (int64_t)g_cu_count;  // resident with room to spare: the grid barrier needs every workgroup running
  if (g > 512) g = 512;
  const int64_t nc = chunks_of(n);
  if (g > nc) g = nc < 1 ? 1 : nc;
  return (int)g;
}

// Typed views of one fast-route instance inside the workspace.
struct FastPtrs {
  FastState *fs;
  uint32_t *keys, *slab_cnt, *wg_gt, *wg_zero, *wg_min, *seg_cnt;
  uint16_t *rows;
  uint2 *slabs, *spill, *list2;
  u64 *fin;
};
inline FastPtrs fast_ptrs(char *base, const FastLayout &L) {
  FastPtrs P;
  P.fs = reinterpret_cast<FastState *>(base + L.off_fs);
  P.keys = reinterpret_cast<uint32_t *>(base + L.off_keys);
  P.slab_cnt = reinterpret_cast<uint32_t *>(base + L.off_cnt);
  P.wg_gt = reinterpret_cast<uint32_t *>(base + L.off_gt);
  P.rows = reinterpret_cast<uint16_t *>(base + L.off_rows);
  P.wg_zero = reinterpret_cast<uint32_t *>(base + L.off_zero);
  P.wg_min = reinterpret_cast<uint32_t *>(base + L.off_min);
  P.slabs = reinterpret_cast<uint2 *>(base + L.off_slabs);
  P.spill = reinterpret_cast<uint2 *>(base + L.off_spill);
  P.list2 = reinterpret_cast<uint2 *>(base + L.off_list2);
  P.seg_cnt = reinterpret_cast<uint32_t *>(base + L.off_seg);
  P.fin = reinterpret_cast<u64 *>(base + L.off_fin);
  return P;
}

template <bool VO, bool STORE>
inline void launch_main(int nk, const FastLayout &L, const FastPtrs &P, hipStream_t st, const float *acc, int64_t n,
                        const MaskPtrs &mp) {
#define SALUN_MAIN(NKT)                                                                                                 \
  hipLaunchKernelGGL((k_main<NKT, VO, STORE>), dim3(L.grid), dim3(SALUN_BLOCK), 0, st, acc, n, P.fs, mp, P.slabs,       \
                     P.slab_cnt, P.wg_gt, P.rows, P.wg_zero, P.wg_min, P.spill, L.cap, L.spill_cap, nk)
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

// k_main .. k_finish + the fallback / tie-pass launch of one fast instance whose FastState already holds the brackets.
inline int run_fast_tail(const float *acc, int64_t n, const KList &kl, const MaskPtrs &mp, bool values_only, char *fast_base,
                         const FastLayout &L, TopkPub *pub, FullState *full, u64 *tie, bool aligned, bool maligned,
                         hipStream_t st) {
  const int nk = kl.nk;
  const FastPtrs P = fast_ptrs(fast_base, L);
  if (values_only) launch_main<true, true>(nk, L, P, st, acc, n, mp);
  else launch_main<false, true>(nk, L, P, st, acc, n, mp);
  SALUN_LAUNCH_CHECK();
  if (values_only)
    hipLaunchKernelGGL(k_resolve<true>, dim3(L.gr * nk), dim3(1024), 0, st, P.fs, P.slabs, P.slab_cnt, P.wg_gt, P.rows,
                       P.wg_zero, P.spill, L.cap, L.spill_cap, L.grid, nk, L.gr, L.seg_cap, P.list2, P.seg_cnt, mp);
  else
    hipLaunchKernelGGL(k_resolve<false>, dim3(L.gr * nk), dim3(1024), 0, st, P.fs, P.slabs, P.slab_cnt, P.wg_gt, P.rows,
                       P.wg_zero, P.spill, L.cap, L.spill_cap, L.grid, nk, L.gr, L.seg_cap, P.list2, P.seg_cnt, mp);
  SALUN_LAUNCH_CHECK();
  if (values_only)
    hipLaunchKernelGGL(k_finish<true>, dim3(nk * L.gf), dim3(1024), 0, st, P.fs, pub, P.list2, P.seg_cnt, L.gr, L.seg_cap,
                       P.wg_min, L.grid, P.fin, nk, mp);
  else
    hipLaunchKernelGGL(k_finish<false>, dim3(nk * L.gf), dim3(1024), 0, st, P.fs, pub, P.list2, P.seg_cnt, L.gr, L.seg_cap,
                       P.wg_min, L.grid, P.fin, nk, mp);
  SALUN_LAUNCH_CHECK();
  const size_t lds_bytes = D0_BINS + sizeof(uint32_t) * (size_t)(nk < 2 ? 2 : nk) * 1024;
  hipLaunchKernelGGL(k_fullscan, dim3(fullscan_grid(n)), dim3(SALUN_BLOCK), lds_bytes, st, acc, n, kl, mp, pub, full, P.fs,
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
  char *fast_base = base + W.off_fast;
  const FastPtrs PO = fast_ptrs(fast_base, W.outer);
  const bool two_level = W.two_level;
  if (!two_level) {
    const int S = sample_size(n);
    hipLaunchKernelGGL(k_sample, dim3(S / 1024), dim3(1024), 0, st, acc, n, S, PO.keys, PO.fs, pub, full, nk);
    SALUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bracket, dim3(nk), dim3(1024), 0, st, PO.keys, n, kl, S, PO.fs);
    SALUN_LAUNCH_CHECK();
  } else {
    // ---- brackets = exact order statistics of a 2^20-element sample (this same route on the sample, values only)
    const int64_t S2 = int64_t(1) << SAMPLE2_LOG2;
    float *samp = reinterpret_cast<float *>(base + W.off_sample);
    TopkPub *ipub = reinterpret_cast<TopkPub *>(base + W.off_ipub);
    FullState *ifull = reinterpret_cast<FullState *>(base + W.off_ifull);
    const FastPtrs PI = fast_ptrs(base + W.off_ifast, W.inner);
    KList k2;
    RankList rl;
    k2.nk = 2 * nk;
    for (int j = 0; j < MAXK; ++j) { k2.k[j] = 0; rl.hi[j] = rl.lo[j] = 0; }
    for (int j = 0; j < nk; ++j) {
      const long long k = ks[j];
      if (k <= 0 || k >= (long long)n) { k2.k[2 * j] = k2.k[2 * j + 1] = 0; continue; }  // no bracket needed
      const double p = (double)k / (double)n;
      const double sigma = std::sqrt((double)S2 * p * (1.0 - p));
      const long long margin = (long long)std::ceil(BRACKET_SIGMAS * sigma) + 8;
      const long long rho = std::llround(p * (double)S2);
      long long rhi = rho - margin, rlo = rho + margin;
      if (rhi < 1) rhi = 0;
      if (rlo >= S2) rlo = 0;            // (the sample's own k == n case is "unbounded below")
      rl.hi[j] = rhi; rl.lo[j] = rlo;
      k2.k[2 * j] = rhi ? rhi : 1;       // unbounded sides are not read back
      k2.k[2 * j + 1] = rlo ? rlo : 1;
    }
    const int S1 = sample_size(S2);
    hipLaunchKernelGGL(k_gather_sample, dim3((unsigned)(S2 / SALUN_BLOCK)), dim3(SALUN_BLOCK), 0, st, acc, n / S2, S2, samp,
                       (int)(S2 / S1), PI.keys, PI.fs, ipub, ifull, 2 * nk);
    SALUN_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bracket, dim3(2 * nk), dim3(1024), 0, st, PI.keys, S2, k2, S1, PI.fs);
    SALUN_LAUNCH_CHECK();
    {  // the sample's streaming pass leaves the histogram rows; its resolution stops there (k_bracket_from_hist)
      MaskPtrs none;
      for (int j = 0; j < MAXK; ++j) none.m[j] = nullptr;
      launch_main<true, false>(2 * nk, W.inner, PI, st, samp, S2, none);  // count only: nothing reads the sample's slabs
      SALUN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_bracket_from_hist, dim3(1), dim3(1024), 0, st, PI.fs, PI.rows, PI.wg_gt, PI.wg_zero, W.inner.grid,
                       n, kl, rl, PO.fs, pub, full);
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
