// salun_gemm.hip — fp32 GEMMs of the diffusion U-Nets on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// FMA chains), and the row softmax between the two GEMMs of an fp32 attention.  gfx950 only.
//
// What runs here (reference modules replaced — the library GEMMs PyTorch dispatches for them):
//   * Linear layers:  DDPM/models/diffusion.py:85-145 (temb / cemb dense pairs, every ResnetBlock's temb_cemb_proj),
//     SD/ldm/modules/attention.py:37-75,149-200 and openaimodel.py:428-520 in the fp32 configuration;
//   * fp32 attention: DDPM/models/diffusion.py:148-192 (AttnBlock: bmm -> softmax -> bmm over 256 tokens),
//     SD/ldm/modules/attention.py:168-192 (einsum -> softmax -> einsum).
//
// ONE kernel computes a TABLE of problems  C_j[M_j, N_j] (+)= alpha * sum over segments s of A_s . B_s^T  (+ bias_j):
//   job      an output matrix (row-major, leading dimension ldc), optional bias per column, write or accumulate;
//   segment  one (A, B, K) pair of a job's reduction: A(i, k) = A[i * a_i + k * a_k], B(j, k) = B[j * b_j + k * b_k]
//            — arbitrary element strides, so x.W^T, dY.W, dY^T.x and the channel-major attention operands of the DDPM
//            (tokens contiguous) are all the same kernel with no transposing copy.
// A plain GEMM is 1 job x 1 segment.  The DDPM's 22 embedding projections (one per ResnetBlock, all reading the same
// [batch, 1024] activation) are 22 jobs in one launch; their input gradient sum_g dproj_g . W_g is 1 job x 22 segments;
// their weight gradients 22 jobs again — 3 launches per pass instead of 66.  A batch dimension (two-level: outer x
// inner strides, i.e. image x head) covers attention.
//
// Tiling: 256 threads = 2 x 2 waves, each wave WT x WT MFMA tiles of 32 x 32 (block tile 64 WT square), reduction in
// chunks of 16 staged through LDS rows of 17 floats (conflict-free ds_read_b32 across the 32 rows of an operand);
// the next chunk's global loads are issued before the current chunk's MFMAs.  Loads are 16-byte wherever a unit-stride
// dimension and the alignment allow (mode chosen per segment on the host), scalar otherwise.  Under-filled launches
// split the reduction over blockIdx.z into fp32 partial images that a second kernel folds in a fixed order (no float
// atomics anywhere: results are deterministic).
#include "salun_common.h"
#include <type_traits>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAXJ = SALUN_GEMM_MAX_JOBS;
constexpr int MAXS = SALUN_GEMM_MAX_SEGS;
constexpr int BK = 16;
constexpr int LDK = BK + 1;

struct Seg {
  const float *A, *B;
  int K, a_i, a_k, b_j, b_k;
  int chunk0;        // index of this segment's first chunk in the job's chunk list
  int amode, bmode;  // 0 scalar, 1 16-byte loads along k (stride_k == 1), 2 16-byte loads along the row index
};
struct Job {
  float *C;
  const float *bias;
  int M, N, ldc, seg0, nseg, nchunks, tile0, tiles_n, accumulate;
  long long ws_off;  // element offset of this job's partial image inside one split slab
};
struct Args {
  Job job[MAXJ];
  Seg seg[MAXS];
  int njobs, nsplit, batch_inner;
  long long a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;  // batch strides (outer, inner) in elements
  float alpha;
  float *ws;
  long long ws_slab;   // elements per split slab (all jobs, all batches)
  long long ws_batch;  // elements per batch inside a slab
};

template <int NV>
struct Stage {
  float4 v[NV];
};

// One operand's share of a chunk for this thread: rows [r0, r0 + rows) x k [k0, k0 + BK), element (row, k) at
// p[row * s_row + k * s_k]; out-of-range elements are zero.
template <int BT, int NV>
__device__ __forceinline__ void stage_load(Stage<NV> &st, const float *__restrict__ p, int s_row, int s_k, int r0,
                                           int nrows, int k0, int nk, int mode, int tid) {
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int e = tid + 256 * r;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 1) {  // four consecutive k of one row
      const int row = e >> 2, kk = (e & 3) * 4;
      const int gr = r0 + row, gk = k0 + kk;
      if (gr < nrows && gk < nk) {
        const float *q = p + (long long)gr * s_row + gk;
        if (gk + 3 < nk) val = *reinterpret_cast<const float4 *>(q);
        else {
          val.x = q[0];
          if (gk + 1 < nk) val.y = q[1];
          if (gk + 2 < nk) val.z = q[2];
        }
      }
    } else if (mode == 2) {  // four consecutive rows at one k
      constexpr int F4 = BT / 4;
      const int kk = e / F4, row = (e % F4) * 4;
      const int gr = r0 + row, gk = k0 + kk;
      if (gr < nrows && gk < nk) {
        const float *q = p + (long long)gk * s_k + gr;
        if (gr + 3 < nrows) val = *reinterpret_cast<const float4 *>(q);
        else {
          val.x = q[0];
          if (gr + 1 < nrows) val.y = q[1];
          if (gr + 2 < nrows) val.z = q[2];
        }
      }
    } else {  // scalar: element e*4+c -> (row, k) = (idx / 16, idx % 16)
      float t[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = e * 4 + c, row = idx >> 4, kk = idx & 15;
        const int gr = r0 + row, gk = k0 + kk;
        t[c] = (gr < nrows && gk < nk) ? p[(long long)gr * s_row + (long long)gk * s_k] : 0.f;
      }
      val = make_float4(t[0], t[1], t[2], t[3]);
    }
    st.v[r] = val;
  }
}

// The same share of a chunk that lies wholly inside an operand staged as float4 rows (mode 1 or 2): one float4 per item
// at a computed address, no guard.  Behind a guard (stage_load) every loaded value is copied into the merged result, and
// the copy waits for the load: s_waitcnt vmcnt(0) straight after it — the "next chunk's loads fly while this one is
// multiplied" of the loop below did not hold for a single load (tools/isa_serial_loads.py).
template <int BT, int NV>
__device__ __forceinline__ void stage_load_vec(Stage<NV> &st, const float *__restrict__ p, int s_row, int s_k, int r0, int k0,
                                               int mode, int tid) {
  constexpr int F4 = BT / 4;
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int e = tid + 256 * r;
    const long long o1 = (long long)(r0 + (e >> 2)) * s_row + k0 + (e & 3) * 4;   // four consecutive k of one row
    const long long o2 = (long long)(k0 + e / F4) * s_k + r0 + (e % F4) * 4;      // four consecutive rows at one k
    st.v[r] = *reinterpret_cast<const float4 *>(p + (mode == 1 ? o1 : o2));
  }
}

template <int BT, int NV>
__device__ __forceinline__ void stage_store(float *__restrict__ lds, const Stage<NV> &st, int mode, int tid) {
#pragma unroll
  for (int r = 0; r < NV; ++r) {
    const int e = tid + 256 * r;
    const float4 v = st.v[r];
    if (mode == 1) {
      float *d = lds + (e >> 2) * LDK + (e & 3) * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else if (mode == 2) {
      constexpr int F4 = BT / 4;
      float *d = lds + ((e % F4) * 4) * LDK + e / F4;
      d[0] = v.x; d[LDK] = v.y; d[2 * LDK] = v.z; d[3 * LDK] = v.w;
    } else {
      const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int idx = e * 4 + c;
        lds[(idx >> 4) * LDK + (idx & 15)] = t[c];
      }
    }
  }
}

template <int WT>
__global__ __launch_bounds__(256) void k_gemm_f32(const Args a) {
  constexpr int BT = 64 * WT;          // block tile (square)
  constexpr int NV = BT * BK / 4 / 256;  // float4 items per thread and operand: WT
  __shared__ float As[BT * LDK];
  __shared__ float Bs[BT * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- which job / tile / batch / split
  int j = 0;
  const int tile = blockIdx.x;
  while (j + 1 < a.njobs && tile >= a.job[j + 1].tile0) ++j;
  const Job J = a.job[j];
  const int t = tile - J.tile0;
  const int m0 = (t / J.tiles_n) * BT, n0 = (t % J.tiles_n) * BT;
  const int bz = blockIdx.y;
  const int bo = bz / a.batch_inner, bi = bz - bo * a.batch_inner;
  const long long a_off = (long long)bo * a.a_bo + (long long)bi * a.a_bi;
  const long long b_off = (long long)bo * a.b_bo + (long long)bi * a.b_bi;
  const long long c_off = (long long)bo * a.c_bo + (long long)bi * a.c_bi;
  const int z = blockIdx.z;
  const int per = (J.nchunks + a.nsplit - 1) / a.nsplit;
  const int c_lo = z * per, c_hi = (c_lo + per < J.nchunks) ? c_lo + per : J.nchunks;

  f32x16 acc[WT][WT];
#pragma unroll
  for (int x = 0; x < WT; ++x)
#pragma unroll
    for (int y = 0; y < WT; ++y)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[x][y][v] = 0.f;

  // FAST: the tile lies inside both operands, every segment is staged as float4 rows and is a whole number of chunks ->
  // the unguarded loads (stage_load_vec); otherwise the guarded ones.  Two copies of the loop, chosen once per workgroup:
  // a per-chunk choice would merge the two kinds of loaded values again.
  auto run = [&](auto fastc) {
    constexpr bool FAST = decltype(fastc)::value;
    int s = J.seg0;
    while (s + 1 < J.seg0 + J.nseg && a.seg[s + 1].chunk0 <= c_lo) ++s;
    Seg S = a.seg[s];
    Stage<NV> ra, rb;
    auto load = [&](int c) {
      const int k0 = (c - S.chunk0) * BK;
      if constexpr (FAST) {
        stage_load_vec<BT, NV>(ra, S.A + a_off, S.a_i, S.a_k, m0, k0, S.amode, tid);
        stage_load_vec<BT, NV>(rb, S.B + b_off, S.b_j, S.b_k, n0, k0, S.bmode, tid);
      } else {
        stage_load<BT, NV>(ra, S.A + a_off, S.a_i, S.a_k, m0, J.M, k0, S.K, S.amode, tid);
        stage_load<BT, NV>(rb, S.B + b_off, S.b_j, S.b_k, n0, J.N, k0, S.K, S.bmode, tid);
      }
    };
    load(c_lo);
    int amode = S.amode, bmode = S.bmode;
    for (int c = c_lo; c < c_hi; ++c) {
      stage_store<BT, NV>(As, ra, amode, tid);
      stage_store<BT, NV>(Bs, rb, bmode, tid);
      __syncthreads();
      if (c + 1 < c_hi) {  // the next chunk's loads fly while this one is multiplied
        if (s + 1 < J.seg0 + J.nseg && a.seg[s + 1].chunk0 <= c + 1) { ++s; S = a.seg[s]; }
        load(c + 1);
        amode = S.amode; bmode = S.bmode;
      }
      const float *ar = As + (wm * 32 * WT + lo) * LDK + hi;
      const float *br = Bs + (wn * 32 * WT + lo) * LDK + hi;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float av[WT], bv[WT];
#pragma unroll
        for (int x = 0; x < WT; ++x) { av[x] = ar[x * 32 * LDK + 2 * kk]; bv[x] = br[x * 32 * LDK + 2 * kk]; }
#pragma unroll
        for (int x = 0; x < WT; ++x)
#pragma unroll
          for (int y = 0; y < WT; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
      __syncthreads();
    }
  };
  if (c_lo < c_hi) {
    bool fast = m0 + BT <= J.M && n0 + BT <= J.N;
    for (int q = J.seg0; q < J.seg0 + J.nseg; ++q) {
      const Seg &T = a.seg[q];
      fast = fast && T.amode != 0 && T.bmode != 0 && T.K % BK == 0;
    }
    if (fast) run(std::true_type{});
    else run(std::false_type{});
  }

  // ---- epilogue: D[i][j] of a 32x32 tile sits in lane (j = lo, half = hi), register v: i = (v&3) + 8 (v>>2) + 4 hi
  const bool direct = a.nsplit == 1;
  float *out = direct ? J.C + c_off
                      : a.ws + (long long)z * a.ws_slab + (long long)bz * a.ws_batch + J.ws_off;
  const int ld = direct ? J.ldc : J.N;
#pragma unroll
  for (int x = 0; x < WT; ++x)
#pragma unroll
    for (int y = 0; y < WT; ++y) {
      const int jj = n0 + wn * 32 * WT + y * 32 + lo;
      if (jj >= J.N) continue;
      const float bsv = (direct && J.bias) ? J.bias[jj] : 0.f;
      // `+=`: the sixteen old values of a tile are read before its first store — element by element every load waited
      // alone behind the previous store (64 serial trips to memory per lane, see conv_igemm's epilogue in salun_conv.hip)
      const int ib = m0 + wm * 32 * WT + x * 32 + 4 * hi;
      float old[16];
      // unconditional loads (rows past M re-read the last row): a load behind a per-lane test becomes a branch with its
      // own wait
      if (direct && J.accumulate) {
#pragma unroll
        for (int v = 0; v < 16; ++v) old[v] = out[(long long)min(ib + (v & 3) + 8 * (v >> 2), J.M - 1) * ld + jj];
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int ii = ib + (v & 3) + 8 * (v >> 2);
        if (ii >= J.M) continue;
        float *d = out + (long long)ii * ld + jj;
        if (direct) {
          const float r = acc[x][y][v] * a.alpha + bsv;
          *d = J.accumulate ? (old[v] + r) : r;
        } else {
          *d = acc[x][y][v];
        }
      }
    }
}

// folds the split slabs in index order, then alpha / bias / accumulate as the direct epilogue does
__global__ __launch_bounds__(256) void k_gemm_finish(const Args a) {
  const Job J = a.job[blockIdx.y];
  const int bz = blockIdx.z;
  const int bo = bz / a.batch_inner, bi = bz - bo * a.batch_inner;
  float *C = J.C + (long long)bo * a.c_bo + (long long)bi * a.c_bi;
  const float *w = a.ws + (long long)bz * a.ws_batch + J.ws_off;
  const long long total = (long long)J.M * J.N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    float s = w[e];
    for (int zz = 1; zz < a.nsplit; ++zz) s += w[(long long)zz * a.ws_slab + e];
    const int i = (int)(e / J.N), jj = (int)(e - (long long)i * J.N);
    const float r = s * a.alpha + (J.bias ? J.bias[jj] : 0.f);
    float *d = C + (long long)i * J.ldc + jj;
    *d = J.accumulate ? (*d + r) : r;
  }
}

// ------------------------------------------------------------------------------------------------ row softmax
// One wave per row (rows of n <= ld contiguous floats): three passes over a row that is a few KiB and stays in L2.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_softmax_rows(float *__restrict__ s, long long rows, int n, int ld) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
    float *row = s + r * ld;
    float m = -INFINITY;
    for (int jj = lane; jj < n; jj += 64) m = fmaxf(m, row[jj]);
    m = wave_max(m);
    float sum = 0.f;
    for (int jj = lane; jj < n; jj += 64) {
      const float e = __expf(row[jj] - m);
      row[jj] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int jj = lane; jj < n; jj += 64) row[jj] *= inv;
  }
}

// dS = scale * P * (dP - sum_j dP_j P_j), written over dP
__global__ __launch_bounds__(256) void k_softmax_rows_bwd(const float *__restrict__ p, float *__restrict__ dp, long long rows,
                                                          int n, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
    const float *pr = p + r * ld;
    float *dr = dp + r * ld;
    float dot = 0.f;
    for (int jj = lane; jj < n; jj += 64) dot += pr[jj] * dr[jj];
    dot = wave_sum(dot);
    for (int jj = lane; jj < n; jj += 64) dr[jj] = scale * (pr[jj] * (dr[jj] - dot));
  }
}

// ------------------------------------------------------------------------------------------------ column sums
// out[j] (+)= sum_i x[i * ld + j] — the bias gradient of a Linear layer.  Stage 1: a workgroup owns 64 columns x one
// slab of rows (a wave reads 64 consecutive floats of a row), 4 row lanes folded through LDS; stage 2 folds the slabs
// in index order.
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ x, float *__restrict__ part, long long M,
                                                        int N, long long ld, int slabs) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6, slab = blockIdx.y;
  const long long per = (M + slabs - 1) / slabs;
  const long long r0 = slab * per, r1 = (r0 + per < M) ? r0 + per : M;
  float acc = 0.f;
  if (col < N)
    for (long long r = r0 + rl; r < r1; r += 4) acc += x[r * ld + col];
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && col < N) part[(long long)slab * N + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_colsum_final(const float *__restrict__ part, float *__restrict__ out, int N, int slabs,
                                                      int accumulate) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= N) return;
  float t = 0.f;
  for (int s2 = 0; s2 < slabs; ++s2) t += part[(long long)s2 * N + col];
  out[col] = accumulate ? out[col] + t : t;
}
inline int colsum_slabs(long long M, int N) {
  long long s = 1024 / ((N + 63) / 64);  // ~1024 workgroups
  if (s < 1) s = 1;
  if (s > (M + 15) / 16) s = (M + 15) / 16;  // >= 16 rows per slab
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

bool aligned16_stride(const void *p, long long bo, long long bi, int s) {
  return salun_aligned16(p) && (bo % 4 == 0) && (bi % 4 == 0) && (s % 4 == 0);
}

}  // namespace

// ================================================================== C-ABI =======
static int gemm_tile_for(const salun_gemm_job_t *jobs, int njobs, int batch) {
  // 128 x 128 block tiles when they alone fill the chip, else 64 x 64
  long long t128 = 0;
  for (int j = 0; j < njobs; ++j) t128 += (long long)((jobs[j].M + 127) / 128) * ((jobs[j].N + 127) / 128);
  return (t128 * batch >= 512) ? 128 : 64;
}

static int gemm_plan(const salun_gemm_job_t *jobs, int njobs, const salun_gemm_seg_t *segs, int nsegs, int batch,
                     int *tile_out, long long *tiles_out, int *split_out, long long *out_elems_out) {
  if (!jobs || !segs || njobs < 1 || njobs > MAXJ || nsegs < 1 || nsegs > MAXS || batch < 1) return SALUN_EINVAL;
  const int bt = gemm_tile_for(jobs, njobs, batch);
  long long tiles = 0, out_elems = 0;
  int min_chunks = 1 << 30;
  for (int j = 0; j < njobs; ++j) {
    const salun_gemm_job_t &J = jobs[j];
    if (J.M < 1 || J.N < 1 || J.nseg < 1 || J.seg0 < 0 || J.seg0 + J.nseg > nsegs || !J.C || J.ldc < J.N) return SALUN_EINVAL;
    tiles += (long long)((J.M + bt - 1) / bt) * ((J.N + bt - 1) / bt);
    out_elems += (long long)J.M * J.N;
    int chunks = 0;
    for (int s = J.seg0; s < J.seg0 + J.nseg; ++s) {
      if (segs[s].K < 1 || !segs[s].A || !segs[s].B) return SALUN_EINVAL;
      chunks += (segs[s].K + BK - 1) / BK;
    }
    if (chunks < min_chunks) min_chunks = chunks;
  }
  // split the reduction while the launch leaves the chip under-filled and every share keeps >= 4 chunks
  int split = 1;
  const long long wgs = tiles * batch;
  if (wgs < 256) {
    split = (int)(512 / (wgs < 1 ? 1 : wgs));
    if (split > 16) split = 16;
    while (split > 1 && min_chunks / split < 4) --split;
  }
  *tile_out = bt; *tiles_out = tiles; *split_out = split < 1 ? 1 : split; *out_elems_out = out_elems;
  return SALUN_OK;
}

SALUN_EXPORT size_t salun_gemm_f32_workspace_bytes(const salun_gemm_job_t *jobs, int njobs, const salun_gemm_seg_t *segs,
                                                   int nsegs, int batch_outer, int batch_inner) {
  int bt, split;
  long long tiles, out_elems;
  if (batch_outer < 1 || batch_inner < 1) return 0;
  if (gemm_plan(jobs, njobs, segs, nsegs, batch_outer * batch_inner, &bt, &tiles, &split, &out_elems) != SALUN_OK) return 0;
  return split > 1 ? sizeof(float) * (size_t)split * (size_t)out_elems * (size_t)(batch_outer * batch_inner) : 0;
}

SALUN_EXPORT int salun_gemm_f32(const salun_gemm_job_t *jobs, int njobs, const salun_gemm_seg_t *segs, int nsegs,
                                int batch_outer, int batch_inner, const int64_t *batch_strides /* a_bo a_bi b_bo b_bi c_bo c_bi */,
                                double alpha, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (batch_outer < 1 || batch_inner < 1) return SALUN_EINVAL;
  const int batch = batch_outer * batch_inner;
  if (batch > 1 && !batch_strides) return SALUN_EINVAL;
  int bt, split;
  long long tiles, out_elems;
  const int rc = gemm_plan(jobs, njobs, segs, nsegs, batch, &bt, &tiles, &split, &out_elems);
  if (rc != SALUN_OK) return rc;
  if (batch > 65535 || tiles > 0x7fffffffLL) return SALUN_EINVAL;
  if (split > 1) {
    const size_t need = sizeof(float) * (size_t)split * (size_t)out_elems * (size_t)batch;
    if (!ws || ws_bytes < need || !salun_aligned4(ws)) split = 1;  // no room: one workgroup per tile does the whole reduction
  }
  Args a;
  a.njobs = njobs; a.nsplit = split; a.batch_inner = batch_inner;
  a.a_bo = a.a_bi = a.b_bo = a.b_bi = a.c_bo = a.c_bi = 0;
  if (batch_strides) {
    a.a_bo = batch_strides[0]; a.a_bi = batch_strides[1]; a.b_bo = batch_strides[2];
    a.b_bi = batch_strides[3]; a.c_bo = batch_strides[4]; a.c_bi = batch_strides[5];
  }
  a.alpha = (float)alpha;
  a.ws = static_cast<float *>(ws);
  a.ws_batch = out_elems;
  a.ws_slab = out_elems * batch;
  long long tile0 = 0, ws_off = 0;
  for (int j = 0; j < njobs; ++j) {
    const salun_gemm_job_t &J = jobs[j];
    Job &d = a.job[j];
    d.C = J.C; d.bias = J.bias; d.M = J.M; d.N = J.N; d.ldc = J.ldc; d.seg0 = J.seg0; d.nseg = J.nseg;
    d.accumulate = J.accumulate ? 1 : 0;
    d.tile0 = (int)tile0; d.tiles_n = (J.N + bt - 1) / bt;
    tile0 += (long long)((J.M + bt - 1) / bt) * d.tiles_n;
    d.ws_off = ws_off; ws_off += (long long)J.M * J.N;
    int chunk0 = 0;
    for (int s = J.seg0; s < J.seg0 + J.nseg; ++s) {
      const salun_gemm_seg_t &S = segs[s];
      Seg &e = a.seg[s];
      e.A = S.A; e.B = S.B; e.K = S.K; e.a_i = S.a_i; e.a_k = S.a_k; e.b_j = S.b_j; e.b_k = S.b_k;
      e.chunk0 = chunk0; chunk0 += (S.K + BK - 1) / BK;
      e.amode = (S.a_k == 1 && aligned16_stride(S.A, a.a_bo, a.a_bi, S.a_i)) ? 1
              : (S.a_i == 1 && aligned16_stride(S.A, a.a_bo, a.a_bi, S.a_k)) ? 2 : 0;
      e.bmode = (S.b_k == 1 && aligned16_stride(S.B, a.b_bo, a.b_bi, S.b_j)) ? 1
              : (S.b_j == 1 && aligned16_stride(S.B, a.b_bo, a.b_bi, S.b_k)) ? 2 : 0;
    }
    d.nchunks = chunk0;
  }
  hipStream_t st = salun_hip_stream(stream);
  const dim3 grid((unsigned)tiles, (unsigned)batch, (unsigned)split);
  if (bt == 128) hipLaunchKernelGGL(k_gemm_f32<2>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_gemm_f32<1>, grid, dim3(256), 0, st, a);
  SALUN_LAUNCH_CHECK();
  if (split > 1) {
    long long max_mn = 1;
    for (int j = 0; j < njobs; ++j) {
      const long long mn = (long long)jobs[j].M * jobs[j].N;
      if (mn > max_mn) max_mn = mn;
    }
    long long gx = (max_mn + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_gemm_finish, dim3((unsigned)gx, (unsigned)njobs, (unsigned)batch), dim3(256), 0, st, a);
    SALUN_LAUNCH_CHECK();
  }
  return SALUN_OK;
}

SALUN_EXPORT int salun_softmax_rows(float *s, int64_t rows, int n, int ld, salun_stream_t stream) {
  if (rows < 0 || n < 1 || ld < n) return SALUN_EINVAL;
  if (rows == 0) return SALUN_OK;
  if (!s) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_softmax_rows, dim3(salun_grid_for(rows, 4)), dim3(256), 0, salun_hip_stream(stream), s,
                     (long long)rows, n, ld);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT int salun_softmax_rows_backward(const float *p, float *dp, int64_t rows, int n, int ld, double scale,
                                             salun_stream_t stream) {
  if (rows < 0 || n < 1 || ld < n) return SALUN_EINVAL;
  if (rows == 0) return SALUN_OK;
  if (!p || !dp) return SALUN_EINVAL;
  hipLaunchKernelGGL(k_softmax_rows_bwd, dim3(salun_grid_for(rows, 4)), dim3(256), 0, salun_hip_stream(stream), p, dp,
                     (long long)rows, n, ld, (float)scale);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

SALUN_EXPORT size_t salun_colsum_f32_workspace_bytes(int64_t M, int N) {
  if (M < 1 || N < 1) return 0;
  return sizeof(float) * (size_t)colsum_slabs(M, N) * (size_t)N;
}

SALUN_EXPORT int salun_colsum_f32(const float *x, float *out, int64_t M, int N, int64_t ld, int accumulate, void *ws,
                                  size_t ws_bytes, salun_stream_t stream) {
  if (M < 1 || N < 1 || ld < N || !x || !out || !ws) return SALUN_EINVAL;
  if (ws_bytes < salun_colsum_f32_workspace_bytes(M, N)) return SALUN_ENOSPC;
  const int slabs = colsum_slabs(M, N);
  hipStream_t st = salun_hip_stream(stream);
  float *part = static_cast<float *>(ws);
  hipLaunchKernelGGL(k_colsum_partial, dim3((N + 63) / 64, slabs), dim3(256), 0, st, x, part, (long long)M, N, (long long)ld,
                     slabs);
  SALUN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colsum_final, dim3((N + 255) / 256), dim3(256), 0, st, part, out, N, slabs, accumulate ? 1 : 0);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// =====================================================================================================================
// K16: bf16 GEMM  y[M, N] = x[M, K] . Wp[N, K]^T (+ bias[N]) (+ addend[M, N])  on v_mfma_f32_32x32x16_bf16 — the Linear
// layers of the Stable-Diffusion transformer blocks in the bf16 configuration (reference: autocast over
// SD/ldm/modules/attention.py:37-75 GEGLU / FeedForward, :149-200 CrossAttention to_q / to_k / to_v / to_out), which
// rounds 1-3 left on the library GEMM.  The input gradient dX = dY . W is the same kernel on the transposed weight
// image WpT[K, N] (both images are packed from the fp32 master weights once per optimizer step).
//
// Structure (cdna_hip_programming.md §5): both operand tiles go global -> LDS directly (`global_load_lds_dwordx4`: 16 B
// per lane, no staging registers, no ds_write pass).  An LDS row is one 64-element (128-byte) k-step of one matrix row;
// the DMA writes lane-linear, so the bank swizzle is applied to the SOURCE address (lane (row, c') fetches chunk
// c' ^ ((row >> 1) & 7) of its row) and again on the fragment reads — `ds_read_b128` then touches 16 distinct 16-byte
// slots per 16-lane group (conflict-free) while every 128-byte global row segment is still read whole.
// The instruction's A operand is the WEIGHT tile (rows = output features) and B the token tile, so a lane ends up with
// 4 consecutive output features of one token per accumulator quad: the epilogue stores 8 bytes per lane (bias / residual
// added in fp32, one rounding) instead of sixteen 2-byte stores.
// Workgroup = WGM x WGN waves, each a 64-feature x 64-token result tile (2 x 2 MFMA tiles); DB = double-buffered LDS
// with one barrier per k-step (loads of step s+1 in flight under the MFMAs of step s), else one buffer / two barriers
// and more workgroups per CU.  Workgroup ids are remapped so that each XCD walks a contiguous run of tiles (token tile
// fixed, feature tiles consecutive): the token tile is re-read from that XCD's L2.
// Requirements: K % 64 == 0, N % (64 WGN) == 0, 16-byte aligned pointers; M is arbitrary (rows past M re-read row M-1
// and are never stored).
namespace {

typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) char *lds_ptr_t;

struct GbArgs {
  const uint16_t *x;       // [M][K] bf16
  const uint16_t *w;       // [N][K] bf16
  const float *bias;       // [N] fp32 or null
  const uint16_t *addend;  // [M][N] bf16 or null
  uint16_t *y;             // [M][N] bf16
  int M, N, K;
  int tiles_m, tiles_n;
};

__device__ __forceinline__ uint32_t gb_pack2(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float gb_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float gb_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

template <int WGM, int WGN, bool DB>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_bf16_nt(const GbArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;  // token rows, feature rows of the block tile
  constexpr int ROWS = BMt + BNt;
  constexpr int STAGE = ROWS * 128;              // bytes: one 64-element k-step of every row
  constexpr int IPW = ROWS / 8 / NW;             // DMA instructions per wave and k-step (8 rows each)
  static_assert(ROWS % (8 * NW) == 0, "tile");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int wm = wave % WGM, wn = wave / WGM;

  // ---- XCD-aware tile order (bijective remap: XCD x gets a contiguous run of tile indices)
  const int ntile = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, q = ntile >> 3, r = ntile & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BMt, n0 = tn * BNt;

  // ---- per-lane DMA sources: instruction i of this wave moves block rows [8 (wave IPW + i), +8)
  uint32_t src_off[IPW];  // element offset of this lane's 16 bytes at k-step 0
  bool src_w[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int rr = 8 * (wave * IPW + i) + (lane >> 3);      // block row
    const int c = (lane & 7) ^ ((rr >> 1) & 7);             // source chunk for LDS chunk (lane & 7)
    if (rr < BMt) {
      int m = m0 + rr;
      if (m > g.M - 1) m = g.M - 1;
      src_off[i] = (uint32_t)m * (uint32_t)g.K + c * 8;
      src_w[i] = false;
    } else {
      src_off[i] = (uint32_t)(n0 + rr - BMt) * (uint32_t)g.K + c * 8;
      src_w[i] = true;
    }
  }
  auto issue = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const uint16_t *p = (src_w[i] ? g.w : g.x) + src_off[i] + ks * 64;
      lds_ptr_t d = (lds_ptr_t)(lds + buf * STAGE + (wave * IPW + i) * 1024);
      __builtin_amdgcn_global_load_lds(p, d, 16, 0, 0);
    }
  };

  f32x16 acc[2][2];  // [feature tile][token tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // fragment reads: row (64-row tile base + 32 t + lo), logical chunk 2 kk + hi -> physical chunk ^ ((row >> 1) & 7);
  // the tile bases are multiples of 32, so the swizzle term depends on the lane only
  const int sw = (lo >> 1) & 7;
  const int x_row = (wm * 64 + lo) * 128;          // token tile region starts at 0
  const int w_row = (BMt + wn * 64 + lo) * 128;    // feature tile region behind it
  int ch[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ch[kk] = ((2 * kk + hi) ^ sw) * 16;

  auto compute = [&](int buf) {
    const char *base = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const g_bf16x8 w0 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + ch[kk]);
      const g_bf16x8 w1 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + 32 * 128 + ch[kk]);
      const g_bf16x8 x0 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + ch[kk]);
      const g_bf16x8 x1 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + 32 * 128 + ch[kk]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][1], 0, 0, 0);
    }
  };

  const int nk = g.K >> 6;
  if (DB) {
    issue(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of step ks has landed
      __syncthreads();                     // ... everyone's has, and everyone is done reading the other buffer
      if (ks + 1 < nk) issue(ks + 1, (ks + 1) & 1);
      compute(ks & 1);
    }
  } else {
    for (int ks = 0; ks < nk; ++ks) {
      issue(ks, 0);
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
      compute(0);
      __syncthreads();
    }
  }

  // ---- epilogue: D[i = feature][j = token]; lane (j = lo, half hi), register v: i = (v & 3) + 8 (v >> 2) + 4 hi
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int m = m0 + wm * 64 + b * 32 + lo;
    if (m >= g.M) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * hi;
        float o0 = acc[a][b][4 * q], o1 = acc[a][b][4 * q + 1], o2 = acc[a][b][4 * q + 2], o3 = acc[a][b][4 * q + 3];
        if (g.bias) {
          const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n);
          o0 += bv.x; o1 += bv.y; o2 += bv.z; o3 += bv.w;
        }
        const size_t e = (size_t)m * g.N + n;
        if (g.addend) {
          const uint2 av = *reinterpret_cast<const uint2 *>(g.addend + e);
          o0 += gb_lo(av.x); o1 += gb_hi(av.x); o2 += gb_lo(av.y); o3 += gb_hi(av.y);
        }
        *reinterpret_cast<uint2 *>(g.y + e) = make_uint2(gb_pack2(o0, o1), gb_pack2(o2, o3));
      }
  }
}

// The same tile machinery as a PERSISTENT kernel: a workgroup walks a contiguous run of tiles (XCD-remapped order: the
// token tile stays, the feature tile advances) with ONE software pipeline flattened over (tile, k-step) — the first
// k-step of tile t+1 is already in flight while tile t's last k-step is multiplied and its epilogue stores run.  The
// SD Linear layers have SHORT reductions (K = 320: five k-steps), so a tile-per-workgroup kernel spends as long
// filling its pipeline as multiplying; here the fill is paid once per workgroup.
template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_bf16_nt_p(const GbArgs g, const int tiles_per_wg) {
  constexpr int NW = WGM * WGN;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  constexpr int ROWS = BMt + BNt;
  constexpr int STAGE = ROWS * 128;
  constexpr int IPW = ROWS / 8 / NW;
  static_assert(ROWS % (8 * NW) == 0, "tile");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int wm = wave % WGM, wn = wave / WGM;
  const int ntile = g.tiles_m * g.tiles_n;
  const int nk = g.K >> 6;

  // workgroup b of XCD x = b & 7 takes the (b >> 3)-th run of that XCD's contiguous tile range
  const int xcd = blockIdx.x & 7, q = ntile >> 3, r = ntile & 7;
  const int xcd_first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int xcd_count = xcd < r ? q + 1 : q;
  const int run0 = (blockIdx.x >> 3) * tiles_per_wg;
  const int my_tiles = min(tiles_per_wg, xcd_count - run0);
  if (my_tiles <= 0) return;
  const int tile0 = xcd_first + run0;

  uint32_t src_off[IPW];
  bool src_w[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) src_w[i] = (8 * (wave * IPW + i) + (lane >> 3)) >= BMt;
  auto aim = [&](int tile) {  // per-lane DMA sources of a tile (k-step 0)
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int m0 = tm * BMt, n0 = tn * BNt;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int rr = 8 * (wave * IPW + i) + (lane >> 3);
      const int c = (lane & 7) ^ ((rr >> 1) & 7);
      if (rr < BMt) {
        int m = m0 + rr;
        if (m > g.M - 1) m = g.M - 1;
        src_off[i] = (uint32_t)m * (uint32_t)g.K + c * 8;
      } else {
        src_off[i] = (uint32_t)(n0 + rr - BMt) * (uint32_t)g.K + c * 8;
      }
    }
  };
  auto issue = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const uint16_t *p = (src_w[i] ? g.w : g.x) + src_off[i] + ks * 64;
      lds_ptr_t d = (lds_ptr_t)(lds + buf * STAGE + (wave * IPW + i) * 1024);
      __builtin_amdgcn_global_load_lds(p, d, 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  const int sw = (lo >> 1) & 7;
  const int x_row = (wm * 64 + lo) * 128;
  const int w_row = (BMt + wn * 64 + lo) * 128;
  int ch[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ch[kk] = ((2 * kk + hi) ^ sw) * 16;

  aim(tile0);
  issue(0, 0);
  int ks = 0, t = 0, buf = 0;
  const int total = my_tiles * nk;
  for (int f = 0; f < total; ++f) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    // next stage of the flattened (tile, k-step) sequence
    int nks = ks + 1, nt = t;
    if (nks == nk) { nks = 0; nt = t + 1; }
    if (f + 1 < total) {
      if (nks == 0) aim(tile0 + nt);
      issue(nks, buf ^ 1);
    }
    const char *base = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const g_bf16x8 w0 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + ch[kk]);
      const g_bf16x8 w1 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + 32 * 128 + ch[kk]);
      const g_bf16x8 x0 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + ch[kk]);
      const g_bf16x8 x1 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + 32 * 128 + ch[kk]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][1], 0, 0, 0);
    }
    if (ks == nk - 1) {  // tile done: store it (the next tile's first k-step is already in flight) and clear
      const int tile = tile0 + t;
      const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
      const int m0 = tm * BMt, n0 = tn * BNt;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + lo;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int n = n0 + wn * 64 + a * 32 + 8 * qq + 4 * hi;
            float o0 = acc[a][b][4 * qq], o1 = acc[a][b][4 * qq + 1], o2 = acc[a][b][4 * qq + 2], o3 = acc[a][b][4 * qq + 3];
            acc[a][b][4 * qq] = 0.f; acc[a][b][4 * qq + 1] = 0.f; acc[a][b][4 * qq + 2] = 0.f; acc[a][b][4 * qq + 3] = 0.f;
            if (m < g.M) {
              if (g.bias) {
                const float4 bv = *reinterpret_cast<const float4 *>(g.bias + n);
                o0 += bv.x; o1 += bv.y; o2 += bv.z; o3 += bv.w;
              }
              const size_t e = (size_t)m * g.N + n;
              if (g.addend) {
                const uint2 av = *reinterpret_cast<const uint2 *>(g.addend + e);
                o0 += gb_lo(av.x); o1 += gb_hi(av.x); o2 += gb_lo(av.y); o3 += gb_hi(av.y);
              }
              *reinterpret_cast<uint2 *>(g.y + e) = make_uint2(gb_pack2(o0, o1), gb_pack2(o2, o3));
            }
          }
      }
    }
    ks = nks; t = nt; buf ^= 1;
  }
}

// Ring form: NST stages of LDS, the DMA of stages f+1 .. f+NST-2 stays in flight ACROSS the barrier of step f (counted
// `s_waitcnt vmcnt(N)`, raw `s_barrier` — `__syncthreads()` would drain the DMA queue, cdna_hip_programming.md §5
// "Pipelining across barriers").  The measurements that asked for it (tools/gemmbench_bf16.py, round 4): with one stage
// in flight a k-step took ~1.8 us against 0.21 us of MFMA work — the loop was bound by the latency of ONE global ->
// LDS transfer per step, whatever the tile shape, single- or double-buffered, persistent or not.
// Epilogue: the 64 x 64 result of a wave goes through that wave's own LDS region as [token][feature] fp32, so the
// global stores are 16 bytes per lane and whole 128-byte lines per 8 lanes (the 8-byte strided stores of the first
// version wrote a 168 MB output at 1.6 TB/s).
// s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left at their maxima) for the counts the ring instantiations need, as
// literal encodings.  (A hipcc / ROCm 7.2 trap met while writing this kernel, recorded here: when the SOURCE argument of
// __builtin_amdgcn_global_load_lds is an rvalue expression such as `src[i] + ks * 64` instead of a named pointer, the
// HOST pass silently fails to instantiate the kernel template and drops its launch stub — the library then fails to
// load with an undefined `__device_stub__` symbol.  Hence the named `p` in `issue` below.)
#define GB_WAIT_VM(N)                                                  \
  do {                                                                 \
    if constexpr ((N) == 0) __builtin_amdgcn_s_waitcnt(0x0f70);        \
    else if constexpr ((N) == 3) __builtin_amdgcn_s_waitcnt(0x0f73);   \
    else if constexpr ((N) == 4) __builtin_amdgcn_s_waitcnt(0x0f74);   \
    else if constexpr ((N) == 5) __builtin_amdgcn_s_waitcnt(0x0f75);   \
    else if constexpr ((N) == 6) __builtin_amdgcn_s_waitcnt(0x0f76);   \
    else if constexpr ((N) == 8) __builtin_amdgcn_s_waitcnt(0x0f78);   \
    else if constexpr ((N) == 10) __builtin_amdgcn_s_waitcnt(0x0f7a);  \
    else if constexpr ((N) == 12) __builtin_amdgcn_s_waitcnt(0x0f7c);  \
    else if constexpr ((N) == 16) __builtin_amdgcn_s_waitcnt(0x4f70);  \
    else if constexpr ((N) == 20) __builtin_amdgcn_s_waitcnt(0x4f74);  \
    else static_assert((N) < 0, "add the literal encoding of this vmcnt"); \
  } while (0)

// BK = 64: 128-byte rows, 8 lanes per row in a DMA unit, chunk swizzle (row >> 1) & 7.  BK = 32: 64-byte rows, 4 lanes per
// row, swizzle (row >> 2) & 3 (16 consecutive rows of one chunk column then cover the 64 banks exactly once) — half the
// LDS per stage, so TWO workgroups fit a CU: with K = 320 .. 640 a tile is only 5 .. 10 k-steps long and a lone
// workgroup spends as long filling its ring and writing its result as multiplying (r04 counters: matrix pipes busy 26 %,
// LDS 21 %, waves parked on waits 43 % — profiles/r04_gemm_sq_pmc_*.csv); a second resident workgroup multiplies meanwhile.
template <int WGM, int WGN, int NST, int BK>
__global__ __launch_bounds__(64 * WGM * WGN, (BK == 32 && WGM * WGN == 8) ? 4 : 1) void k_gemm_bf16_nt_r(const GbArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  constexpr int ROWS = BMt + BNt;
  constexpr int ROWB = BK * 2;           // bytes per staged row
  constexpr int LPR = BK / 8;            // lanes (16-byte chunks) per row
  constexpr int RPU = 64 / LPR;          // rows per 1 KB DMA unit
  constexpr int STAGE = ROWS * ROWB;
  constexpr int IPW = ROWS / RPU / NW;
  constexpr int KK = BK / 16;
  constexpr int EROW = 64 * 4 + 16;  // epilogue row: 64 fp32 features + pad (16-byte aligned, spreads the banks)
  constexpr int EP = (NW * 64 * EROW <= NST * STAGE) ? 1 : 2;   // passes of the epilogue staging (64 or 32 tokens per wave)
  static_assert(ROWS % (RPU * NW) == 0 && NST >= 3 && NST <= 4 && (BK == 64 || BK == 32), "tile");
  static_assert(NW * (64 / EP) * EROW <= NST * STAGE, "epilogue staging fits the ring");
  static_assert(IPW * (NST - 2) < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int wm = wave % WGM, wn = wave / WGM;

  const int ntile = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, q = ntile >> 3, r = ntile & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BMt, n0 = tn * BNt;

  // Full per-lane source POINTERS, fixed before the loop: a per-lane choice between the two base pointers inside the
  // loop makes hipcc fetch the base from the kernel-argument segment with an ordinary global load — and an ordinary load
  // next to LDS-DMA makes it wait vmcnt(0), which drains the ring (cdna_hip_programming.md §5, trap (b)).
  const uint16_t *src[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int rr = RPU * (wave * IPW + i) + lane / LPR;
    const int c = (BK == 64) ? ((lane & 7) ^ ((rr >> 1) & 7)) : ((lane & 3) ^ ((rr >> 2) & 3));
    int m = m0 + rr;
    if (m > g.M - 1) m = g.M - 1;
    const uint16_t *px = g.x + (size_t)m * g.K + c * 8;
    const uint16_t *pw = g.w + (size_t)(n0 + rr - BMt) * g.K + c * 8;
    src[i] = (rr < BMt) ? px : pw;
  }
  auto issue = [&](int ks, int buf) {
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      lds_ptr_t d = (lds_ptr_t)(lds + buf * STAGE + (wave * IPW + i) * 1024);
      const uint16_t *p = src[i] + ks * BK;
      __builtin_amdgcn_global_load_lds(p, d, 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  const int sw = (BK == 64) ? ((lo >> 1) & 7) : ((lo >> 2) & 3);
  const int x_row = (wm * 64 + lo) * ROWB;
  const int w_row = (BMt + wn * 64 + lo) * ROWB;
  int ch[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) ch[kk] = ((2 * kk + hi) ^ sw) * 16;

  auto compute = [&](int buf) {
    const char *base = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const g_bf16x8 w0 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + ch[kk]);
      const g_bf16x8 w1 = *reinterpret_cast<const g_bf16x8 *>(base + w_row + 32 * ROWB + ch[kk]);
      const g_bf16x8 x0 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + ch[kk]);
      const g_bf16x8 x1 = *reinterpret_cast<const g_bf16x8 *>(base + x_row + 32 * ROWB + ch[kk]);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][1], 0, 0, 0);
    }
  };

  const int nk = g.K / BK;
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0, s0);
  int buf = 0, nbuf = NST - 1;
  for (int f = 0; f < nk; ++f) {
    // stage f must have landed; up to NST-2 later stages (fewer at the tail) stay in flight across the barrier
    const int later = nk - 1 - f;
    if (later >= NST - 2) GB_WAIT_VM(IPW * (NST - 2));
    else if (NST == 4 && later == 1) GB_WAIT_VM(IPW);
    else GB_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();  // every wave's share of stage f is in LDS; everyone has finished reading stage f-1
    if (f + NST - 1 < nk) issue(f + NST - 1, nbuf);  // ... whose buffer is refilled now
    compute(buf);
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }

  // ---- epilogue through LDS: wave-private region, [token 64 / EP][feature 64] fp32 rows of EROW bytes
  __builtin_amdgcn_s_barrier();  // all waves are done with the stage buffers
  char *er = lds + wave * ((64 / EP) * EROW);
  const int fr = (lane & 7) * 8;       // 8 features per lane, 8 lanes per token row
  float bsv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsv[e] = 0.f;
  if (g.bias) {
    const float4 b0 = *reinterpret_cast<const float4 *>(g.bias + n0 + wn * 64 + fr);
    const float4 b1 = *reinterpret_cast<const float4 *>(g.bias + n0 + wn * 64 + fr + 4);
    bsv[0] = b0.x; bsv[1] = b0.y; bsv[2] = b0.z; bsv[3] = b0.w; bsv[4] = b1.x; bsv[5] = b1.y; bsv[6] = b1.z; bsv[7] = b1.w;
  }
#pragma unroll
  for (int pass = 0; pass < EP; ++pass) {
    // the residual rows of the pass are requested first, all of them and unconditionally (tokens past M re-read the last
    // one): read one by one next to their stores, each load waited alone behind the previous row's store
    uint4 adv[8 / EP];
    if (g.addend) {
#pragma unroll
      for (int i = 0; i < 8 / EP; ++i) {
        const int m = min(m0 + wm * 64 + pass * 32 + i * 8 + (lane >> 3), g.M - 1);
        adv[i] = *reinterpret_cast<const uint4 *>(g.addend + (size_t)m * g.N + n0 + wn * 64 + fr);
      }
    }
#pragma unroll
    for (int bb = 0; bb < 2 / EP; ++bb) {
      const int b = (EP == 2) ? pass : bb;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // D[i = feature][j = token]: lane (token lo, half hi), registers 4q..4q+3: features 8q + 4hi + 0..3
          float4 v4 = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
          *reinterpret_cast<float4 *>(er + (bb * 32 + lo) * EROW + (a * 32 + 8 * q + 4 * hi) * 4) = v4;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are done (wave-private region: no barrier)
#pragma unroll
    for (int i = 0; i < 8 / EP; ++i) {
      const int trow = i * 8 + (lane >> 3);
      const int m = m0 + wm * 64 + pass * 32 + trow;
      const float4 u0 = *reinterpret_cast<const float4 *>(er + trow * EROW + fr * 4);
      const float4 u1 = *reinterpret_cast<const float4 *>(er + trow * EROW + fr * 4 + 16);
      if (m < g.M) {
        float o[8] = {u0.x + bsv[0], u0.y + bsv[1], u0.z + bsv[2], u0.w + bsv[3],
                      u1.x + bsv[4], u1.y + bsv[5], u1.z + bsv[6], u1.w + bsv[7]};
        const size_t e = (size_t)m * g.N + n0 + wn * 64 + fr;
        if (g.addend) {
          const uint4 av = adv[i];
          o[0] += gb_lo(av.x); o[1] += gb_hi(av.x); o[2] += gb_lo(av.y); o[3] += gb_hi(av.y);
          o[4] += gb_lo(av.z); o[5] += gb_hi(av.z); o[6] += gb_lo(av.w); o[7] += gb_hi(av.w);
        }
        *reinterpret_cast<uint4 *>(g.y + e) = make_uint4(gb_pack2(o[0], o[1]), gb_pack2(o[2], o[3]), gb_pack2(o[4], o[5]),
                                                         gb_pack2(o[6], o[7]));
      }
    }
    if (EP == 2 && pass == 0) __builtin_amdgcn_s_waitcnt(0xc07f);  // the rows are read before the second half overwrites them
  }
}

template <int WGM, int WGN, int NST, int BK = 64>
int launch_gemm_bf16_r(const GbArgs &g0, hipStream_t st) {
  GbArgs g = g0;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  g.tiles_m = (g.M + BMt - 1) / BMt;
  g.tiles_n = g.N / BNt;
  const size_t ldsb = (size_t)NST * (BMt + BNt) * BK * 2;
  static unsigned long long configured = 0;  // one bit per device: the opt-in is per device, not per process
  if (salun_once_needed(&configured)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bf16_nt_r<WGM, WGN, NST, BK>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    salun_once_mark(&configured);
  }
  hipLaunchKernelGGL((k_gemm_bf16_nt_r<WGM, WGN, NST, BK>), dim3(g.tiles_m * g.tiles_n), dim3(64 * WGM * WGN), ldsb, st, g);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ------------------------------------------------------------------------------------------- K16 ring as a convolution
// The same ring with the A rows GATHERED: row m of the implicit GEMM is output pixel m, and for reduction stage
// (tap (r, s), 32 input channels from c0) its 64 bytes are x[n][oh*stride - pad + r][ow*stride - pad + s][c0 .. c0+31] of
// an NHWC tensor — or zeros outside the image.  LDS-DMA takes a per-lane source address, so the gather costs nothing
// extra: the per-pixel base pointer and a 9-bit validity mask are fixed before the loop, the (tap, c0) offset is
// wave-uniform, invalid rows read a 16-byte zero word.  Weight rows come from the packed image wp[K][R*R][C].  Output
// y[m][k] (NHWC) = sum + bias[k] + nbias[n][k] + addend[m][k], rounded once to bf16 (the epilogues of K11's forward,
// reference SD openaimodel.py ResBlock: `h + emb_out[..., None, None]`, skip `x + h`).
__device__ uint4 g_tn_zero = {0u, 0u, 0u, 0u};

struct IrArgs {
  const uint16_t *x;       // [N][H][W][Cin] bf16
  const uint16_t *wp;      // [Kout][R*R][Cin] bf16
  const float *bias;       // [Kout] or null
  const float *nbias;      // [N][Kout] or null
  const uint16_t *addend;  // [M][Kout] bf16 or null
  uint16_t *y;             // [M][Kout] bf16
  int M, H, W, Cin, OH, OW, Kout, R, stride, pad;
  int tiles_m, tiles_n;
};

template <int WGM, int WGN, int NST>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 8) ? 4 : 1) void k_conv_bf16_ring(const IrArgs g) {
  constexpr int NW = WGM * WGN;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  constexpr int ROWS = BMt + BNt;
  constexpr int ROWB = 64;               // 32 channels per staged row
  constexpr int STAGE = ROWS * ROWB;
  constexpr int IPW = ROWS / 16 / NW;
  constexpr int EROW = 64 * 4 + 16;
  constexpr int EP = (NW * 64 * EROW <= NST * STAGE) ? 1 : 2;
  static_assert(ROWS % (16 * NW) == 0 && NST >= 3 && NST <= 4, "tile");
  static_assert(NW * (64 / EP) * EROW <= NST * STAGE, "epilogue staging fits the ring");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lo = lane & 31, hi = lane >> 5;
  const int wm = wave % WGM, wn = wave / WGM;
  const int ntile = g.tiles_m * g.tiles_n;
  int tile;
  {
    const int id = blockIdx.x, xcd = id & 7, q = ntile >> 3, r = ntile & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BMt, n0 = tn * BNt;
  const int RS = g.R * g.R;

  const uint16_t *zp = reinterpret_cast<const uint16_t *>(&g_tn_zero);
  const uint16_t *base[IPW];
  uint32_t vmask[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int u = wave * IPW + i;
    const int rr = 16 * u + (lane >> 2);
    const int c = (lane & 3) ^ ((rr >> 2) & 3);
    if (u < BMt / 16) {      // pixel rows (wave-uniform: a unit never straddles the two operands)
      const int m = m0 + rr;
      const int ohow = g.OH * g.OW;
      const int mm = m < g.M ? m : g.M - 1;
      const int n = mm / ohow;
      const int rem = mm - n * ohow;
      const int oh = rem / g.OW, ow = rem - oh * g.OW;
      const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
      uint32_t vm = 0;
      if (m < g.M)
        for (int t = 0; t < RS; ++t) {
          const int ih = ih0 + t / g.R, iw = iw0 + t % g.R;
          if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) vm |= 1u << t;
        }
      vmask[i] = vm;
      base[i] = g.x + (((long long)n * g.H + ih0) * g.W + iw0) * g.Cin + c * 8;   // dereferenced only where vmask allows
    } else {
      const int k = n0 + rr - BMt;
      vmask[i] = k < g.Kout ? 0x1ffu : 0u;
      base[i] = g.wp + (long long)(k < g.Kout ? k : 0) * RS * g.Cin + c * 8;
    }
  }
  // the stage the NEXT issue() fetches: tap (ir, is), channel offset ic0 — advanced once per call
  int ir = 0, is = 0, ic0 = 0;
  auto issue = [&](int buf) {
    const int tap = ir * g.R + is;
    const long long off_x = ((long long)ir * g.W + is) * g.Cin + ic0;
    const long long off_w = (long long)tap * g.Cin + ic0;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int u = wave * IPW + i;
      lds_ptr_t d = (lds_ptr_t)(lds + buf * STAGE + u * 1024);
      const uint16_t *q = base[i] + (u < BMt / 16 ? off_x : off_w);
      const uint16_t *p = ((vmask[i] >> tap) & 1u) ? q : zp;
      __builtin_amdgcn_global_load_lds(p, d, 16, 0, 0);
    }
    ic0 += 32;
    if (ic0 == g.Cin) {
      ic0 = 0;
      if (++is == g.R) { is = 0; ++ir; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  const int sw = (lo >> 2) & 3;
  const int x_row = (wm * 64 + lo) * ROWB;
  const int w_row = (BMt + wn * 64 + lo) * ROWB;
  const int ch0 = ((0 + hi) ^ sw) * 16, ch1 = ((2 + hi) ^ sw) * 16;

  auto compute = [&](int buf) {
    const char *bs = lds + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = kk ? ch1 : ch0;
      const g_bf16x8 w0 = *reinterpret_cast<const g_bf16x8 *>(bs + w_row + ch);
      const g_bf16x8 w1 = *reinterpret_cast<const g_bf16x8 *>(bs + w_row + 32 * ROWB + ch);
      const g_bf16x8 x0 = *reinterpret_cast<const g_bf16x8 *>(bs + x_row + ch);
      const g_bf16x8 x1 = *reinterpret_cast<const g_bf16x8 *>(bs + x_row + 32 * ROWB + ch);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, x1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[1][1], 0, 0, 0);
    }
  };

  const int nk = RS * (g.Cin >> 5);
#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0);
  int buf = 0, nbuf = NST - 1;
  for (int f = 0; f < nk; ++f) {
    const int later = nk - 1 - f;
    if (later >= NST - 2) GB_WAIT_VM(IPW * (NST - 2));
    else if (NST == 4 && later == 1) GB_WAIT_VM(IPW);
    else GB_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (f + NST - 1 < nk) issue(nbuf);
    compute(buf);
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }

  // ---- epilogue through LDS (as k_gemm_bf16_nt_r), plus the per-image channel offset
  __builtin_amdgcn_s_barrier();
  char *er = lds + wave * ((64 / EP) * EROW);
  const int fr = (lane & 7) * 8;
  const int kcol = n0 + wn * 64 + fr;
  float bsv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsv[e] = 0.f;
  if (g.bias) {
    const float4 b0 = *reinterpret_cast<const float4 *>(g.bias + kcol);
    const float4 b1 = *reinterpret_cast<const float4 *>(g.bias + kcol + 4);
    bsv[0] = b0.x; bsv[1] = b0.y; bsv[2] = b0.z; bsv[3] = b0.w; bsv[4] = b1.x; bsv[5] = b1.y; bsv[6] = b1.z; bsv[7] = b1.w;
  }
  const int ohow = g.OH * g.OW;
  // The per-image channel offset: a wave's 64 pixels start at a multiple of 64, so with OH * OW a multiple of 64 (every
  // level of the SD U-Net) they belong to ONE image and its eight values are read once; otherwise per row, as before.
  const bool one_img = g.nbias && (ohow & 63) == 0;
  float nb1[8];
  if (one_img) {
    const float *nb = g.nbias + (size_t)(min(m0 + wm * 64, g.M - 1) / ohow) * g.Kout + kcol;
    const float4 n0v = *reinterpret_cast<const float4 *>(nb), n1v = *reinterpret_cast<const float4 *>(nb + 4);
    nb1[0] = n0v.x; nb1[1] = n0v.y; nb1[2] = n0v.z; nb1[3] = n0v.w; nb1[4] = n1v.x; nb1[5] = n1v.y; nb1[6] = n1v.z; nb1[7] = n1v.w;
  }
#pragma unroll
  for (int pass = 0; pass < EP; ++pass) {
    // the residual rows of the pass are requested first, unconditionally (see k_gemm_bf16_nt_r)
    uint4 adv[8 / EP];
    if (g.addend) {
#pragma unroll
      for (int i = 0; i < 8 / EP; ++i) {
        const int m = min(m0 + wm * 64 + pass * 32 + i * 8 + (lane >> 3), g.M - 1);
        adv[i] = *reinterpret_cast<const uint4 *>(g.addend + (size_t)m * g.Kout + kcol);
      }
    }
#pragma unroll
    for (int bb = 0; bb < 2 / EP; ++bb) {
      const int b = (EP == 2) ? pass : bb;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 v4 = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
          *reinterpret_cast<float4 *>(er + (bb * 32 + lo) * EROW + (a * 32 + 8 * q + 4 * hi) * 4) = v4;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int i = 0; i < 8 / EP; ++i) {
      const int trow = i * 8 + (lane >> 3);
      const int m = m0 + wm * 64 + pass * 32 + trow;
      const float4 u0 = *reinterpret_cast<const float4 *>(er + trow * EROW + fr * 4);
      const float4 u1 = *reinterpret_cast<const float4 *>(er + trow * EROW + fr * 4 + 16);
      if (m < g.M) {
        float o[8] = {u0.x + bsv[0], u0.y + bsv[1], u0.z + bsv[2], u0.w + bsv[3],
                      u1.x + bsv[4], u1.y + bsv[5], u1.z + bsv[6], u1.w + bsv[7]};
        if (one_img) {
#pragma unroll
          for (int e8 = 0; e8 < 8; ++e8) o[e8] += nb1[e8];
        } else if (g.nbias) {
          const float *nb = g.nbias + (size_t)(m / ohow) * g.Kout + kcol;
          const float4 n0v = *reinterpret_cast<const float4 *>(nb);
          const float4 n1v = *reinterpret_cast<const float4 *>(nb + 4);
          o[0] += n0v.x; o[1] += n0v.y; o[2] += n0v.z; o[3] += n0v.w; o[4] += n1v.x; o[5] += n1v.y; o[6] += n1v.z; o[7] += n1v.w;
        }
        const size_t e = (size_t)m * g.Kout + kcol;
        if (g.addend) {
          const uint4 av = adv[i];
          o[0] += gb_lo(av.x); o[1] += gb_hi(av.x); o[2] += gb_lo(av.y); o[3] += gb_hi(av.y);
          o[4] += gb_lo(av.z); o[5] += gb_hi(av.z); o[6] += gb_lo(av.w); o[7] += gb_hi(av.w);
        }
        *reinterpret_cast<uint4 *>(g.y + e) = make_uint4(gb_pack2(o[0], o[1]), gb_pack2(o[2], o[3]), gb_pack2(o[4], o[5]),
                                                         gb_pack2(o[6], o[7]));
      }
    }
    if (EP == 2 && pass == 0) __builtin_amdgcn_s_waitcnt(0xc07f);
  }
}

template <int WGM, int WGN, int NST>
int launch_conv_ring(const IrArgs &g0, hipStream_t st) {
  IrArgs g = g0;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  g.tiles_m = (g.M + BMt - 1) / BMt;
  g.tiles_n = g.Kout / BNt;
  const size_t ldsb = (size_t)NST * (BMt + BNt) * 64;
  static unsigned long long configured = 0;  // one bit per device: the opt-in is per device, not per process
  if (salun_once_needed(&configured)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv_bf16_ring<WGM, WGN, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    salun_once_mark(&configured);
  }
  hipLaunchKernelGGL((k_conv_bf16_ring<WGM, WGN, NST>), dim3(g.tiles_m * g.tiles_n), dim3(64 * WGM * WGN), ldsb, st, g);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// ------------------------------------------------------------------------------------------------ K16 TN (weight gradients)
// dW[Na][Nb] (fp32) (+)= sum_m dY[m][Na] . X[m][Nb]: the reduction index is the SLOW axis of both operands, so the 16-byte
// pieces the DMA drops into LDS hold 8 features of ONE token and an MFMA operand (8 tokens of one feature) is gathered by
// the transposing LDS read `ds_read_b64_tr_b16` (two per operand; layout and lane mapping as in K11's backward-weight,
// salun_conv_bf16.hip `tr_operand`).  Same ring as above (stages in flight across raw barriers); a stage = RST tokens of
// the (TA + TB) features of the tile, stored as [32-feature block][token][64 bytes] — lane-linear for the DMA (a wave
// instruction fills 16 tokens of one block), conflict-free for the transposing reads (a half-wave reads 4 whole 64-byte
// token rows twice).  Tokens past M and feature blocks past Na / Nb read a 16-byte zero word instead.
// The reduction is cut into `splits` ranges of whole stages (blockIdx.y); one range writes / adds into dW directly,
// several write fp32 partials that k_tn_reduce folds in index order (deterministic, no atomics).
typedef short g_s16x4 __attribute__((ext_vector_type(4)));
typedef g_s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr_t;

struct TnArgs {
  const uint16_t *dy;  // [M][Na] bf16
  const uint16_t *x;   // [M][Nb] bf16
  float *out;          // dW [Na][Nb] (splits == 1) or partials [splits][Na][Nb]
  int M, Na, Nb;
  int tiles_a, tiles_b;
  int stages, per_split;  // stages of RST tokens in total / per split
  int nsplit;
  int accumulate;         // splits == 1 only: dW += instead of dW =
};

// One stage of MFMA work as ONE hand-scheduled block.  With the `__builtin_amdgcn_ds_read_tr16_b64` intrinsic hipcc's
// wait-count pass cannot tell the transposing read from the LDS-DMA writes still in flight and puts `s_waitcnt vmcnt(0)` in
// front of the first read of every stage, which drains the ring; inline assembly carries no memory operand, so the counted
// waits + barrier of the loop stay the only synchronisation (they are sufficient: a stage is read only after its own wait
// and barrier).  Fragments live in fixed registers v[160:191] (two sets: the reads of 16-token step k+1 are in flight
// under the MFMAs of step k); `lgkmcnt` is counted in LDS reads — the block issues nothing else on that counter.
// Set S: dY operands v[S:S+3], v[S+4:S+7] (feature blocks 0, 1), X operands v[S+8:S+11], v[S+12:S+15].
#define TN_STR2(x) #x
#define TN_STR(x) TN_STR2(x)
#define TN_RD(S0, S1, S2, S3, S4, S5, S6, S7, OFF)                                          \
  "ds_read_b64_tr_b16 v[" S0 "], %4 offset:" TN_STR(OFF) "\n\t"                              \
  "ds_read_b64_tr_b16 v[" S1 "], %4 offset:" TN_STR(OFF) "+256\n\t"                          \
  "ds_read_b64_tr_b16 v[" S4 "], %6 offset:" TN_STR(OFF) "\n\t"                              \
  "ds_read_b64_tr_b16 v[" S5 "], %6 offset:" TN_STR(OFF) "+256\n\t"                          \
  "ds_read_b64_tr_b16 v[" S2 "], %5 offset:" TN_STR(OFF) "\n\t"                              \
  "ds_read_b64_tr_b16 v[" S3 "], %5 offset:" TN_STR(OFF) "+256\n\t"                          \
  "ds_read_b64_tr_b16 v[" S6 "], %7 offset:" TN_STR(OFF) "\n\t"                              \
  "ds_read_b64_tr_b16 v[" S7 "], %7 offset:" TN_STR(OFF) "+256\n\t"
#define TN_RD_A(OFF) TN_RD("160:161", "162:163", "164:165", "166:167", "168:169", "170:171", "172:173", "174:175", OFF)
#define TN_RD_B(OFF) TN_RD("176:177", "178:179", "180:181", "182:183", "184:185", "186:187", "188:189", "190:191", OFF)
#define TN_MM(D0, D1, X0, X1)                                            \
  "v_mfma_f32_32x32x16_bf16 %0, v[" D0 "], v[" X0 "], %0\n\t"             \
  "v_mfma_f32_32x32x16_bf16 %1, v[" D0 "], v[" X1 "], %1\n\t"             \
  "v_mfma_f32_32x32x16_bf16 %2, v[" D1 "], v[" X0 "], %2\n\t"             \
  "v_mfma_f32_32x32x16_bf16 %3, v[" D1 "], v[" X1 "], %3\n\t"
#define TN_MM_A TN_MM("160:163", "164:167", "168:171", "172:175")
#define TN_MM_B TN_MM("176:179", "180:183", "184:187", "188:191")
#define TN_CLOBBERS                                                                                                      \
  "memory", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", \
      "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", \
      "v189", "v190", "v191"

// a0, a1: LDS byte addresses of this lane's piece in the wave's two dY feature blocks; b0, b1: in its two X blocks
template <int RST>
__device__ __forceinline__ void tn_stage(f32x16 &c00, f32x16 &c01, f32x16 &c10, f32x16 &c11, uint32_t a0, uint32_t a1,
                                         uint32_t b0, uint32_t b1) {
  if constexpr (RST == 32) {
    asm volatile(TN_RD_A(0) TN_RD_B(1024)
                 "s_waitcnt lgkmcnt(8)\n\t" TN_MM_A
                 "s_waitcnt lgkmcnt(0)\n\t" TN_MM_B
                 : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11)
                 : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
                 : TN_CLOBBERS);
  } else {
    asm volatile(TN_RD_A(0) TN_RD_B(1024)
                 "s_waitcnt lgkmcnt(8)\n\t" TN_MM_A TN_RD_A(2048)
                 "s_waitcnt lgkmcnt(8)\n\t" TN_MM_B TN_RD_B(3072)
                 "s_waitcnt lgkmcnt(8)\n\t" TN_MM_A
                 "s_waitcnt lgkmcnt(0)\n\t" TN_MM_B
                 : "+v"(c00), "+v"(c01), "+v"(c10), "+v"(c11)
                 : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
                 : TN_CLOBBERS);
  }
}

template <int WGA, int WGB, int RST, int NST>
__global__ __launch_bounds__(64 * WGA * WGB) void k_gemm_bf16_tn(const TnArgs g) {
  constexpr int NW = WGA * WGB;
  constexpr int TA = 64 * WGA, TB = 64 * WGB;
  constexpr int BLK = RST * 64;                 // bytes of one [RST tokens][32 features] block
  constexpr int NBLK = (TA + TB) / 32;
  constexpr int STAGE = NBLK * BLK;
  constexpr int UPB = RST / 16;                 // 1 KB DMA units per block
  constexpr int IPW = NBLK * UPB / NW;
  static_assert((NBLK * UPB) % NW == 0 && NST >= 3 && NST <= 4 && (RST == 32 || RST == 64), "tile");
  static_assert(IPW * (NST - 2) < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wa = wave % WGA, wb = wave / WGA;
  // One linear grid over (split, tile), split-major: consecutive work items are the tiles of ONE token range, and an XCD
  // (workgroup id & 7) takes a contiguous run of them — the b-tiles that share a dY column block then share that XCD's
  // L2.  (With the split on blockIdx.y the hardware's id -> XCD mapping no longer matched the remap for y > 0 and dY
  // came from memory once per b-tile: 446 MB against 197 MB algorithmic, profiles/r04_pmc_traffic.json before the change.)
  const int ntile = g.tiles_a * g.tiles_b;
  int work;
  {
    const int total = ntile * g.nsplit;
    const int id = blockIdx.x, xcd = id & 7, q = total >> 3, r = total & 7;
    work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int split = work / ntile;
  const int tile = work - split * ntile;
  const int ta = tile / g.tiles_b, tb = tile - ta * g.tiles_b;
  const int a0 = ta * TA, b0 = tb * TB;
  const int sb = split * g.per_split;
  const int se = min(g.stages, sb + g.per_split);
  const int nk = se - sb;

  // per-lane source pointers and strides, fixed before the loop (see the note on `src` in k_gemm_bf16_nt_r)
  const uint16_t *zp = reinterpret_cast<const uint16_t *>(&g_tn_zero);
  const uint16_t *src[IPW];
  int adv[IPW];
  const int tok0 = sb * RST + (lane >> 2);      // + 16 * (unit's token group)
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int u = wave * IPW + i;
    const int blk = u / UPB, tg = u % UPB;
    const bool isa = blk < TA / 32;
    const int col = isa ? a0 + blk * 32 : b0 + (blk - TA / 32) * 32;
    const int ld = isa ? g.Na : g.Nb;
    const bool ok = col < ld;
    const uint16_t *base = isa ? g.dy : g.x;
    src[i] = ok ? base + (size_t)(tok0 + 16 * tg) * ld + col + (lane & 3) * 8 : zp;
    adv[i] = ok ? RST * ld : 0;
  }
  int left[UPB];                                // tokens of this lane's row still inside M, per token group
#pragma unroll
  for (int t = 0; t < UPB; ++t) left[t] = g.M - (tok0 + 16 * t);

  auto issue = [&](int s, int buf) {            // s: stage index relative to sb
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int u = wave * IPW + i;
      lds_ptr_t d = (lds_ptr_t)(lds + buf * STAGE + u * 1024);
      const uint16_t *q = src[i] + (size_t)s * adv[i];
      const uint16_t *p = (left[u % UPB] > s * RST) ? q : zp;
      __builtin_amdgcn_global_load_lds(p, d, 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // transposing reads: 16-lane group gq, lane sl -> tokens 8*(gq>>1) + (sl>>2) (and +4), features 16*(gq&1) + 4*(sl&3)
  const int gq = lane >> 4, sl = lane & 15;
  const uint32_t frag = (uint32_t)((8 * (gq >> 1) + (sl >> 2)) * 64 + (16 * (gq & 1) + 4 * (sl & 3)) * 2);
  const uint32_t a_off = (uint32_t)(wa * 2 * BLK) + frag;
  const uint32_t b_off = (uint32_t)((TA / 32 + wb * 2) * BLK) + frag;

  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_ptr_t)lds;
  auto compute = [&](int buf) {
    const uint32_t base = lds0 + (uint32_t)(buf * STAGE);
    tn_stage<RST>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], base + a_off, base + a_off + BLK, base + b_off, base + b_off + BLK);
  };

#pragma unroll
  for (int s0 = 0; s0 < NST - 1; ++s0)
    if (s0 < nk) issue(s0, s0);
  int buf = 0, nbuf = NST - 1;
  for (int f = 0; f < nk; ++f) {
    const int later = nk - 1 - f;
    if (later >= NST - 2) GB_WAIT_VM(IPW * (NST - 2));
    else if (NST == 4 && later == 1) GB_WAIT_VM(IPW);
    else GB_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (f + NST - 1 < nk) issue(f + NST - 1, nbuf);
    compute(buf);
    buf = (buf + 1 == NST) ? 0 : buf + 1;
    nbuf = (nbuf + 1 == NST) ? 0 : nbuf + 1;
  }

  // D[i = dY feature][j = X feature]: lane (j = lane & 31, half hi), register v: i = (v & 3) + 8 (v >> 2) + 4 hi.
  // 32 lanes write 128 contiguous bytes of one dW row.
  const int lo = lane & 31, hi = lane >> 5;
  float *out = g.out + (size_t)split * g.Na * g.Nb;
  const bool add = g.accumulate != 0;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int c = b0 + wb * 64 + b * 32 + lo;
    if (c >= g.Nb) continue;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int nb = a0 + wa * 64 + a * 32 + 4 * hi;
      float old[16];  // `+=`: all sixteen old values are read before the first store (no load waits behind a store),
      if (add) {      // unconditionally: rows past Na re-read the last row
#pragma unroll
        for (int v = 0; v < 16; ++v) old[v] = out[(size_t)min(nb + (v & 3) + 8 * (v >> 2), g.Na - 1) * g.Nb + c];
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = nb + (v & 3) + 8 * (v >> 2);
        if (n < g.Na) out[(size_t)n * g.Nb + c] = add ? old[v] + acc[a][b][v] : acc[a][b][v];
      }
    }
  }
}

// dw[i] (+)= sum_z part[z][i], z in index order; 4 elements per thread
__global__ __launch_bounds__(256) void k_tn_reduce(const float *__restrict__ part, float *__restrict__ dw, int64_t n4, int64_t n,
                                                   int splits, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 s = accumulate ? *reinterpret_cast<const float4 *>(dw + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
      const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)z * n + 4 * i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4 *>(dw + 4 * i) = s;
  }
}

#ifndef SALUN_TN_TARGET
#define SALUN_TN_TARGET 256  // workgroups a launch is split up to: 512 until backward-weight moved to the side stream (profiles/r06_sd_split_targets.txt)
#endif
struct TnPlan { int variant, ta, tb, rst, tiles_a, tiles_b, stages, splits, per_split; };

// variant 1: 128 x 128, 64-token stages x 3 (96 KB, one workgroup per CU); 2: 128 x 128, 32-token stages x 4 (64 KB, two
// per CU); 3: 256 x 128, 8 waves, 64-token stages x 3 (144 KB)
TnPlan tn_plan(int64_t M, int Na, int Nb, int variant) {
  TnPlan p;
  if (variant == 0) variant = 2;
  p.variant = variant;
  p.ta = variant == 3 ? 256 : 128; p.tb = 128;
  p.rst = variant == 2 ? 32 : 64;
  p.tiles_a = (Na + p.ta - 1) / p.ta; p.tiles_b = (Nb + p.tb - 1) / p.tb;
  p.stages = (int)((M + p.rst - 1) / p.rst);
  const int tiles = p.tiles_a * p.tiles_b;
  const int min_stages = 256 / p.rst;           // a split reduces over >= 256 tokens
  int splits = (SALUN_TN_TARGET + tiles - 1) / tiles;
  if (splits > p.stages / min_stages) splits = p.stages / min_stages;
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  p.per_split = (p.stages + splits - 1) / splits;
  p.splits = (p.stages + p.per_split - 1) / p.per_split;
  return p;
}

template <int WGA, int WGB, int RST, int NST>
int launch_gemm_bf16_tn(const TnArgs &g, int splits, hipStream_t st) {
  const size_t ldsb = (size_t)NST * (64 * WGA + 64 * WGB) / 32 * RST * 64;
  static unsigned long long configured = 0;  // one bit per device: the opt-in is per device, not per process
  if (salun_once_needed(&configured)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bf16_tn<WGA, WGB, RST, NST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    salun_once_mark(&configured);
  }
  hipLaunchKernelGGL((k_gemm_bf16_tn<WGA, WGB, RST, NST>), dim3(g.tiles_a * g.tiles_b * splits), dim3(64 * WGA * WGB), ldsb, st, g);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

// fp32 [N][K] -> bf16 [K][N] (the weight image the input-gradient GEMM reads); 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_pack_bf16_t(const float *__restrict__ w, uint16_t *__restrict__ wt, int N, int K) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = n0 + ty + 8 * r, k = k0 + tx;
    tile[ty + 8 * r][tx] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = k0 + ty + 8 * r, n = n0 + tx;
    if (k < K && n < N) wt[(size_t)k * N + n] = __builtin_bit_cast(uint16_t, (__bf16)tile[tx][ty + 8 * r]);
  }
}
__global__ __launch_bounds__(256) void k_pack_bf16(const float *__restrict__ w, uint16_t *__restrict__ wp, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4 *>(w + i);
      *reinterpret_cast<uint2 *>(wp + i) = make_uint2(gb_pack2(v.x, v.y), gb_pack2(v.z, v.w));
    } else {
      for (int64_t j2 = i; j2 < n; ++j2) wp[j2] = __builtin_bit_cast(uint16_t, (__bf16)w[j2]);
    }
  }
}

template <int WGM, int WGN, bool DB>
int launch_gemm_bf16(const GbArgs &g0, hipStream_t st) {
  GbArgs g = g0;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  g.tiles_m = (g.M + BMt - 1) / BMt;
  g.tiles_n = g.N / BNt;
  const size_t ldsb = (size_t)(DB ? 2 : 1) * (BMt + BNt) * 128;
  static unsigned long long configured = 0;  // one bit per device: the opt-in is per device, not per process
  if (salun_once_needed(&configured)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bf16_nt<WGM, WGN, DB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    salun_once_mark(&configured);
  }
  hipLaunchKernelGGL((k_gemm_bf16_nt<WGM, WGN, DB>), dim3(g.tiles_m * g.tiles_n), dim3(64 * WGM * WGN), ldsb, st, g);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

template <int WGM, int WGN>
int launch_gemm_bf16_p(const GbArgs &g0, hipStream_t st) {
  GbArgs g = g0;
  constexpr int BMt = 64 * WGM, BNt = 64 * WGN;
  g.tiles_m = (g.M + BMt - 1) / BMt;
  g.tiles_n = g.N / BNt;
  const size_t ldsb = (size_t)2 * (BMt + BNt) * 128;
  static unsigned long long configured = 0;  // one bit per device: the opt-in is per device, not per process
  if (salun_once_needed(&configured)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_bf16_nt_p<WGM, WGN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    salun_once_mark(&configured);
  }
  const int ntile = g.tiles_m * g.tiles_n;
  // 2 workgroups per CU (LDS: 2 x 64-80 KB); each walks a run of tiles inside its XCD's contiguous range
  const int per_xcd = (ntile + 7) / 8;
  int runs = 64;                                   // workgroups per XCD
  if (runs > per_xcd) runs = per_xcd;
  const int tiles_per_wg = (per_xcd + runs - 1) / runs;
  runs = (per_xcd + tiles_per_wg - 1) / tiles_per_wg;
  hipLaunchKernelGGL((k_gemm_bf16_nt_p<WGM, WGN>), dim3(runs * 8), dim3(64 * WGM * WGN), ldsb, st, g, tiles_per_wg);
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}

}  // namespace

// Internal (not exported): salun_conv2d_bf16_forward (salun_conv_bf16.hip) tries this first.  Returns 1 when the ring
// kernel took the convolution, 0 when the shape is left to the register-staged kernel (few tiles: that one can split the
// reduction; channel counts that do not tile), a negative error otherwise.
extern "C" int salun_conv2d_bf16_forward_ring(const uint16_t *x, const uint16_t *wp, const float *bias, const float *nbias,
                                              const uint16_t *addend, uint16_t *y, int N, int H, int W, int C, int K, int R,
                                              int stride, int pad, hipStream_t st) {
  static const int on = [] { const char *e = getenv("SALUN_CONV_RING"); return e ? atoi(e) : 1; }();
  const int OH = (H + 2 * pad - R) / stride + 1, OW = (W + 2 * pad - R) / stride + 1;
  const int64_t M = (int64_t)N * OH * OW;
  if (!on || C % 32 != 0 || K % 64 != 0 || (R != 1 && R != 3) || M >= (1 << 30)) return 0;
  // Measured on the SD layer table (tools/convbench_bf16.py, round 4): the ring wins the 1 x 1 convolutions (299 vs 229 and
  // 282 vs 188 TFLOP/s) and LOSES the 3 x 3 ones (442 vs 540 TFLOP/s over the table: 90 .. 720 stages of 8 MFMAs per wave,
  // gathered 64-byte rows through the LDS-DMA path, which the register-staged kernel with its 128 x 256 tiles out-runs).
  // on == 3 forces it for every eligible shape (tests, A/B).
  if (R != 1 && on != 3 && on != 2) return 0;
  if ((bias && !salun_aligned16(bias)) || (nbias && !salun_aligned16(nbias)) || (addend && !salun_aligned16(addend)) ||
      !salun_aligned16(y))
    return 0;
  IrArgs g{x, wp, bias, nbias, addend, y, (int)M, H, W, C, OH, OW, K, R, stride, pad, 0, 0};
  const int64_t t256 = (M + 255) / 256;
  if (on == 2 && K % 128 == 0) return launch_conv_ring<2, 2, 4>(g, st) == SALUN_OK ? 1 : SALUN_EIO;   // A/B: pin the 128 x 128 form
  if (K % 128 == 0 && t256 * (K / 128) >= 128) return launch_conv_ring<4, 2, 3>(g, st) == SALUN_OK ? 1 : SALUN_EIO;
  if (K % 128 != 0 && t256 * (K / 64) >= 128) return launch_conv_ring<4, 1, 3>(g, st) == SALUN_OK ? 1 : SALUN_EIO;
  if (K % 128 == 0 && ((M + 127) / 128) * (K / 128) >= 128) return launch_conv_ring<2, 2, 4>(g, st) == SALUN_OK ? 1 : SALUN_EIO;
  return 0;
}

SALUN_EXPORT int salun_gemm_bf16_supported(int64_t M, int N, int K) {
  return (M >= 1 && N >= 64 && K >= 64 && N % 64 == 0 && K % 64 == 0 && M * (int64_t)K < (1ll << 31) &&
          (int64_t)N * K < (1ll << 31)) ? 1 : 0;
}

// variant: 0 = choose; 1 = 128 tokens x 128 features double-buffered, 2 = same single-buffered, 3 = 256 tokens x 64
// features double-buffered, 4 = same single-buffered, 5 / 6 = the persistent forms of 1 / 3 (A/B measurements:
// tools/gemmbench_bf16.py)
SALUN_EXPORT int salun_gemm_bf16_nt(const void *x, const void *w, const float *bias, const void *addend, void *y, int64_t M,
                                    int N, int K, int variant, salun_stream_t stream) {
  if (!salun_gemm_bf16_supported(M, N, K) || !x || !w || !y) return SALUN_EINVAL;
  if (!salun_aligned16(x) || !salun_aligned16(w) || !salun_aligned16(y) || (bias && !salun_aligned16(bias)) ||
      (addend && !salun_aligned16(addend)))
    return SALUN_EINVAL;
  GbArgs g;
  g.x = static_cast<const uint16_t *>(x); g.w = static_cast<const uint16_t *>(w); g.bias = bias;
  g.addend = static_cast<const uint16_t *>(addend); g.y = static_cast<uint16_t *>(y);
  g.M = (int)M; g.N = N; g.K = K; g.tiles_m = g.tiles_n = 0;
  hipStream_t st = salun_hip_stream(stream);
  if (variant == 0) {
    // measured on the SD shapes (profiles/r04_gemmbench_bf16.txt): the 8-wave 256 x 128 ring with 32-deep stages (two
    // workgroups per CU) wins wherever it has >= ~128 tiles; the 4-wave 128 x 128 ring on smaller problems; feature counts
    // that are multiples of 64 only: 256 x 64, 32-deep
    if (N % 128 != 0) variant = 13;
    else variant = (((M + 255) / 256) * (int64_t)(N / 128) >= 128) ? 11 : 10;
  }
  if ((variant == 1 || variant == 2) && N % 128 != 0) return SALUN_EINVAL;
  switch (variant) {
    case 1: return launch_gemm_bf16<2, 2, true>(g, st);
    case 2: return launch_gemm_bf16<2, 2, false>(g, st);
    case 3: return launch_gemm_bf16<4, 1, true>(g, st);
    case 4: return launch_gemm_bf16<4, 1, false>(g, st);
    case 5: return (N % 128 == 0) ? launch_gemm_bf16_p<2, 2>(g, st) : SALUN_EINVAL;
    case 6: return launch_gemm_bf16_p<4, 1>(g, st);
    case 7: return (N % 128 == 0) ? launch_gemm_bf16_r<2, 2, 4>(g, st) : SALUN_EINVAL;   // 128 x 128, 4 stages (128 KB)
    case 8: return (N % 128 == 0) ? launch_gemm_bf16_r<4, 2, 3>(g, st) : SALUN_EINVAL;   // 256 x 128, 8 waves, 3 stages (144 KB)
    case 9: return launch_gemm_bf16_r<4, 1, 3>(g, st);                                   // 256 x 64, 3 stages (120 KB)
    case 10: return (N % 128 == 0) ? launch_gemm_bf16_r<2, 2, 3>(g, st) : SALUN_EINVAL;  // 128 x 128, 3 stages (96 KB)
    case 11: return (N % 128 == 0) ? launch_gemm_bf16_r<4, 2, 3, 32>(g, st) : SALUN_EINVAL;  // 256 x 128, 32-deep stages (72 KB: two per CU)
    case 12: return (N % 128 == 0) ? launch_gemm_bf16_r<2, 2, 4, 32>(g, st) : SALUN_EINVAL;  // 128 x 128, 32-deep x 4 (64 KB: two per CU)
    case 13: return launch_gemm_bf16_r<4, 1, 3, 32>(g, st);                                  // 256 x 64, 32-deep (60 KB: two per CU)
    default: return SALUN_EINVAL;
  }
}

// dW fp32 [Na][Nb] (+= when accumulate) = dY^T . X over M tokens; dY bf16 [M][Na], X bf16 [M][Nb].  Na, Nb multiples of
// 32.  `ws` holds the split partials (salun_gemm_bf16_tn_workspace_bytes; may be null when that is 0).
SALUN_EXPORT int salun_gemm_bf16_tn_supported(int64_t M, int Na, int Nb) {
  return M >= 1 && M < (int64_t(1) << 24) && Na >= 32 && Nb >= 32 && Na % 32 == 0 && Nb % 32 == 0 &&
         (int64_t)M * Na < (int64_t(1) << 31) && (int64_t)M * Nb < (int64_t(1) << 31);
}

SALUN_EXPORT size_t salun_gemm_bf16_tn_workspace_bytes(int64_t M, int Na, int Nb, int variant) {
  if (!salun_gemm_bf16_tn_supported(M, Na, Nb)) return 0;
  const TnPlan p = tn_plan(M, Na, Nb, variant);
  return p.splits > 1 ? (size_t)p.splits * Na * Nb * sizeof(float) : 0;
}

SALUN_EXPORT int salun_gemm_bf16_tn(const void *dy, const void *x, float *dw, int64_t M, int Na, int Nb, int accumulate,
                                    int variant, void *ws, size_t ws_bytes, salun_stream_t stream) {
  if (!salun_gemm_bf16_tn_supported(M, Na, Nb) || !dy || !x || !dw || variant < 0 || variant > 3) return SALUN_EINVAL;
  if (!salun_aligned16(dy) || !salun_aligned16(x) || !salun_aligned16(dw)) return SALUN_EINVAL;
  const TnPlan p = tn_plan(M, Na, Nb, variant);
  const size_t need = p.splits > 1 ? (size_t)p.splits * Na * Nb * sizeof(float) : 0;
  if (need && (!ws || ws_bytes < need || !salun_aligned16(ws))) return SALUN_ENOSPC;
  TnArgs g;
  g.dy = static_cast<const uint16_t *>(dy); g.x = static_cast<const uint16_t *>(x);
  g.out = p.splits > 1 ? static_cast<float *>(ws) : dw;
  g.M = (int)M; g.Na = Na; g.Nb = Nb; g.tiles_a = p.tiles_a; g.tiles_b = p.tiles_b;
  g.stages = p.stages; g.per_split = p.per_split; g.nsplit = p.splits; g.accumulate = (p.splits == 1 && accumulate) ? 1 : 0;
  hipStream_t st = salun_hip_stream(stream);
  int rc;
  if (p.variant == 1) rc = launch_gemm_bf16_tn<2, 2, 64, 3>(g, p.splits, st);
  else if (p.variant == 2) rc = launch_gemm_bf16_tn<2, 2, 32, 4>(g, p.splits, st);
  else rc = launch_gemm_bf16_tn<4, 2, 64, 3>(g, p.splits, st);
  if (rc != SALUN_OK) return rc;
  if (p.splits > 1) {
    const int64_t n = (int64_t)Na * Nb, n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_tn_reduce, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float *>(ws), dw, n4, n, p.splits,
                       accumulate);
    SALUN_LAUNCH_CHECK();
  }
  return SALUN_OK;
}

SALUN_EXPORT int salun_pack_bf16(const float *w, void *wp, int N, int K, int transposed, salun_stream_t stream) {
  if (N < 1 || K < 1 || !w || !wp) return SALUN_EINVAL;
  hipStream_t st = salun_hip_stream(stream);
  if (transposed) {
    hipLaunchKernelGGL(k_pack_bf16_t, dim3((K + 31) / 32, (N + 31) / 32), dim3(256), 0, st, w, static_cast<uint16_t *>(wp), N, K);
  } else {
    if (!salun_aligned16(w) || !salun_aligned16(wp)) return SALUN_EINVAL;
    const int64_t n = (int64_t)N * K;
    hipLaunchKernelGGL(k_pack_bf16, dim3(salun_grid_for(n, 1024)), dim3(256), 0, st, w, static_cast<uint16_t *>(wp), n);
  }
  SALUN_LAUNCH_CHECK();
  return SALUN_OK;
}
