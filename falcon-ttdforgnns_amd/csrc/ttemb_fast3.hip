// Sorted / grouped MFMA path for 3-core tables (the configuration every driver of the
// reference uses: T == 3).
//
// Idea (prefix reuse, cf. the reference's unwired Efficient_TT/efficient_tt_cuda.cu:159-377,
// done here without its global prefix cache, pointer arrays or host round trips):
//   * ids are radix-sorted once per call, so ids that share (i0, i1) -- a "group" -- sit
//     next to each other;
//   * one wavefront walks a contiguous range of the sorted ids in chunks of <= 16 ids of one
//     group.  Per group it forms the prefix product P = G0[i0] . G1[i1]  (q0 x q1 r2) with
//     fp32 MFMA (v_mfma_f32_16x16x4_f32) and keeps it in LDS; per chunk it multiplies P
//     (as a q0q1 x r2 matrix) with the chunk's stacked G2 rows (r2 x 16 q2) -- again fp32
//     MFMA -- and writes whole D-float rows with 16-byte stores.
// Stage 1 is therefore paid once per group instead of once per id, stage 2 runs as a real
// GEMM (M = q0q1, K = r2, N = 16 q2), and partial products never touch HBM.
//
// fp32 MFMA is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md §3), so results
// agree with the generic kernel / the reference's fp32 GEMMs to rounding.
#include "ttemb_common.h"
#include "ttemb_cache.h"

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>


namespace ttemb {

typedef float f32x4 __attribute__((ext_vector_type(4)));


constexpr int kChunk = 16;        // ids per stage-2 GEMM (N = 16 * q2 columns)
constexpr uint32_t kMultiBit = 0x80000000u;

template <int Q0, int Q1, int Q2, int R1, int R2>
struct Cfg {
  static constexpr int M2 = Q0 * Q1;             // rows of the stage-2 GEMM
  static constexpr int MT2 = (M2 + 15) / 16;     // 16-row MFMA tiles of it
  static constexpr int KS1 = R1 / 4;             // k-steps (K = 4 per MFMA)
  static constexpr int KS2 = R2 / 4;
  static constexpr int N1 = Q1 * R2;             // columns of the prefix product
  static constexpr int NT1 = N1 / 16;
  static constexpr int NT2 = Q2;                 // (16 ids * Q2 columns) / 16
  static constexpr int D = Q0 * Q1 * Q2;
  static constexpr int ROW0 = Q0 * R1;           // floats per core row
  static constexpr int ROW1 = R1 * Q1 * R2;
  static constexpr int ROW2 = R2 * Q2;
  static constexpr int LDA = R2 + 1;             // P rows padded: conflict-free A-operand reads
  static constexpr int LDB = ROW2 + 4;           // staged G2 rows (16-byte aligned rows)
  static constexpr int LDO = D + 4;              // staged output rows
  static constexpr int P_FLOATS = ((MT2 * 16 * LDA + 3) / 4) * 4;
  static constexpr int B_FLOATS = kChunk * LDB;
  static constexpr int O_FLOATS = kChunk * LDO;
  // the output rows reuse the staged-G2 region (every G2 read precedes every row write)
  // backward chunk kernel: ds_read_b32 banks are (addr/4) mod 32 per 32-lane half (lane groups hi = {0,1} and
  // {2,3}), so the stride between the two `hi` rows of an operand must be 16 mod 32 for a conflict-free read
  static constexpr int LDPB = (R2 % 32 == 0) ? R2 + 16 : R2;                            // P rows, read as [4s+hi][lo]
  static constexpr int LDBB = (Q2 % 2 == 1 && ROW2 % 32 == 16) ? ROW2 : ROW2 + 4;       // G2 rows, read as [hi][lo*q2+kk]
  static constexpr int BB2_FLOATS = kChunk * LDBB;
  static constexpr int LDOB = ((D + 15) / 32) * 32 + 16;                                // d_output rows, read as [hi][m*q2+kk]
  static constexpr int OB_FLOATS = kChunk * LDOB;
  static constexpr int PB_FLOATS = ((M2 * LDPB + 3) / 4) * 4;  // backward reads only the M2 real rows of P
  static constexpr int BO_FLOATS = B_FLOATS > O_FLOATS ? B_FLOATS : O_FLOATS;
  static constexpr int WAVE_FLOATS = P_FLOATS + BO_FLOATS;
  // backward: [P | staged G2 rows, later dP | staged d_output rows, later G1[i1]]
  static constexpr int RT1 = (R1 + 15) / 16;     // 16-wide tiles over the ranks
  static constexpr int RT2 = (R2 + 15) / 16;
  static constexpr int LDG = N1 + 1;             // staged G1 row stride (conflict-free column reads)
  static constexpr int LD2 = ROW2 + 1;           // row stride of the LDS dG2 accumulator (spreads banks)
  static constexpr int BB_FLOATS = B_FLOATS > P_FLOATS ? B_FLOATS : P_FLOATS;
  static constexpr int DB_FLOATS = ((O_FLOATS > R1 * LDG ? O_FLOATS : R1 * LDG) + 3) / 4 * 4;
  static constexpr int BWD_WAVE_FLOATS = P_FLOATS + BB_FLOATS + DB_FLOATS;
  static_assert(M2 % 4 == 0, "q0*q1 must be a multiple of the MFMA K");
  static_assert(Q0 <= 16, "stage 1 pads q0 to one 16-row tile");
  static_assert(R1 % 4 == 0 && R2 % 4 == 0, "ranks must be multiples of the MFMA K");
  static_assert(N1 % 16 == 0, "q1*r2 must tile by 16");
  static_assert(D % 4 == 0 && ROW2 % 4 == 0, "rows move as float4");
};

// ---------------------------------------------------------------------------------
// Grouping pass: a counting sort of the live ids by group' = i1 * p0 + i0, cut into chunks
// (three small kernels; rocprim's radix/merge sort needs ~20 launches and > 100 us at these sizes).
//   key   = (i1 * p0 + i0) * p2 + i2   -- the id with its digits reordered: ids of one
//           (i0, i1) group end up adjacent, and consecutive groups share i1;
//   value = output row | kMultiBit when the bag holds several ids.
// One returning atomic per id (its arrival rank inside the group) in the first kernel, ONE rocPRIM
// exclusive scan that carries two running sums in a 64-bit word (ids before the group | chunks before
// the group), and an atomic-free scatter.  The scatter also writes the *chunk table*: a chunk is <= 16
// consecutive ids of one group, and its 16-byte descriptor {position, group, length | flags, first chunk
// of the next group} is all the chain kernels need to walk the grouped ids -- they read descriptors with
// scalar loads and never decode a key, compare neighbours or shuffle.  The order of ids inside a
// group is arrival order: every consumer is insensitive to it except for fp32 summation order in
// the backward.
// ---------------------------------------------------------------------------------
constexpr int kTile = 256;   // threads per workgroup in the grouping kernels
constexpr uint32_t kFirstBit = 0x100u, kLastBit = 0x200u;   // flags next to a chunk's length

struct GroupPlan {           // device pointers into the caller's plan buffer / workspace
  uint32_t* keys_in;         // [nnz] ungrouped keys
  uint32_t* vals_in;
  uint32_t* rank_in;         // arrival rank of the id inside its group
  uint32_t* i2s;             // [nnz] grouped: last index digit of the id
  uint32_t* vals;            // [nnz] grouped: output row | kMultiBit
  uint32_t* counts;          // [G+1] ids per group (entry G stays 0)
  uint64_t* gpre;            // [G+1] low word: first grouped position of the group; high word: its first chunk.
                             //       entry G = (live ids, chunks)
  uint4* ctab;               // [max_chunks] chunk descriptors
  float* ptab;               // [G][M2*R2] prefix product P = G0[i0] . G1[i1] of every non-empty group
  float* etab;               // [nnz][ROW2] dG2 contribution rows, in grouped order
  float* dptab;              // [G][M2*R2] dP of every non-empty group
  float* g2part;             // [tiles][p2][ROW2] per-tile partial dG2
  float* g0part;             // [G][ROW0] per-group contribution to dG0
};

__global__ __launch_bounds__(kTile) void fast3_prep_kernel(
    const int64_t* __restrict__ indices, const int64_t* __restrict__ rowidx,
    const int64_t* __restrict__ offsets, int64_t nnz,
    const int32_t* __restrict__ nnz_dev, uint32_t sentinel, uint32_t p0, uint32_t p1, uint32_t p2,
    GroupPlan plan) {
  const int64_t n = (int64_t)blockIdx.x * kTile + threadIdx.x;
  const int64_t cnt = live_count(nnz, nnz_dev);
  if (n >= cnt) return;
  int64_t id = indices[n];
  id = id < 0 ? 0 : (id >= (int64_t)sentinel ? (int64_t)sentinel - 1 : id);
  const int64_t row = rowidx[n];
  const bool multi = !bag_is_single(rowidx, offsets, n, cnt, row);
  const uint32_t u = (uint32_t)id;
  const uint32_t i0 = u / (p1 * p2);
  const uint32_t rem = u - i0 * (p1 * p2);
  const uint32_t i1 = rem / p2;
  const uint32_t i2 = rem - i1 * p2;
  const uint32_t group = i1 * p0 + i0;
  plan.keys_in[n] = group * p2 + i2;
  plan.vals_in[n] = (uint32_t)row | (multi ? kMultiBit : 0u);
  plan.rank_in[n] = atomicAdd(&plan.counts[group], 1u);
}

// what the scan adds up per group: ids in the low word, chunks of <= kChunk ids in the high word
struct PackCounts {
  __host__ __device__ uint64_t operator()(uint32_t c) const {
    return (uint64_t)c | ((uint64_t)((c + kChunk - 1) / kChunk) << 32);
  }
};

__global__ __launch_bounds__(kTile) void fast3_scatter_kernel(int64_t nnz, const int32_t* __restrict__ nnz_dev,
                                                              uint32_t p2, GroupPlan plan) {
  const int64_t n = (int64_t)blockIdx.x * kTile + threadIdx.x;
  if (n >= live_count(nnz, nnz_dev)) return;
  const uint32_t key = plan.keys_in[n];
  const uint32_t g = key / p2;
  const uint32_t rank = plan.rank_in[n];
  const uint64_t pre = plan.gpre[g];
  const uint32_t dst = (uint32_t)pre + rank;
  plan.i2s[dst] = key - g * p2;
  plan.vals[dst] = plan.vals_in[n];
  if (rank % kChunk == 0) {  // this id opens a chunk of its group: it writes the descriptor
    const uint32_t c = plan.counts[g];
    const uint32_t chunks = (c + kChunk - 1) / kChunk, k = rank / kChunk;
    const uint32_t first_chunk = (uint32_t)(pre >> 32);
    const uint32_t len = c - rank < (uint32_t)kChunk ? c - rank : (uint32_t)kChunk;
    plan.ctab[first_chunk + k] = make_uint4(dst, g, len | (k == 0 ? kFirstBit : 0u) | (k + 1 == chunks ? kLastBit : 0u),
                                            first_chunk + chunks);
  }
}

// ---------------------------------------------------------------------------------
// Prefix products: P[g] = G0[i0] . G1[i1]  (q0 x q1 r2, kept as a (q0 q1) x r2 matrix) for every
// non-empty group, once per call, into a table the chain kernels read like any other operand.
// (Computing P inside the chain kernels put ~24 dependent-latency core-row loads, 20 MFMAs and an
// LDS round trip on the critical path of every group: 22 us of an 80 us forward.)  One wavefront
// holds G1[i1] as MFMA B operands in registers and walks kPrefixGroups values of i0, 16/q0 groups
// per MFMA tile (the tile's 16 rows are the q0 rows of those groups).
// ---------------------------------------------------------------------------------
constexpr int kPrefixGroups = 32;
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_prefix_kernel(const float* __restrict__ G0, const float* __restrict__ G1,
                                                          uint32_t p0, GroupPlan plan) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  constexpr int GM = 16 / Q0;  // groups per MFMA tile
  static_assert(16 % Q0 == 0, "q0 must divide the MFMA tile height");
  const int lane = threadIdx.x, hi = lane >> 4, lo = lane & 15;
  const uint32_t i1 = blockIdx.y;
  const uint32_t i0_begin = blockIdx.x * kPrefixGroups;
  const uint32_t i0_end = i0_begin + kPrefixGroups < p0 ? i0_begin + kPrefixGroups : p0;
  // any work at all?  (one lane per i0 of the slice)
  const uint32_t my = i0_begin + lane;
  const bool mine = lane < kPrefixGroups && my < i0_end && plan.counts[i1 * p0 + my] != 0;
  const unsigned long long live = __ballot(mine);
  if (!live) return;
  const float* g1 = G1 + (size_t)i1 * C::ROW1;
  float bv[C::KS1][C::NT1];
#pragma unroll
  for (int s = 0; s < C::KS1; ++s)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) bv[s][nt] = g1[(4 * s + hi) * C::N1 + 16 * nt + lo];
  for (uint32_t base = 0; base < (uint32_t)kPrefixGroups; base += GM) {
    if (!((live >> base) & ((1ull << GM) - 1))) continue;  // none of these groups holds an id
    const uint32_t i0a = i0_begin + base + lo / Q0;         // A operand: row lo = (group lo / q0, core row lo % q0)
    float av[C::KS1];
#pragma unroll
    for (int s = 0; s < C::KS1; ++s)
      av[s] = i0a < i0_end ? G0[(size_t)i0a * C::ROW0 + (lo % Q0) * R1 + 4 * s + hi] : 0.f;
    f32x4 acc[C::NT1];
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) {
      acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s][nt], acc[nt], 0, 0, 0);
    }
    // accumulator row 4 hi + r = (group (4 hi + r) / q0, core row a = (4 hi + r) % q0), column n = 16 nt + lo = (j, c2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * hi + r;
      const uint32_t gi = base + row / Q0;
      const int a = row % Q0;
      if (!((live >> gi) & 1ull)) continue;
      float* dst = plan.ptab + (size_t)(i1 * p0 + i0_begin + gi) * (C::M2 * R2);
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        dst[(a * Q1 + n / R2) * R2 + n % R2] = acc[nt][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// forward
//
// One wavefront takes kCPW consecutive chunk descriptors.  Everything it needs for a chunk is known two
// steps ahead, so the loop is a software pipeline with no data-dependent control flow: while chunk c is
// multiplied, the G2 rows (and, when c+1 opens a group, the prefix product) of chunk c+1 are in flight
// into registers and the (i2, row) pairs of chunk c+2 are being fetched.  Lanes are tied to ids four by
// four (lane = 4 * id + piece): a lane loads the pieces j, j+4, ... of "its" id's rows and later stores
// the same pieces of its output row, so no lane ever needs another lane's index.
// ---------------------------------------------------------------------------------
#ifndef TTEMB_CPW
#define TTEMB_CPW 6
#endif
constexpr int kCPW = TTEMB_CPW;   // chunks per wavefront

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_forward_kernel(const float* __restrict__ G2, GroupPlan plan, uint32_t G,
                                                           float* __restrict__ out) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, lo = lane & 15;
  const int b_l = lane >> 2, j_l = lane & 3;   // this lane's id inside a chunk, and which pieces of its rows
  float* pbuf = smem;
  float* bbuf = pbuf + C::P_FLOATS;
  float* obuf = bbuf;

  const uint32_t nchunks = (uint32_t)(plan.gpre[G] >> 32);
  const uint32_t c0 = blockIdx.x * kCPW;
  if (c0 >= nchunks) return;
  const uint32_t c1 = c0 + kCPW < nchunks ? c0 + kCPW : nchunks;

  constexpr int F4G = C::ROW2 / 4, NLG = (F4G + 3) / 4;     // float4 pieces of a G2 row / per lane
  constexpr int D4 = C::D / 4, NLO = (D4 + 3) / 4;          // float4 pieces of an output row / per lane
  constexpr int PF = C::M2 * R2, NLP = (PF + kWave - 1) / kWave;
  const uint4 none = make_uint4(0u, 0u, 0u, 0u);

  auto fetch_meta = [&](const uint4& d, uint32_t& i2, uint32_t& val) {
    i2 = 0u;  // row 0 stands in for unused slots
    val = 0u;
    if (b_l < (int)(d.z & 0xffu)) {
      i2 = plan.i2s[d.x + b_l];
      val = plan.vals[d.x + b_l];
    }
  };
  float4 pre_g[NLG];
  float pre_p[NLP];
  auto request = [&](const uint4& d, uint32_t i2, bool with_p) {
    const float* row = G2 + i2 * (uint32_t)C::ROW2;
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      pre_g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (F4G % 4 == 0 || idx < F4G) pre_g[k] = *reinterpret_cast<const float4*>(row + 4 * idx);
    }
    if (with_p) {
      const float* p = plan.ptab + (size_t)d.y * PF;
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;
        pre_p[it] = (PF % kWave == 0 || e < PF) ? p[e] : 0.f;
      }
    }
  };

  uint4 d_cur = plan.ctab[c0];
  uint4 d_nxt = c0 + 1 < c1 ? plan.ctab[c0 + 1] : none;
  uint32_t i2_cur, val_cur, i2_nxt = 0u, val_nxt = 0u;
  fetch_meta(d_cur, i2_cur, val_cur);
  request(d_cur, i2_cur, true);
  if (c0 + 1 < c1) fetch_meta(d_nxt, i2_nxt, val_nxt);

  for (uint32_t c = c0; c < c1; ++c) {
    const int len = (int)(d_cur.z & 0xffu);
    // ---- the prefix product of a new group -> LDS, as a (q0 q1) x r2 matrix ----
    if (c == c0 || (d_cur.z & kFirstBit)) {
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;
        if (PF % kWave == 0 || e < PF) pbuf[(e / R2) * C::LDA + e % R2] = pre_p[it];
      }
    }
    // ---- the chunk's G2 rows: registers -> LDS ----
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      if (F4G % 4 == 0 || idx < F4G) *reinterpret_cast<float4*>(bbuf + b_l * C::LDB + 4 * idx) = pre_g[k];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- next chunk's rows and the chunk after's indices go out now; they land while this one computes ----
    const uint32_t val = val_cur;
    const uint4 d_nn = c + 2 < c1 ? plan.ctab[c + 2] : none;
    if (c + 1 < c1) request(d_nxt, i2_nxt, (d_nxt.z & kFirstBit) != 0u);
    uint32_t i2_nn = 0u, val_nn = 0u;
    if (c + 2 < c1) fetch_meta(d_nn, i2_nn, val_nn);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);

    // ---- stage 2: (q0 q1 x r2) . (r2 x 16 q2), one 16-row tile of P at a time ----
    float bv[C::KS2][C::NT2];
#pragma unroll
    for (int s = 0; s < C::KS2; ++s)
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        const int n = 16 * nt + lo;
        bv[s][nt] = bbuf[(n / Q2) * C::LDB + (4 * s + hi) * Q2 + n % Q2];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every staged-G2 read is done: the region becomes the row buffer
#pragma unroll
    for (int mt = 0; mt < C::MT2; ++mt) {
      f32x4 acc[C::NT2];
#pragma unroll
      for (int s = 0; s < C::KS2; ++s) {
        const float a = pbuf[(16 * mt + lo) * C::LDA + 4 * s + hi];
#pragma unroll
        for (int nt = 0; nt < C::NT2; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[s][nt], s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[nt], 0, 0, 0);
      }
      // rows -> LDS, id-major
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        const int n = 16 * nt + lo;
        const int b = n / Q2, kk = n % Q2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mt + 4 * hi + r;
          if (m < C::M2) obuf[b * C::LDO + m * Q2 + kk] = acc[nt][r];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- 16-byte global stores: four lanes per row ----
    if (b_l < len) {
      float* dst = out + (val & ~kMultiBit) * (uint32_t)C::D;  // B*D < 2^32: checked on the host
#pragma unroll
      for (int k = 0; k < NLO; ++k) {
        const int idx = j_l + 4 * k;
        if (D4 % 4 == 0 || idx < D4) {
          const float4 x = *reinterpret_cast<const float4*>(obuf + b_l * C::LDO + 4 * idx);
          if (val & kMultiBit) {
            atomicAdd(dst + 4 * idx + 0, x.x);
            atomicAdd(dst + 4 * idx + 1, x.y);
            atomicAdd(dst + 4 * idx + 2, x.z);
            atomicAdd(dst + 4 * idx + 3, x.w);
          } else {
            *reinterpret_cast<float4*>(dst + 4 * idx) = x;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    d_cur = d_nxt;
    d_nxt = d_nn;
    val_cur = val_nxt;
    i2_nxt = i2_nn;
    val_nxt = val_nn;
  }
}

// ---------------------------------------------------------------------------------
// backward, atomics-free formulation (four kernels)
//
//  A. chunk kernel: per chunk
//        dP  += dO . G2s^T                        (q0q1 x r2, accumulated over the chunks of a group)
//        E    = P^T . dO   -> one (r2 q2)-float row per id, stored at the id's grouped position
//     and per group dP is stored in its own slot.  Plain stores only.
//  B. dG2[i2] = sum of the E rows whose id has that i2 (tile-local bucket sums, per-tile slabs).
//  C. group epilogue: per non-empty group (groups are (i1, i0)-ordered)
//        dG1[i1] += G0[i0]^T . dP  (registers while i1 repeats),  dG0[i0] += dP . G1[i1]^T.
//  D. finalize: dG2 = sum of slabs, dG0 = sum of per-group parts.
// LDS float atomics cost ~160 LDS cycles per wave-instruction on gfx950 (measured), global
// ones ~1.3 TB/s chip-wide; a store pass + per-destination sum pass is several times cheaper.
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_bwd_chunk_kernel(const float* __restrict__ G2, uint32_t G,
                                                             const float* __restrict__ d_out, GroupPlan plan) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, lo = lane & 15;
  const int b_l = lane >> 2, j_l = lane & 3;   // this lane's id inside a chunk, and which pieces of its rows
  float* pbuf = smem;
  float* bbuf = pbuf + C::PB_FLOATS;   // staged G2 rows
  float* dbuf = bbuf + C::BB2_FLOATS;  // staged d_output rows

  // A wavefront owns the groups whose FIRST chunk lies in its kCPW descriptors: it skips the tail of a
  // group that began earlier and follows its last group to the end, so every group is handled by exactly
  // one wavefront and its dP needs no partial sums.
  const uint32_t nchunks = (uint32_t)(plan.gpre[G] >> 32);
  const uint32_t c0 = blockIdx.x * kCPW;
  if (c0 >= nchunks) return;
  const uint32_t c1 = c0 + kCPW < nchunks ? c0 + kCPW : nchunks;
  uint32_t c = c0;
  uint4 d_cur = plan.ctab[c];
  if (!(d_cur.z & kFirstBit)) {
    c = d_cur.w;  // first chunk of the next group
    if (c >= c1) return;
    d_cur = plan.ctab[c];
  }

  // ---- lane-constant LDS offsets of the MFMA operands ----
  // dP step (b4, kk): lane group `hi` contributes id b = 4 b4 + hi, column kk of that id
  int offA[C::MT2], offB[C::RT2], offE[C::NT2];
#pragma unroll
  for (int mt = 0; mt < C::MT2; ++mt) {
    const int m = 16 * mt + lo < C::M2 ? 16 * mt + lo : C::M2 - 1;  // rows past M2 are discarded
    offA[mt] = hi * C::LDOB + m * Q2;
  }
#pragma unroll
  for (int t = 0; t < C::RT2; ++t) offB[t] = hi * C::LDBB + ((16 * t + lo) % R2) * Q2;
#pragma unroll
  for (int nt = 0; nt < C::NT2; ++nt) {
    const int col = 16 * nt + lo;
    offE[nt] = (col / Q2) * C::LDOB + col % Q2 + hi * Q2;
  }
  constexpr int F4G = C::ROW2 / 4, NLG = (F4G + 3) / 4;   // float4 pieces of a G2 row / per lane (4 lanes per id)
  constexpr int F4D = C::D / 4, NLD = (F4D + 3) / 4;      // float4 pieces of a d_output row / per lane
  constexpr int PF4 = C::M2 * R2 / 4, NLP = (PF4 + kWave - 1) / kWave;
  const uint4 none = make_uint4(0u, 0u, 0u, 0u);

  auto fetch_meta = [&](const uint4& d, uint32_t& i2, uint32_t& val) {
    i2 = 0u;
    val = 0xffffffffu;  // no row: the staged gradient row is zero
    if (b_l < (int)(d.z & 0xffu)) {
      i2 = plan.i2s[d.x + b_l];
      val = plan.vals[d.x + b_l] & ~kMultiBit;
    }
  };
  // the chunk's G2 rows, d_output rows and (for a group's first chunk) P travel through registers: they are
  // requested one chunk ahead so that their HBM / L2 latency hides behind the previous chunk's MFMAs
  float4 pre_g[NLG], pre_d[NLD], pre_p[NLP];
  auto request = [&](const uint4& d, uint32_t i2, uint32_t val) {
    const float* row = G2 + i2 * (uint32_t)C::ROW2;
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      pre_g[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (F4G % 4 == 0 || idx < F4G) pre_g[k] = *reinterpret_cast<const float4*>(row + 4 * idx);
    }
    const float* grow = d_out + val * (uint32_t)C::D;  // B*D < 2^32: checked on the host
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = j_l + 4 * k;
      pre_d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((F4D % 4 == 0 || idx < F4D) && val != 0xffffffffu) pre_d[k] = *reinterpret_cast<const float4*>(grow + 4 * idx);
    }
    if (d.z & kFirstBit) {
      const float* p = plan.ptab + (size_t)d.y * (C::M2 * R2);
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;
        pre_p[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (PF4 % kWave == 0 || e < PF4) pre_p[it] = *reinterpret_cast<const float4*>(p + 4 * e);
      }
    }
  };

  f32x4 dp[C::MT2][C::RT2];
  auto store_dp = [&](uint32_t group) {
    float* dst = plan.dptab + (size_t)group * (C::M2 * R2);
#pragma unroll
    for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
      for (int t = 0; t < C::RT2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mt + 4 * hi + r;
          const int c2 = 16 * t + lo;
          if (m < C::M2 && c2 < R2) dst[m * R2 + c2] = dp[mt][t][r];
        }
  };

  // which chunk follows `cc` for this wavefront: the next one, unless `cc` closed a group at or past the range end
  auto has_next = [&](uint32_t cc, const uint4& d) { return cc + 1 < nchunks && !((d.z & kLastBit) && cc + 1 >= c1); };

  bool more1 = has_next(c, d_cur);
  uint4 d_nxt = more1 ? plan.ctab[c + 1] : none;
  uint32_t i2_cur, val_cur, i2_nxt = 0u, val_nxt = 0xffffffffu;
  fetch_meta(d_cur, i2_cur, val_cur);
  request(d_cur, i2_cur, val_cur);
  if (more1) fetch_meta(d_nxt, i2_nxt, val_nxt);

  for (;;) {
    const int len = (int)(d_cur.z & 0xffu);
    const uint32_t here = d_cur.x;
    // ---- group change: P of the new group -> LDS, dP starts from zero ----
    if (d_cur.z & kFirstBit) {
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) dp[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;  // float4 number e of the (q0 q1) x r2 matrix
        if (PF4 % kWave == 0 || e < PF4)
          *reinterpret_cast<float4*>(pbuf + (4 * e / R2) * C::LDPB + (4 * e) % R2) = pre_p[it];
      }
    }
    // ---- the chunk's rows: registers -> LDS ----
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      if (F4G % 4 == 0 || idx < F4G) *reinterpret_cast<float4*>(bbuf + b_l * C::LDBB + 4 * idx) = pre_g[k];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int idx = j_l + 4 * k;
      if (F4D % 4 == 0 || idx < F4D) *reinterpret_cast<float4*>(dbuf + b_l * C::LDOB + 4 * idx) = pre_d[k];
    }
    __builtin_amdgcn_sched_barrier(0);  // the row registers are free again only after the stores above
    // ---- request the next chunk's rows and the chunk after's indices now; they land while this chunk computes ----
    const uint4 d_done = d_cur;
    const bool more2 = more1 && has_next(c + 1, d_nxt);
    const uint4 d_nn = more2 ? plan.ctab[c + 2] : none;
    if (more1) request(d_nxt, i2_nxt, val_nxt);
    uint32_t i2_nn = 0u, val_nn = 0xffffffffu;
    if (more2) fetch_meta(d_nn, i2_nn, val_nn);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);

    // ---- dP += dO (q0q1 x 16 q2) . G2s^T (16 q2 x r2) ----
#pragma unroll
    for (int b4 = 0; b4 < kChunk / 4; ++b4)
#pragma unroll
      for (int kk = 0; kk < Q2; ++kk) {
        float av[C::MT2], bv[C::RT2];
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt) av[mt] = dbuf[offA[mt] + b4 * 4 * C::LDOB + kk];
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) {
          bv[t] = bbuf[offB[t] + b4 * 4 * C::LDBB + kk];
          if (16 * t + lo >= R2) bv[t] = 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
            dp[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], dp[mt][t], 0, 0, 0);
        if (kk == Q2 - 1) __builtin_amdgcn_sched_barrier(0);  // keeps operand loads from piling up in registers
      }

    // ---- E = P^T (r2 x q0q1) . dO (q0q1 x 16 q2) ----
    f32x4 e[C::RT2][C::NT2];
#pragma unroll
    for (int t = 0; t < C::RT2; ++t)
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) e[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < C::M2 / 4; ++s) {
      float av[C::RT2];
#pragma unroll
      for (int t = 0; t < C::RT2; ++t) {
        av[t] = pbuf[(4 * s + hi) * C::LDPB + (16 * t + lo) % R2];
        if (16 * t + lo >= R2) av[t] = 0.f;
      }
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        const float bv = dbuf[offE[nt] + 4 * s * Q2];
#pragma unroll
        for (int t = 0; t < C::RT2; ++t)
          e[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, e[t][nt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // E leaves straight from the accumulators.  Lane (hi, lo) holds E[c2 = 16 t + 4 hi + r][col = 16 nt + lo]
    // with col = id * q2 + kk, and the E table keeps an id's row as [kk][c2] (the reduce kernel sums rows
    // element by element, the finalize kernel puts dG2 back into [c2][kk]): the chunk's rows are then one
    // contiguous block indexed col * r2 + c2, and the 64 lanes of one (t, nt) write 16-byte pieces of it.
    {
      float* dst = plan.etab + (size_t)here * C::ROW2;
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        const int col = 16 * nt + lo;
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) {
          const int c2 = 16 * t + 4 * hi;
          if (col < len * Q2 && c2 < R2)
            *reinterpret_cast<float4*>(dst + col * R2 + c2) = make_float4(e[t][nt][0], e[t][nt][1], e[t][nt][2], e[t][nt][3]);
        }
      }
    }
    if (d_done.z & kLastBit) store_dp(d_done.y);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every LDS read of this chunk is done before the next rows land
    if (!more1) break;
    ++c;
    d_cur = d_nxt;
    d_nxt = d_nn;
    more1 = more2;
    i2_nxt = i2_nn;
    val_nxt = val_nn;
  }
}

// B. dG2 reduce.  A workgroup takes kRowsB consecutive E rows, buckets them by i2 inside LDS
// (a tile-local counting sort of row numbers), then each wave sums the rows of "its" i2 values
// in registers and stores one (r2 q2)-float row per i2 into the tile's slab of partial sums
// (plain stores: atomics from every tile onto the 45 KB of dG2 ran at ~0.1 TB/s).  E rows are
// read exactly once, 16 bytes per lane.  fast3_finalize_kernel adds the slabs up.
template <int ROW2, int kRowsB, int NWB>
__global__ __launch_bounds__(NWB * 64) void fast3_dg2_reduce_kernel(GroupPlan plan, int G, uint32_t p2) {
  extern __shared__ uint32_t lds_u[];   // [p2 + 1] bucket starts | [p2] cursors | [kRowsB] row list (uint16)
  uint32_t* bstart = lds_u;
  uint32_t* cursor = lds_u + p2 + 1;
  unsigned short* rows = reinterpret_cast<unsigned short*>(cursor + p2);
  constexpr int F4 = ROW2 / 4;      // float4 per row
  constexpr int SUB = kWave / F4;   // rows handled per load instruction
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t total = (uint32_t)plan.gpre[G];
  const uint32_t s0 = blockIdx.x * kRowsB;
  const uint32_t n_rows = s0 >= total ? 0u : (s0 + kRowsB < total ? kRowsB : total - s0);
  float* slab = plan.g2part + (size_t)blockIdx.x * p2 * ROW2;  // this tile's partial dG2, every row written
  for (uint32_t e = tid; e <= p2; e += NWB * 64) bstart[e] = 0;
  __syncthreads();
  // histogram of i2 over the tile (integer LDS atomics; 8 ids per thread)
  uint32_t my_i2[kRowsB / (NWB * 64)], my_rank[kRowsB / (NWB * 64)];
#pragma unroll
  for (int k = 0; k < kRowsB / (NWB * 64); ++k) {
    const uint32_t r = k * NWB * 64 + tid;
    my_i2[k] = 0xffffffffu;
    if (r < n_rows) {
      my_i2[k] = plan.i2s[s0 + r];
      my_rank[k] = atomicAdd(&bstart[my_i2[k] + 1], 1u);
    }
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of <= 1024 buckets by one wave
    uint32_t carry = 0;
    for (uint32_t base = 0; base < p2; base += kWave) {
      const uint32_t i = base + lane;
      uint32_t v = i < p2 ? bstart[i + 1] : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += up;
      }
      if (i < p2) bstart[i + 1] = carry + incl;
      carry += __shfl(incl, kWave - 1, kWave);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRowsB / (NWB * 64); ++k)
    if (my_i2[k] != 0xffffffffu) rows[bstart[my_i2[k]] + my_rank[k]] = (unsigned short)(k * NWB * 64 + tid);
  __syncthreads();
  // wave w sums the buckets i2 = w, w + NWB, ...
  const int sub = lane / F4, c4 = lane - sub * F4;
  for (uint32_t i2 = wave; i2 < p2; i2 += NWB) {
    const uint32_t b0 = bstart[i2], b1 = bstart[i2 + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub < SUB) {
      // several independent row loads in flight per lane group (the loop is latency-bound otherwise)
      constexpr int U = 6;
      for (uint32_t j = b0 + sub; j < b1; j += U * SUB) {
        uint32_t rr[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rr[u] = j + u * SUB < b1 ? rows[j + u * SUB] : 0xffffffffu;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rr[u] != 0xffffffffu)
            v[u] = *reinterpret_cast<const float4*>(plan.etab + (size_t)(s0 + rr[u]) * ROW2 + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
        }
      }
    }
#pragma unroll
    for (int k = 1; k < SUB; ++k) {
      const int src = (lane + k * F4) & 63;
      const float x = __shfl(acc.x, src, kWave), y = __shfl(acc.y, src, kWave);
      const float z = __shfl(acc.z, src, kWave), w = __shfl(acc.w, src, kWave);
      if (lane < F4) { acc.x += x; acc.y += y; acc.z += z; acc.w += w; }
    }
    if (lane < F4) *reinterpret_cast<float4*>(slab + (size_t)i2 * ROW2 + 4 * lane) = acc;
  }
}

// C. group epilogue: one wavefront per kGroupsC consecutive groups (empty ones are skipped)
#ifndef TTEMB_GROUPS_C
#define TTEMB_GROUPS_C 8
#endif
constexpr int kGroupsC = TTEMB_GROUPS_C;
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_group_epilogue_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, uint32_t p0, uint32_t G, GroupPlan plan,
    float* __restrict__ dG1) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  __shared__ __attribute__((aligned(16))) float dpbuf[C::P_FLOATS];
  __shared__ __attribute__((aligned(16))) float g1buf[R1 * C::LDG];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, lo = lane & 15;
  const uint32_t g_begin = blockIdx.x * kGroupsC;
  const uint32_t g_end = g_begin + kGroupsC < G ? g_begin + kGroupsC : G;

  f32x4 g1acc[C::RT1][C::NT1];
#pragma unroll
  for (int t = 0; t < C::RT1; ++t)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint32_t cur_i1 = 0xffffffffu;
  auto flush_g1 = [&]() {
    float* dst = dG1 + (size_t)cur_i1 * C::ROW1;
#pragma unroll
    for (int t = 0; t < C::RT1; ++t)
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * t + 4 * hi + r;
          if (c < R1) atomicAdd(dst + c * C::N1 + 16 * nt + lo, g1acc[t][nt][r]);
        }
        g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  };

  // which of this wave's groups are non-empty (one lane per group), then walk them with the next
  // group's dP and G0 operand already requested while the current one is being multiplied
  const uint32_t cnt_l = (lane < kGroupsC && g_begin + lane < g_end) ? plan.counts[g_begin + lane] : 0u;
  unsigned long long live = __ballot(cnt_l != 0);
  if (!live) return;
  constexpr int PER = (C::M2 * R2 + kWave - 1) / kWave;
  constexpr int KS0 = (Q0 + 3) / 4;
  float nxt[PER], nxt_g0[KS0][C::RT1];
  auto request = [&](uint32_t g) {
    const float* src = plan.dptab + (size_t)g * (C::M2 * R2);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = i * kWave + lane;
      nxt[i] = e < C::M2 * R2 ? src[e] : 0.f;
    }
    const uint32_t i1n = g / p0;
    const float* g0 = G0 + (size_t)(g - i1n * p0) * C::ROW0;
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) {
        const int a = 4 * s + hi;
        nxt_g0[s][t] = (a < Q0 && 16 * t + lo < R1) ? g0[a * R1 + 16 * t + lo] : 0.f;
      }
  };
  request(g_begin + __builtin_ctzll(live));
  while (live) {
    const uint32_t g = g_begin + __builtin_ctzll(live);
    live &= live - 1;
    const uint32_t i1 = g / p0;
    // dP of the group -> LDS matrix [m2][c2]
    float g0v[KS0][C::RT1];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = i * kWave + lane;
      if (e < C::M2 * R2) dpbuf[(e / R2) * C::LDA + e % R2] = nxt[i];
    }
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) g0v[s][t] = nxt_g0[s][t];
    if (live) request(g_begin + __builtin_ctzll(live));
    if (i1 != cur_i1) {
      if (cur_i1 != 0xffffffffu) flush_g1();
      cur_i1 = i1;
      const float* g1 = G1 + (size_t)i1 * C::ROW1;
#pragma unroll
      for (int it = 0; it < (C::ROW1 + kWave - 1) / kWave; ++it) {
        const int e = it * kWave + lane;
        if (e < C::ROW1) g1buf[(e / C::N1) * C::LDG + e % C::N1] = g1[e];
      }
    }
    __syncthreads();
    // dG1[i1] += G0[i0]^T (r1 x q0) . dP (q0 x q1 r2)
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int a = 4 * s + hi;
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        const float bv = a < Q0 ? dpbuf[(a * Q1 + n / R2) * C::LDA + n % R2] : 0.f;
#pragma unroll
        for (int t = 0; t < C::RT1; ++t)
          g1acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v[s][t], bv, g1acc[t][nt], 0, 0, 0);
      }
    }
    // dG0[i0] += dP (q0 x q1 r2) . G1[i1]^T (q1 r2 x r1); four interleaved accumulation chains
    f32x4 g0part[4][C::RT1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) g0part[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < C::N1 / 4; ++s) {
      const int n = 4 * s + hi;
      const float av = lo < Q0 ? dpbuf[(lo * Q1 + n / R2) * C::LDA + n % R2] : 0.f;
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) {
        const float bv = 16 * t + lo < R1 ? g1buf[(16 * t + lo) * C::LDG + n] : 0.f;
        g0part[s & 3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, g0part[s & 3][t], 0, 0, 0);
      }
    }
    float* dst0 = plan.g0part + (size_t)g * C::ROW0;  // summed over i1 by fast3_finalize_kernel
#pragma unroll
    for (int t = 0; t < C::RT1; ++t) {
      const f32x4 sum = (g0part[0][t] + g0part[1][t]) + (g0part[2][t] + g0part[3][t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 4 * hi + r;
        if (a < Q0 && 16 * t + lo < R1) dst0[a * R1 + 16 * t + lo] = sum[r];
      }
    }
    __syncthreads();
  }
  if (cur_i1 != 0xffffffffu) flush_g1();
}

// D. finalize: dG2 = sum of the per-tile slabs; dG0[i0] = sum over i1 of the per-group
// contributions of the non-empty groups.  A workgroup owns 32 consecutive outputs; its 8 lane
// rows split the terms, so every load instruction reads 128 contiguous bytes per row and many
// are in flight; the 8 partial sums meet in LDS.  Every output is written exactly once.
__global__ __launch_bounds__(256) void fast3_finalize_kernel(GroupPlan plan, int tiles, int p0, int p1,
                                                             int g2_floats, int row0, int q2, int r2,
                                                             float* __restrict__ dG0, float* __restrict__ dG2) {
  __shared__ float part[8][33];
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + x;
  const int n0 = p0 * row0;
  float s = 0.f;
  if (e < g2_floats) {
    for (int t = y; t < tiles; t += 8) s += plan.g2part[(size_t)t * g2_floats + e];
  } else if (e < g2_floats + n0) {
    const int o = e - g2_floats;
    const int i0 = o / row0, c = o - i0 * row0;
    for (int i1 = y; i1 < p1; i1 += 8) {
      const int g = i1 * p0 + i0;
      if (plan.counts[g]) s += plan.g0part[(size_t)g * row0 + c];
    }
  }
  part[y][x] = s;
  __syncthreads();
  if (y == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += part[k][x];
    if (e < g2_floats) {  // the slabs hold rows as [kk][c2] (see fast3_bwd_chunk_kernel); dG2 rows are [c2][kk]
      const int row2 = q2 * r2;
      const int i2 = e / row2, w = e - i2 * row2;
      const int kk = w / r2, c2 = w - kk * r2;
      dG2[i2 * row2 + c2 * q2 + kk] = tot;
    } else if (e < g2_floats + n0) {
      dG0[e - g2_floats] = tot;
    }
  }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
enum Fast3Kind { kNone = 0, kProducts, kArxiv, kPapers };

static Fast3Kind classify(const DevShape& s) {
  if (s.T != 3) return kNone;
  if ((long long)s.L[0] * s.p[0] >= 0x7fffffffll) return kNone;  // ids must fit the uint32 sort key
  auto is = [&](int q0, int q1, int q2, int r1, int r2) {
    return s.q[0] == q0 && s.q[1] == q1 && s.q[2] == q2 && s.R[1] == r1 && s.R[2] == r2;
  };
  if (is(4, 5, 5, 16, 16)) return kProducts;
  if (is(4, 4, 8, 8, 8)) return kArxiv;
  if (is(8, 4, 4, 32, 32)) return kPapers;
  return kNone;
}

bool fast3_supported(const DevShape& s) { return classify(s) != kNone; }

static int64_t num_groups(const DevShape& s) { return (int64_t)s.p[0] * s.p[1]; }

// rocPRIM's look-back scan needs a few bytes per block of ~1K items; reserve a generous bound so
// that sizing the workspace needs no HIP call (the launch checks the real requirement)
static int64_t scan_temp_bytes(int64_t G) { return 64 * 1024 + (G + 1) / 8; }

bool fast3_pays(const DevShape& s, int64_t nnz) { return nnz >= 2 * num_groups(s); }

#ifndef TTEMB_ROWS_B
#define TTEMB_ROWS_B 2048
#endif
constexpr int kRowsB = TTEMB_ROWS_B;  // E rows per workgroup of the dG2 reduce
constexpr int NWB = 16;
static int64_t reduce_tiles(int64_t nnz) { return (nnz + kRowsB - 1) / kRowsB; }

// The grouping that forward and backward share ("plan"): grouped (i2, row) pairs, group sizes / starts and
// the chunk table.  It lives in a caller buffer when one is given, else in the workspace.
static int64_t max_chunks(const DevShape& s, int64_t nnz) {
  const int64_t G = num_groups(s);
  return nnz / kChunk + (nnz < G ? nnz : G) + 1;
}

int64_t fast3_plan_bytes(const DevShape& s, int64_t nnz) {
  const int64_t G = num_groups(s);
  return 2 * align256(nnz * 4) + align256((G + 1) * 4) + align256((G + 1) * 8) + align256(max_chunks(s, nnz) * 16);
}

static void carve_plan_part(const DevShape& s, int64_t nnz, char* base, GroupPlan* pl) {
  const int64_t G = num_groups(s);
  pl->i2s = (uint32_t*)base;
  base += align256(nnz * 4);
  pl->vals = (uint32_t*)base;
  base += align256(nnz * 4);
  pl->counts = (uint32_t*)base;
  base += align256((G + 1) * 4);
  pl->gpre = (uint64_t*)base;
  base += align256((G + 1) * 8);
  pl->ctab = (uint4*)base;
}

// workspace layout: [plan part unless external] [prefix products] [grouping scratch] [backward tables]
static int64_t carve_workspace(const DevShape& s, int64_t nnz, bool bwd, bool plan_inside, bool need_grouping,
                               char* base, GroupPlan* pl, char** scan_tmp) {
  const int64_t G = num_groups(s);
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  if (plan_inside) {
    char* p = take(fast3_plan_bytes(s, nnz));
    if (pl && p) carve_plan_part(s, nnz, p, pl);
  }
  {
    float* pt = (float*)take(G * (int64_t)s.q[0] * s.q[1] * s.R[2] * 4);
    if (pl) pl->ptab = pt;
  }
  if (need_grouping) {
    uint32_t* a = (uint32_t*)take(nnz * 4);
    uint32_t* b = (uint32_t*)take(nnz * 4);
    uint32_t* c = (uint32_t*)take(nnz * 4);
    char* t = take(scan_temp_bytes(G));
    if (pl) {
      pl->keys_in = a;
      pl->vals_in = b;
      pl->rank_in = c;
    }
    if (scan_tmp) *scan_tmp = t;
  }
  if (bwd) {
    float* e = (float*)take(nnz * (int64_t)s.row_len[2] * 4);
    float* d = (float*)take(G * (int64_t)s.q[0] * s.q[1] * s.R[2] * 4);
    float* g2 = (float*)take(reduce_tiles(nnz) * (int64_t)s.p[2] * s.row_len[2] * 4);
    float* g0 = (float*)take(G * (int64_t)s.row_len[0] * 4);
    if (pl) {
      pl->etab = e;
      pl->dptab = d;
      pl->g2part = g2;
      pl->g0part = g0;
    }
  }
  return off;
}

int64_t fast3_workspace_bytes(const DevShape& s, int32_t op, int64_t nnz, int64_t B) {
  (void)B;
  return carve_workspace(s, nnz, op == TTEMB_OP_BACKWARD, true, true, nullptr, nullptr, nullptr) + 256;
}

// fill plan->{i2s, vals, counts, gpre, ctab} from the ids
static int group_ids(const DevShape& s, const int64_t* indices, const int64_t* rowidx, const int64_t* offsets,
                     int64_t nnz, const int32_t* nnz_dev, GroupPlan* plan, char* scan_tmp, hipStream_t st) {
  const int64_t G = num_groups(s);
  size_t tmp_bytes = 0;
  auto packed = rocprim::make_transform_iterator(plan->counts, PackCounts());
  hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, packed, plan->gpre, (uint64_t)0, (size_t)(G + 1),
                                         rocprim::plus<uint64_t>(), st, false);
  if (e != hipSuccess) return check_hip(e, "exclusive_scan(size)");
  if ((int64_t)tmp_bytes > scan_temp_bytes(G)) return fail(TTEMB_E_WORKSPACE, "scan scratch bound too small");
  const uint32_t sentinel = (uint32_t)((unsigned long long)s.L[0] * s.p[0]);
  int rc = check_hip(hipMemsetAsync(plan->counts, 0, (size_t)(G + 1) * 4, st), "memset counts");
  if (rc) return rc;
  const unsigned tiles = (unsigned)((nnz + kTile - 1) / kTile);
  hipLaunchKernelGGL(fast3_prep_kernel, dim3(tiles), dim3(kTile), 0, st, indices, rowidx, offsets, nnz, nnz_dev, sentinel,
                     (uint32_t)s.p[0], (uint32_t)s.p[1], (uint32_t)s.p[2], *plan);
  rc = check_hip(hipGetLastError(), "fast3_prep_kernel");
  if (rc) return rc;
  // gpre[g] = (first grouped position, first chunk) of group g; gpre[G] = (live ids, chunks)
  e = rocprim::exclusive_scan(scan_tmp, tmp_bytes, packed, plan->gpre, (uint64_t)0, (size_t)(G + 1),
                              rocprim::plus<uint64_t>(), st, false);
  if (e != hipSuccess) return check_hip(e, "exclusive_scan");
  hipLaunchKernelGGL(fast3_scatter_kernel, dim3(tiles), dim3(kTile), 0, st, nnz, nnz_dev, (uint32_t)s.p[2], *plan);
  return check_hip(hipGetLastError(), "fast3_scatter_kernel");
}

// resolve where the plan lives, carve the workspace, group the ids unless a ready plan was passed
static int prepare(const DevShape& s, bool bwd, const int64_t* indices, const int64_t* rowidx,
                   const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev, void* ws, int64_t ws_bytes, void* plan_buf, int64_t plan_bytes,
                   bool plan_ready, GroupPlan* plan, hipStream_t st) {
  memset(plan, 0, sizeof(*plan));
  const bool external = plan_buf != nullptr && plan_bytes >= fast3_plan_bytes(s, nnz);
  const bool reuse = external && plan_ready;
  char* scan_tmp = nullptr;
  const int64_t need = carve_workspace(s, nnz, bwd, !external, !reuse, reinterpret_cast<char*>(ws), plan, &scan_tmp);
  if (need > 0 && ws == nullptr) return fail(TTEMB_E_WORKSPACE, "fast path needs a workspace");
  if (need > ws_bytes)
    return fail(TTEMB_E_WORKSPACE, "fast path needs %lld workspace bytes, got %lld", (long long)need, (long long)ws_bytes);
  if (external) carve_plan_part(s, nnz, reinterpret_cast<char*>(plan_buf), plan);
  if (reuse) return TTEMB_OK;
  return group_ids(s, indices, rowidx, offsets, nnz, nnz_dev, plan, scan_tmp, st);
}

// P of every non-empty group from the cores as they are now (forward and backward each do this: the backward
// differentiates the chain at the current cores, like the reference's recompute)
template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_prefix(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, hipStream_t st) {
  hipLaunchKernelGGL((fast3_prefix_kernel<Q0, Q1, Q2, R1, R2>),
                     dim3((unsigned)((s.p[0] + kPrefixGroups - 1) / kPrefixGroups), (unsigned)s.p[1]), dim3(64), 0, st,
                     cores.c[0], cores.c[1], (uint32_t)s.p[0], plan);
  return check_hip(hipGetLastError(), "fast3_prefix_kernel");
}

static unsigned chunk_waves(const DevShape& s, int64_t nnz) { return (unsigned)((max_chunks(s, nnz) + kCPW - 1) / kCPW); }

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, float* output,
                       hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  int rc = run_prefix<Q0, Q1, Q2, R1, R2>(s, cores, plan, st);
  if (rc) return rc;
  const size_t lds = C::WAVE_FLOATS * sizeof(float);
  profile_begin(0, st);
  hipLaunchKernelGGL((fast3_forward_kernel<Q0, Q1, Q2, R1, R2>), dim3(chunk_waves(s, nnz)), dim3(64), lds, st, cores.c[2],
                     plan, (uint32_t)num_groups(s), output);
  profile_end(0, st);
  return check_hip(hipGetLastError(), "fast3_forward_kernel");
}

int launch_forward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev,
                         float* output, void* ws, int64_t ws_bytes, void* plan_buf, int64_t plan_bytes,
                         hipStream_t st) {
  if (nnz <= 0) return TTEMB_OK;
  GroupPlan plan;
  int rc = prepare(s, false, indices, rowidx, offsets, nnz, nnz_dev, ws, ws_bytes, plan_buf, plan_bytes, false, &plan, st);
  if (rc) return rc;
  switch (classify(s)) {
    case kProducts: return run_forward<4, 5, 5, 16, 16>(s, cores, plan, nnz, output, st);
    case kArxiv: return run_forward<4, 4, 8, 8, 8>(s, cores, plan, nnz, output, st);
    case kPapers: return run_forward<8, 4, 4, 32, 32>(s, cores, plan, nnz, output, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_backward(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz,
                        const float* d_output, const CorePtrsMut& d_cores, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const int64_t G = num_groups(s);
  profile_begin(1, st);
  int rc = run_prefix<Q0, Q1, Q2, R1, R2>(s, cores, plan, st);
  if (rc) return rc;
  const size_t lds = (size_t)(C::PB_FLOATS + C::BB2_FLOATS + C::OB_FLOATS) * sizeof(float);
  profile_begin(2, st);
  hipLaunchKernelGGL((fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2>), dim3(chunk_waves(s, nnz)), dim3(64), lds, st, cores.c[2],
                     (uint32_t)G, d_output, plan);
  profile_end(2, st);
  rc = check_hip(hipGetLastError(), "fast3_bwd_chunk_kernel");
  if (rc) return rc;
  const int tiles = (int)reduce_tiles(nnz);
  hipLaunchKernelGGL((fast3_dg2_reduce_kernel<C::ROW2, kRowsB, NWB>), dim3((unsigned)tiles), dim3(NWB * 64),
                     (size_t)(2 * s.p[2] + 1) * 4 + kRowsB * 2, st, plan, (int)G, (uint32_t)s.p[2]);
  rc = check_hip(hipGetLastError(), "fast3_dg2_reduce_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((fast3_group_epilogue_kernel<Q0, Q1, Q2, R1, R2>), dim3((unsigned)((G + kGroupsC - 1) / kGroupsC)),
                     dim3(64), 0, st, cores.c[0], cores.c[1], (uint32_t)s.p[0], (uint32_t)G, plan, d_cores.c[1]);
  rc = check_hip(hipGetLastError(), "fast3_group_epilogue_kernel");
  if (rc) return rc;
  {
    const int g2_floats = s.p[2] * C::ROW2;
    const int outs = g2_floats + s.p[0] * C::ROW0;
    hipLaunchKernelGGL(fast3_finalize_kernel, dim3((unsigned)((outs + 31) / 32)), dim3(256), 0, st, plan, tiles,
                       s.p[0], s.p[1], g2_floats, (int)C::ROW0, Q2, R2, d_cores.c[0], d_cores.c[2]);
  }
  profile_end(1, st);
  return check_hip(hipGetLastError(), "fast3_finalize_kernel");
}

int launch_backward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                          const int64_t* rowidx, int64_t nnz, const int32_t* nnz_dev,
                          const float* d_output, const CorePtrsMut& d_cores, void* ws, int64_t ws_bytes,
                          const void* plan_buf, int64_t plan_bytes, hipStream_t st) {
  // dG1 is accumulated with float atomics; dG0 and dG2 are written whole by the finalize kernel
  for (int t = 0; t < s.T; ++t) {
    if (nnz > 0 && t != 1) continue;
    int rc = check_hip(hipMemsetAsync(d_cores.c[t], 0, (size_t)s.p[t] * s.row_len[t] * 4, st), "memset d_core");
    if (rc) return rc;
  }
  if (nnz <= 0) return TTEMB_OK;
  GroupPlan plan;
  int rc = prepare(s, true, indices, rowidx, nullptr, nnz, nnz_dev, ws, ws_bytes, const_cast<void*>(plan_buf), plan_bytes,
                   plan_buf != nullptr, &plan, st);
  if (rc) return rc;
  switch (classify(s)) {
    case kProducts: return run_backward<4, 5, 5, 16, 16>(s, cores, plan, nnz, d_output, d_cores, st);
    case kArxiv: return run_backward<4, 4, 8, 8, 8>(s, cores, plan, nnz, d_output, d_cores, st);
    case kPapers: return run_backward<8, 4, 4, 32, 32>(s, cores, plan, nnz, d_output, d_cores, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

}  // namespace ttemb
