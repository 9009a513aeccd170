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


namespace ttemb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef TTEMB_FWD_VARIANT
#define TTEMB_FWD_VARIANT 0
#endif
#ifndef TTEMB_FWD_WAVES
#define TTEMB_FWD_WAVES 1   // wavefronts per forward workgroup (each is independent)
#endif

constexpr int kChunk = 16;        // ids per stage-2 GEMM (N = 16 * q2 columns)
constexpr int kRange = 64;        // sorted ids walked by one wavefront
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
// Grouping pass: a counting sort of the live ids by group' = i1 * p0 + i0 (three small
// kernels; rocprim's radix/merge sort needs ~20 launches and > 100 us at these sizes).
//   key   = (i1 * p0 + i0) * p2 + i2   -- the id with its digits reordered: ids of one
//           (i0, i1) group end up adjacent, and consecutive groups share i1;
//   value = output row | kMultiBit when the bag holds several ids.
// One returning atomic per id (its arrival rank inside the group) in the first kernel, a
// rocPRIM exclusive scan of the group sizes, and an atomic-free scatter.  The order of ids inside a group is arrival order: every
// consumer is insensitive to it except for fp32 summation order in the backward.
// ---------------------------------------------------------------------------------
constexpr int kTile = 256;   // threads per workgroup in the grouping kernels

struct GroupPlan {           // device pointers into the caller's workspace
  uint32_t* keys_in;         // [nnz] ungrouped keys
  uint32_t* vals_in;
  uint32_t* rank_in;         // arrival rank of the id inside its group
  uint32_t* keys;            // [nnz] grouped
  uint32_t* vals;
  uint32_t* counts;          // [G+1] ids per group (entry G stays 0)
  uint32_t* gstart;          // [G+1] first grouped position of each group; [G] = live ids
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

__global__ __launch_bounds__(kTile) void fast3_scatter_kernel(int64_t nnz, const int32_t* __restrict__ nnz_dev,
                                                              uint32_t p2, GroupPlan plan) {
  const int64_t n = (int64_t)blockIdx.x * kTile + threadIdx.x;
  if (n >= live_count(nnz, nnz_dev)) return;
  const uint32_t key = plan.keys_in[n];
  const uint32_t dst = plan.gstart[key / p2] + plan.rank_in[n];
  plan.keys[dst] = key;
  plan.vals[dst] = plan.vals_in[n];
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64 * TTEMB_FWD_WAVES) void fast3_forward_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2,
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t nnz,
    const int32_t* __restrict__ nnz_dev, uint32_t p0, uint32_t p2, float* __restrict__ out) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hi = lane >> 4, lo = lane & 15;
  float* pbuf = smem + wave * C::WAVE_FLOATS;
  float* bbuf = pbuf + C::P_FLOATS;
  float* obuf = bbuf;

  const int64_t cnt = live_count(nnz, nnz_dev);
  const int64_t begin = ((int64_t)blockIdx.x * TTEMB_FWD_WAVES + wave) * kRange;
  if (begin >= cnt) return;
  const int range = (int)(begin + kRange < cnt ? kRange : cnt - begin);
  // the whole window's (group, i2, value) triples live in registers: one lane per grouped id
  uint32_t grp_r = 0xffffffffu, i2_r = 0, val_r = 0;
  if (lane < range) {
    const uint32_t key = keys[begin + lane];
    grp_r = key / p2;
    i2_r = key - grp_r * p2;
    val_r = vals[begin + lane];
  }

  constexpr int F4G = C::ROW2 / 4, NLG = (kChunk * F4G + kWave - 1) / kWave;
  constexpr int D4 = C::D / 4, NLO = (kChunk * D4 + kWave - 1) / kWave;
  struct Chunk {
    int len;
    uint32_t group, i2, val;
  };
  // chunk = leading run (<= 16) of ids that share the group of the id at `at`
  auto discover = [&](int at) {
    Chunk c;
    c.len = 0;
    c.group = 0xffffffffu;
    c.i2 = 0;
    c.val = 0;
    if (at >= range) return c;
    const uint32_t g = __shfl(grp_r, (at + lo) & 63, kWave);
    c.val = __shfl(val_r, (at + lo) & 63, kWave);
    c.group = __shfl(grp_r, at, kWave);
    const unsigned long long same = __ballot(hi == 0 && at + lo < range && g == c.group);
    c.len = __builtin_ctzll(~same);
    const uint32_t i2 = __shfl(i2_r, (at + lo) & 63, kWave);
    c.i2 = lo < c.len ? i2 : 0u;  // row 0 stands in for unused slots
    return c;
  };
  // the chunk's G2 rows are requested one chunk ahead (registers), so their L2 latency hides
  // behind the previous chunk's MFMAs
  float4 pre_g[NLG];
  auto request_rows = [&](const Chunk& c) {
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4G, c4 = f - b * F4G;
      const uint32_t row2 = __shfl(c.i2, b < kChunk ? b : 0, kWave);
      pre_g[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < kChunk * F4G) pre_g[it] = *reinterpret_cast<const float4*>(G2 + (row2 * (uint32_t)C::ROW2 + 4u * c4));
    }
  };

  uint32_t cur_group = 0xffffffffu;
  int pos = 0;
  Chunk cur = discover(pos);
  if (cur.len) request_rows(cur);
  while (cur.len) {
    // ---- stage 1 (once per group): P = G0[i0] . G1[i1] -> LDS ----
    if (cur.group != cur_group) {
      cur_group = cur.group;
      const uint32_t i1 = cur_group / p0;
      const uint32_t i0 = cur_group - i1 * p0;
      const float* g0 = G0 + (size_t)i0 * C::ROW0;
      const float* g1 = G1 + (size_t)i1 * C::ROW1;
      f32x4 acc[C::NT1];
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) {
        const int k = 4 * s + hi;
        const float a = lo < Q0 ? g0[lo * R1 + k] : 0.f;
#pragma unroll
        for (int nt = 0; nt < C::NT1; ++nt) {
          const float b = g1[k * C::N1 + 16 * nt + lo];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[nt], 0, 0, 0);
        }
      }
      // accumulator (row 4*hi + r, col 16*nt + lo) -> P as a (q0 q1) x r2 matrix
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        const int j = n / R2, c2 = n % R2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 4 * hi + r;
          if (a < Q0) pbuf[(a * Q1 + j) * C::LDA + c2] = acc[nt][r];
        }
      }
    }
    // ---- the chunk's G2 rows: registers -> LDS ----
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4G, c4 = f - b * F4G;
      if (f < kChunk * F4G) *reinterpret_cast<float4*>(bbuf + b * C::LDB + 4 * c4) = pre_g[it];
    }
#if TTEMB_FWD_VARIANT != 1
    __builtin_amdgcn_sched_barrier(0);
#endif
    const int len = cur.len;
    const uint32_t val = cur.val;
    pos += len;
    cur = discover(pos);
    if (cur.len) request_rows(cur);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#if TTEMB_FWD_VARIANT != 1
    __builtin_amdgcn_sched_barrier(0);
#endif

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
    // ---- 16-byte global stores of whole rows ----
#pragma unroll
    for (int it = 0; it < NLO; ++it) {
      const int f = it * kWave + lane;
      const int b = f / D4, c4 = f - b * D4;
      const uint32_t v = __shfl(val, b < kChunk ? b : 0, kWave);
      if (f < kChunk * D4 && b < len) {
        const float4 x = *reinterpret_cast<const float4*>(obuf + b * C::LDO + 4 * c4);
        float* dst = out + ((v & ~kMultiBit) * (uint32_t)C::D + 4u * c4);  // B*D < 2^32: checked on the host
        if (v & kMultiBit) {
          atomicAdd(dst + 0, x.x);
          atomicAdd(dst + 1, x.y);
          atomicAdd(dst + 2, x.z);
          atomicAdd(dst + 3, x.w);
        } else {
          *reinterpret_cast<float4*>(dst) = x;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------
// backward, atomics-free formulation (three kernels)
//
//  A. chunk kernel (stateless, one wavefront per 64 sorted ids): per chunk
//        dP  += dO . G2s^T                        (q0q1 x r2, per group-run)
//        E    = P^T . dO   -> one (r2 q2)-float row per id, stored at the id's grouped position
//     and per group-run the dP partial sum is stored in its own slot.  Plain stores only.
//  B. dG2[i2] = sum of the E rows whose id has that i2: each wave of a workgroup owns a slice of
//     the i2 range and accumulates its rows into an LDS copy with plain read-modify-writes.
//  C. group epilogue: per non-empty group (ranks are (i1, i0)-ordered) dP = sum of its parts,
//        dG1[i1] += G0[i0]^T . dP  (registers while i1 repeats),  dG0[i0] += dP . G1[i1]^T.
// LDS float atomics cost ~160 LDS cycles per wave-instruction on gfx950 (measured), global
// ones ~1.3 TB/s chip-wide; a store pass + per-destination sum pass is several times cheaper.
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_bwd_chunk_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2,
    int64_t nnz, const int32_t* __restrict__ nnz_dev, uint32_t p0, uint32_t p2,
    const float* __restrict__ d_out, GroupPlan plan) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x;
  const int hi = lane >> 4, lo = lane & 15;
  float* pbuf = smem;
  float* bbuf = pbuf + C::PB_FLOATS;  // staged G2 rows
  float* dbuf = bbuf + C::BB2_FLOATS; // staged d_output rows

  // A wavefront owns the groups that START inside its 64-id window [begin, end): it skips a
  // leading group that began earlier and follows its last group past `end`, so every group is
  // handled by exactly one wavefront and its dP needs no partial sums.
  const int64_t cnt = live_count(nnz, nnz_dev);
  const int64_t begin = (int64_t)blockIdx.x * kRange;
  if (begin >= cnt) return;
  const int64_t end = begin + kRange < cnt ? begin + kRange : cnt;
  int64_t pos = begin;
  if (begin > 0) {
    const uint32_t prev = plan.keys[begin - 1] / p2;
    while (pos < end) {  // wave-uniform scan, 64 keys at a time
      const uint32_t k = pos + lane < cnt ? plan.keys[pos + lane] : 0xffffffffu;
      const unsigned long long cont = __ballot(pos + lane < cnt && k / p2 == prev);
      const int run = __builtin_ctzll(~cont);
      pos += run;
      if (run < kWave) break;
    }
    if (pos >= end) return;  // the window holds nothing but the tail of an earlier group
  }

  // ---- lane-constant LDS offsets of the MFMA operands ----
  // dP step (b4, kk): lane group `hi` contributes id b = 4 b4 + hi, column kk of that id
  int offA[C::MT2], offB[C::RT2], offE[C::NT2];
#pragma unroll
  for (int mt = 0; mt < C::MT2; ++mt) {
    const int m = 16 * mt + lo < C::M2 ? 16 * mt + lo : C::M2 - 1;  // rows past M2 are discarded
    offA[mt] = hi * C::LDO + m * Q2;
  }
#pragma unroll
  for (int t = 0; t < C::RT2; ++t) offB[t] = hi * C::LDBB + ((16 * t + lo) % R2) * Q2;
#pragma unroll
  for (int nt = 0; nt < C::NT2; ++nt) {
    const int col = 16 * nt + lo;
    offE[nt] = (col / Q2) * C::LDO + col % Q2 + hi * Q2;
  }
  constexpr int F4G = C::ROW2 / 4, NLG = (kChunk * F4G + kWave - 1) / kWave;  // G2-row float4 loads per lane
  constexpr int F4D = C::D / 4, NLD = (kChunk * F4D + kWave - 1) / kWave;     // d_output-row float4 loads per lane

  f32x4 dp[C::MT2][C::RT2];
  uint32_t cur_group = 0xffffffffu;
  auto store_dp = [&]() {
    float* dst = plan.dptab + (size_t)cur_group * (C::M2 * R2);
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

  // ---- chunk discovery over a sliding 64-id register window ----
  int64_t win = -kWave;
  uint32_t grp_r = 0xffffffffu, i2_r = 0, val_r = 0;
  struct Chunk {
    int len;            // ids in the chunk, 0 = nothing left for this wavefront
    uint32_t group, i2, val;
  };
  auto discover = [&](int64_t at, uint32_t open_group) {
    Chunk c;
    c.len = 0;
    c.group = 0xffffffffu;
    c.i2 = 0;
    c.val = 0;
    if (at >= cnt) return c;
    if (at + kChunk > win + kWave) {  // slide the window; (group, i2) are decoded once per id here
      win = at;
      grp_r = 0xffffffffu;
      if (win + lane < cnt) {
        const uint32_t key = plan.keys[win + lane];
        grp_r = key / p2;
        i2_r = key - grp_r * p2;
        val_r = plan.vals[win + lane];
      }
    }
    const int off = (int)(at - win);
    const uint32_t g = __shfl(grp_r, (off + lo) & 63, kWave);
    c.val = __shfl(val_r, (off + lo) & 63, kWave);
    c.group = __shfl(grp_r, off, kWave);
    if (at >= end && c.group != open_group) return c;  // past the window: only the open group continues
    const unsigned long long same = __ballot(hi == 0 && at + lo < cnt && g == c.group);
    c.len = __builtin_ctzll(~same);
    const uint32_t i2 = __shfl(i2_r, (off + lo) & 63, kWave);
    c.i2 = lo < c.len ? i2 : 0u;
    return c;
  };
  // the chunk's G2 rows and d_output rows travel through registers: they are requested one
  // chunk ahead so that their HBM / L2 latency hides behind the previous chunk's MFMAs
  float4 pre_g[NLG], pre_d[NLD];
  auto request_rows = [&](const Chunk& c) {
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4G, c4 = f - b * F4G;
      const uint32_t row2 = __shfl(c.i2, b < kChunk ? b : 0, kWave);
      pre_g[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < kChunk * F4G) pre_g[it] = *reinterpret_cast<const float4*>(G2 + (row2 * (uint32_t)C::ROW2 + 4u * c4));
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4D, c4 = f - b * F4D;
      const uint32_t v = __shfl(c.val, b < kChunk ? b : 0, kWave);
      pre_d[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < kChunk * F4D && b < c.len)
#if defined(TTEMB_ABL) && (TTEMB_ABL & 2)
        pre_d[it] = *reinterpret_cast<const float4*>(d_out + ((v & 1023u) * (uint32_t)C::D + 4u * c4));  // ablation: cache-resident rows
#else
        pre_d[it] = *reinterpret_cast<const float4*>(d_out + ((v & ~kMultiBit) * (uint32_t)C::D + 4u * c4));  // B*D < 2^32: checked on the host
#endif
    }
  };

  Chunk cur = discover(pos, cur_group);
  if (cur.len) request_rows(cur);
  while (cur.len) {
    // ---- group change: close the previous group, form P = G0[i0] . G1[i1] ----
    if (cur.group != cur_group) {
      if (cur_group != 0xffffffffu) store_dp();
      cur_group = cur.group;
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) dp[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const uint32_t i1 = cur_group / p0;
      const uint32_t i0 = cur_group - i1 * p0;
      const float* g0 = G0 + (size_t)i0 * C::ROW0;
      const float* g1 = G1 + (size_t)i1 * C::ROW1;
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 16))
      f32x4 acc[C::NT1];
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) {
        const int k = 4 * s + hi;
        const float a = lo < Q0 ? g0[lo * R1 + k] : 0.f;
#pragma unroll
        for (int nt = 0; nt < C::NT1; ++nt) {
          const float b = g1[k * C::N1 + 16 * nt + lo];
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        const int j = n / R2, c2 = n % R2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 4 * hi + r;
          if (a < Q0) pbuf[(a * Q1 + j) * C::LDPB + c2] = acc[nt][r];
        }
      }
#endif
    }
    // ---- the chunk's rows: registers -> LDS ----
#pragma unroll
    for (int it = 0; it < NLG; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4G, c4 = f - b * F4G;
      if (f < kChunk * F4G) *reinterpret_cast<float4*>(bbuf + b * C::LDBB + 4 * c4) = pre_g[it];
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int f = it * kWave + lane;
      const int b = f / F4D, c4 = f - b * F4D;
      if (f < kChunk * F4D) *reinterpret_cast<float4*>(dbuf + b * C::LDO + 4 * c4) = pre_d[it];
    }
    __builtin_amdgcn_sched_barrier(0);  // the row registers are free again only after the stores above
    // ---- request the next chunk's rows now; they land while this chunk computes ----
    const int64_t here = pos;
    const int len = cur.len;
    pos += len;
    cur = discover(pos, cur_group);
    if (cur.len) request_rows(cur);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);

#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 4))
    // ---- dP += dO (q0q1 x 16 q2) . G2s^T (16 q2 x r2) ----
#pragma unroll
    for (int b4 = 0; b4 < kChunk / 4; ++b4)
#pragma unroll
      for (int kk = 0; kk < Q2; ++kk) {
        float av[C::MT2], bv[C::RT2];
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt) av[mt] = dbuf[offA[mt] + b4 * 4 * C::LDO + kk];
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

#endif
    // ---- E = P^T (r2 x q0q1) . dO (q0q1 x 16 q2) ----
    f32x4 e[C::RT2][C::NT2];
#pragma unroll
    for (int t = 0; t < C::RT2; ++t)
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) e[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 8))
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
#endif
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
#if defined(TTEMB_ABL) && (TTEMB_ABL & 1)
          if (col < len * Q2 && c2 < R2 && e[t][nt][0] == 123.456f)   // ablation: no E traffic
#else
          if (col < len * Q2 && c2 < R2)
#endif
            *reinterpret_cast<float4*>(dst + col * R2 + c2) = make_float4(e[t][nt][0], e[t][nt][1], e[t][nt][2], e[t][nt][3]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every LDS read of this chunk is done before the next rows land
  }
  if (cur_group != 0xffffffffu) store_dp();
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
  const uint32_t total = plan.gstart[G];
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
      const uint32_t key = plan.keys[s0 + r];
      my_i2[k] = key - (key / p2) * p2;
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

// The grouping that forward and backward share ("plan"): grouped keys / values and the group
// sizes / starts.  It lives in a caller buffer when one is given, else in the workspace.
int64_t fast3_plan_bytes(const DevShape& s, int64_t nnz) {
  return 2 * align256(nnz * 4) + 2 * align256((num_groups(s) + 1) * 4);
}

static void carve_plan_part(const DevShape& s, int64_t nnz, char* base, GroupPlan* pl) {
  const int64_t G = num_groups(s);
  pl->keys = (uint32_t*)base;
  pl->vals = (uint32_t*)(base + align256(nnz * 4));
  pl->counts = (uint32_t*)(base + 2 * align256(nnz * 4));
  pl->gstart = (uint32_t*)(base + 2 * align256(nnz * 4) + align256((G + 1) * 4));
}

// workspace layout: [plan part unless external] [grouping scratch] [backward tables]
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

// fill plan->{keys, vals, counts, gstart} from the ids
static int group_ids(const DevShape& s, const int64_t* indices, const int64_t* rowidx, const int64_t* offsets,
                     int64_t nnz, const int32_t* nnz_dev, GroupPlan* plan, char* scan_tmp, hipStream_t st) {
  const int64_t G = num_groups(s);
  size_t tmp_bytes = 0;
  uint32_t* nul = nullptr;
  hipError_t e = rocprim::exclusive_scan(nullptr, tmp_bytes, nul, nul, 0u, (size_t)(G + 1), rocprim::plus<uint32_t>(), st, false);
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
  // gstart[g] = first grouped position of group g; gstart[G] = number of live ids
  e = rocprim::exclusive_scan(scan_tmp, tmp_bytes, plan->counts, plan->gstart, 0u, (size_t)(G + 1),
                              rocprim::plus<uint32_t>(), st, false);
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

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward(const DevShape& s, const CorePtrs& cores, const GroupPlan& ids, int64_t nnz,
                       const int32_t* nnz_dev, float* output, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const size_t lds = TTEMB_FWD_WAVES * C::WAVE_FLOATS * sizeof(float);
  const unsigned blocks = (unsigned)((nnz + kRange * TTEMB_FWD_WAVES - 1) / (kRange * TTEMB_FWD_WAVES));
  profile_begin(0, st);
  hipLaunchKernelGGL((fast3_forward_kernel<Q0, Q1, Q2, R1, R2>), dim3(blocks), dim3(64 * TTEMB_FWD_WAVES), lds, st, cores.c[0],
                     cores.c[1], cores.c[2], ids.keys, ids.vals, nnz, nnz_dev, (uint32_t)s.p[0],
                     (uint32_t)s.p[2], output);
  profile_end(0, st);
  return check_hip(hipGetLastError(), "fast3_forward_kernel");
}

int launch_forward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev,
                         float* output, void* ws, int64_t ws_bytes, void* plan_buf, int64_t plan_bytes,
                         hipStream_t st) {
  if (nnz <= 0) return TTEMB_OK;
  GroupPlan ids;
  int rc = prepare(s, false, indices, rowidx, offsets, nnz, nnz_dev, ws, ws_bytes, plan_buf, plan_bytes, false, &ids, st);
  if (rc) return rc;
  switch (classify(s)) {
    case kProducts: return run_forward<4, 5, 5, 16, 16>(s, cores, ids, nnz, nnz_dev, output, st);
    case kArxiv: return run_forward<4, 4, 8, 8, 8>(s, cores, ids, nnz, nnz_dev, output, st);
    case kPapers: return run_forward<8, 4, 4, 32, 32>(s, cores, ids, nnz, nnz_dev, output, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_backward(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz,
                        const int32_t* nnz_dev, const float* d_output, const CorePtrsMut& d_cores,
                        hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const size_t lds = (size_t)(C::PB_FLOATS + C::BB2_FLOATS + C::O_FLOATS) * sizeof(float);
  const unsigned ranges = (unsigned)((nnz + kRange - 1) / kRange);
  profile_begin(1, st);
  profile_begin(2, st);
  hipLaunchKernelGGL((fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2>), dim3(ranges), dim3(64), lds, st, cores.c[0],
                     cores.c[1], cores.c[2], nnz, nnz_dev, (uint32_t)s.p[0], (uint32_t)s.p[2], d_output, plan);
  profile_end(2, st);
  int rc = check_hip(hipGetLastError(), "fast3_bwd_chunk_kernel");
  if (rc) return rc;
  const int tiles = (int)reduce_tiles(nnz);
  hipLaunchKernelGGL((fast3_dg2_reduce_kernel<C::ROW2, kRowsB, NWB>), dim3((unsigned)tiles), dim3(NWB * 64),
                     (size_t)(2 * s.p[2] + 1) * 4 + kRowsB * 2, st, plan, (int)num_groups(s), (uint32_t)s.p[2]);
  rc = check_hip(hipGetLastError(), "fast3_dg2_reduce_kernel");
  if (rc) return rc;
  const int64_t G = num_groups(s);
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
    case kProducts: return run_backward<4, 5, 5, 16, 16>(s, cores, plan, nnz, nnz_dev, d_output, d_cores, st);
    case kArxiv: return run_backward<4, 4, 8, 8, 8>(s, cores, plan, nnz, nnz_dev, d_output, d_cores, st);
    case kPapers: return run_backward<8, 4, 4, 32, 32>(s, cores, plan, nnz, nnz_dev, d_output, d_cores, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

}  // namespace ttemb
