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


namespace ttemb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// Grouping pass: a counting sort of the live ids by group' = i1 * p0 + i0 (three tiny
// kernels; rocprim's radix/merge sort needs ~20 launches and > 100 us at these sizes).
//   key   = (i1 * p0 + i0) * p2 + i2   -- the id with its digits reordered: ids of one
//           (i0, i1) group end up adjacent, and consecutive groups share i1, which is what
//           lets the backward kernel keep dG1[i1] in registers;
//   value = output row | kMultiBit when the bag holds several ids.
// The order of ids inside a group is whatever the atomics produce; every consumer is
// insensitive to it except for the summation order of the backward (fp32 rounding only).
// ---------------------------------------------------------------------------------
__global__ void fast3_keys_hist_kernel(const int64_t* __restrict__ indices,
                                       const int64_t* __restrict__ rowidx, int64_t nnz,
                                       const int32_t* __restrict__ nnz_dev, uint32_t sentinel,
                                       uint32_t p0, uint32_t p1, uint32_t p2,
                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                       uint32_t* __restrict__ counts) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t cnt = live_count(nnz, nnz_dev);
  if (n >= cnt) return;
  int64_t id = indices[n];
  id = id < 0 ? 0 : (id >= (int64_t)sentinel ? (int64_t)sentinel - 1 : id);
  const int64_t row = rowidx[n];
  const bool multi = (n > 0 && rowidx[n - 1] == row) || (n + 1 < cnt && rowidx[n + 1] == row);
  const uint32_t u = (uint32_t)id;
  const uint32_t i0 = u / (p1 * p2);
  const uint32_t rem = u - i0 * (p1 * p2);
  const uint32_t i1 = rem / p2;
  const uint32_t i2 = rem - i1 * p2;
  const uint32_t group = i1 * p0 + i0;
  keys[n] = group * p2 + i2;
  vals[n] = (uint32_t)row | (multi ? kMultiBit : 0u);
  atomicAdd(&counts[group], 1u);
}

// in-place exclusive scan of counts[0..G) by one 1024-thread workgroup: counts -> cursors
__global__ __launch_bounds__(1024) void fast3_scan_kernel(uint32_t* __restrict__ counts, int G) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  const int per = (G + 1023) / 1024;
  const int lo = tid * per;
  const int hi = lo + per < G ? lo + per : G;
  uint32_t sum = 0;
  for (int i = lo; i < hi; ++i) sum += counts[i];
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = tid >= off ? part[tid - off] : 0u;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;
  for (int i = lo; i < hi; ++i) {
    const uint32_t c = counts[i];
    counts[i] = run;
    run += c;
  }
}

__global__ void fast3_scatter_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                     int64_t nnz, const int32_t* __restrict__ nnz_dev, uint32_t p2,
                                     uint32_t* __restrict__ cursor, uint32_t* __restrict__ keys_out,
                                     uint32_t* __restrict__ vals_out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= live_count(nnz, nnz_dev)) return;
  const uint32_t key = keys[n];
  const uint32_t dst = atomicAdd(&cursor[key / p2], 1u);
  keys_out[dst] = key;
  vals_out[dst] = vals[n];
}

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(256) void fast3_forward_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2,
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t nnz,
    const int32_t* __restrict__ nnz_dev, uint32_t p0, uint32_t p2, float* __restrict__ out) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 4, lo = lane & 15;
  float* pbuf = smem + wave * C::WAVE_FLOATS;
  float* bbuf = pbuf + C::P_FLOATS;
  float* obuf = bbuf;

  const int64_t cnt = live_count(nnz, nnz_dev);
  const int64_t begin = ((int64_t)blockIdx.x * 4 + wave) * kRange;
  if (begin >= cnt) return;
  const int range = (int)(begin + kRange < cnt ? kRange : cnt - begin);
  // the whole range's (key, value) pairs live in registers: one lane per sorted id
  uint32_t key_r = 0xffffffffu, val_r = 0;
  if (lane < range) {
    key_r = keys[begin + lane];
    val_r = vals[begin + lane];
  }

  uint32_t cur_group = 0xffffffffu;
  int pos = 0;
  while (pos < range) {
    // ---- chunk = leading run (<= 16) of ids that share the group of the id at `pos` ----
    const uint32_t key = __shfl(key_r, (pos + lo) & 63, kWave);
    const uint32_t val = __shfl(val_r, (pos + lo) & 63, kWave);
    const uint32_t group0 = __shfl(key_r, pos, kWave) / p2;
    const uint32_t my_group = key / p2;
    const unsigned long long same = __ballot(hi == 0 && pos + lo < range && my_group == group0);
    const int len = __builtin_ctzll(~same);
    const uint32_t i2 = lo < len ? key - my_group * p2 : 0u;  // lanes lo..lo+48 hold copies

    // ---- stage the chunk's G2 rows (row 0 stands in for unused slots) ----
    {
      constexpr int F4 = C::ROW2 / 4;  // float4 per row
#pragma unroll
      for (int it = 0; it < (kChunk * F4 + kWave - 1) / kWave; ++it) {
        const int f = it * kWave + lane;
        const int b = f / F4, c4 = f - b * F4;
        const uint32_t row2 = __shfl(i2, b < kChunk ? b : 0, kWave);
        if (f < kChunk * F4) {
          const float4 v = *reinterpret_cast<const float4*>(G2 + (size_t)row2 * C::ROW2 + 4 * c4);
          *reinterpret_cast<float4*>(bbuf + b * C::LDB + 4 * c4) = v;
        }
      }
    }

    // ---- stage 1 (once per group): P = G0[i0] . G1[i1] -> LDS ----
    if (group0 != cur_group) {
      cur_group = group0;
      const uint32_t i1 = group0 / p0;
      const uint32_t i0 = group0 - i1 * p0;
      const float* g0 = G0 + (size_t)i0 * C::ROW0;
      const float* g1 = G1 + (size_t)i1 * C::ROW1;
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
      for (int nt = 0; nt < C::NT2; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::KS2; ++s) {
        const float a = pbuf[(16 * mt + lo) * C::LDA + 4 * s + hi];
#pragma unroll
        for (int nt = 0; nt < C::NT2; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[s][nt], acc[nt], 0, 0, 0);
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
    {
      constexpr int D4 = C::D / 4;
#pragma unroll
      for (int it = 0; it < (kChunk * D4 + kWave - 1) / kWave; ++it) {
        const int f = it * kWave + lane;
        const int b = f / D4, c4 = f - b * D4;
        const uint32_t v = __shfl(val, b < kChunk ? b : 0, kWave);
        if (f < kChunk * D4 && b < len) {
          const float4 x = *reinterpret_cast<const float4*>(obuf + b * C::LDO + 4 * c4);
          float* dst = out + (size_t)(v & ~kMultiBit) * C::D + 4 * c4;
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
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pos += len;
  }
}

// ---------------------------------------------------------------------------------
// backward (dense core gradients; the fused optimiser epilogue runs afterwards)
//
// Same walk as the forward.  Per chunk of <= 16 ids of one (i0, i1) group, with dO the
// chunk's gradient rows viewed as a (q0 q1) x (16 q2) matrix and G2s the stacked G2 rows:
//     dP   += dO . G2s^T                 (q0 q1 x r2, accumulated over the group's chunks)
//     dG2s  = P^T . dO                   (r2 x 16 q2, scattered to dG2[i2] of each id)
// per group, once its chunks are done:
//     dG1[i1] += G0[i0]^T . dP           (r1 x q1 r2; stays in registers while i1 repeats)
//     dG0[i0] += dP . G1[i1]^T           (q0 x r1)
// All four are fp32 MFMA.  dG2 is accumulated in an LDS copy per workgroup (when the core
// fits) and flushed once with 256-byte atomic rows; dG1 is flushed when the wave's i1
// changes; dG0 is flushed per group (64 floats).  The reference issues 1424 global float
// atomics PER ID (FBTT/tt_embeddings_cuda.cu:364-379); here it is a few per id.
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2, int NW, bool G2LDS>
__global__ __launch_bounds__(NW * 64) void fast3_backward_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2,
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t nnz,
    const int32_t* __restrict__ nnz_dev, uint32_t p0, uint32_t p2, int64_t ids_per_wave,
    const float* __restrict__ d_out, float* __restrict__ dG0, float* __restrict__ dG1,
    float* __restrict__ dG2, int g2_floats, int dbg) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 4, lo = lane & 15;
  float* g2acc = smem;
  const int g2_rows = g2_floats / C::ROW2;
  const int g2_region = G2LDS ? ((g2_rows * C::LD2 + 3) & ~3) : 0;
  float* pbuf = smem + g2_region + wave * C::BWD_WAVE_FLOATS;
  float* bbuf = pbuf + C::P_FLOATS;   // staged G2 rows; reused for dP at group end
  float* dbuf = bbuf + C::BB_FLOATS;  // staged d_output rows; reused for G1[i1] at group end

  if (G2LDS) {
    for (int e = threadIdx.x; e < g2_rows * C::LD2; e += NW * 64) g2acc[e] = 0.f;
    __syncthreads();
  }

  const int64_t cnt = live_count(nnz, nnz_dev);
  const int64_t begin = ((int64_t)blockIdx.x * NW + wave) * ids_per_wave;
  const int64_t end = begin + ids_per_wave < cnt ? begin + ids_per_wave : cnt;

  f32x4 dp[C::MT2][C::RT2];     // dP of the current group
  f32x4 g1acc[C::RT1][C::NT1];  // dG1[i1] of the current i1
#pragma unroll
  for (int t = 0; t < C::RT1; ++t)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
    for (int t = 0; t < C::RT2; ++t) dp[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint32_t cur_group = 0xffffffffu, cur_i1 = 0xffffffffu, cur_i0 = 0;

  // dG1[i1] += accumulators ; accumulators = 0
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

  // group epilogue: fold the group's dP into dG1 (registers) and dG0 (atomics)
  auto flush_group = [&]() {
    float* dpbuf = bbuf;
    float* g1buf = dbuf;
    // dP accumulators (row 16 mt + 4 hi + r, col 16 t + lo) -> LDS matrix [m2][c2]
#pragma unroll
    for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
      for (int t = 0; t < C::RT2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mt + 4 * hi + r;
          const int c2 = 16 * t + lo;
          if (m < C::M2 && c2 < R2) dpbuf[m * C::LDA + c2] = dp[mt][t][r];
        }
        dp[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    // G1[i1] (r1 x q1 r2) -> LDS with a padded row stride, coalesced reads
    const float* g1 = G1 + (size_t)cur_i1 * C::ROW1;
#pragma unroll
    for (int it = 0; it < (C::ROW1 + kWave - 1) / kWave; ++it) {
      const int e = it * kWave + lane;
      if (e < C::ROW1) {
        const int c = e / C::N1, n = e - c * C::N1;
        g1buf[c * C::LDG + n] = g1[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // dG1[i1] += G0[i0]^T (r1 x q0) . dP (q0 x q1 r2)
    const float* g0 = G0 + (size_t)cur_i0 * C::ROW0;
#pragma unroll
    for (int s = 0; s < (Q0 + 3) / 4; ++s) {
      const int a = 4 * s + hi;
      float av[C::RT1];
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) av[t] = (a < Q0 && 16 * t + lo < R1) ? g0[a * R1 + 16 * t + lo] : 0.f;
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        const float bv = a < Q0 ? dpbuf[(a * Q1 + n / R2) * C::LDA + n % R2] : 0.f;
#pragma unroll
        for (int t = 0; t < C::RT1; ++t)
          g1acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, g1acc[t][nt], 0, 0, 0);
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
    f32x4 g0acc[C::RT1];
#pragma unroll
    for (int t = 0; t < C::RT1; ++t) g0acc[t] = (g0part[0][t] + g0part[1][t]) + (g0part[2][t] + g0part[3][t]);
    float* dst0 = dG0 + (size_t)cur_i0 * C::ROW0;
#pragma unroll
    for (int t = 0; t < C::RT1; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 4 * hi + r;
        if (a < Q0 && 16 * t + lo < R1) atomicAdd(dst0 + a * R1 + 16 * t + lo, g0acc[t][r]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  int64_t pos = begin;
  while (pos < end) {
    // ---- chunk discovery (as in the forward) ----
    uint32_t key = 0xffffffffu, val = 0;
    if (lo + pos < end && hi == 0) {
      key = keys[pos + lo];
      val = vals[pos + lo];
    }
    const uint32_t key0 = __shfl(key, 0, kWave);
    const uint32_t group0 = key0 / p2;
    const uint32_t my_group = key / p2;
    const unsigned long long same = __ballot(hi == 0 && lo + pos < end && my_group == group0);
    const int len = __builtin_ctzll(~same);
    const uint32_t i2 = (lo < len && hi == 0) ? key - my_group * p2 : 0u;

    if (group0 != cur_group) {
      if (cur_group != 0xffffffffu && !(dbg & 2)) flush_group();
      const uint32_t i1 = group0 / p0;
      if (i1 != cur_i1) {
        if (cur_i1 != 0xffffffffu) flush_g1();
        cur_i1 = i1;
      }
      cur_group = group0;
      cur_i0 = group0 - i1 * p0;
      // ---- stage 1: P = G0[i0] . G1[i1] -> LDS ----
      const float* g0 = G0 + (size_t)cur_i0 * C::ROW0;
      const float* g1 = G1 + (size_t)cur_i1 * C::ROW1;
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
          if (a < Q0) pbuf[(a * Q1 + j) * C::LDA + c2] = acc[nt][r];
        }
      }
    }

    // ---- stage the chunk's G2 rows and d_output rows (zero rows for unused slots) ----
    __builtin_amdgcn_sched_barrier(0);
    {
      constexpr int F4 = C::ROW2 / 4;
#pragma unroll
      for (int it = 0; it < (kChunk * F4 + kWave - 1) / kWave; ++it) {
        const int f = it * kWave + lane;
        const int b = f / F4, c4 = f - b * F4;
        const uint32_t row2 = __shfl(i2, b < kChunk ? b : 0, kWave);
        if (f < kChunk * F4) {
          const float4 v = *reinterpret_cast<const float4*>(G2 + (size_t)row2 * C::ROW2 + 4 * c4);
          *reinterpret_cast<float4*>(bbuf + b * C::LDB + 4 * c4) = v;
        }
      }
      constexpr int D4 = C::D / 4;
#pragma unroll
      for (int it = 0; it < (kChunk * D4 + kWave - 1) / kWave; ++it) {
        const int f = it * kWave + lane;
        const int b = f / D4, c4 = f - b * D4;
        const uint32_t v = __shfl(val, b < kChunk ? b : 0, kWave);
        if (f < kChunk * D4) {
          float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
          if (b < len && !(dbg & 8)) x = *reinterpret_cast<const float4*>(d_out + (size_t)(v & ~kMultiBit) * C::D + 4 * c4);
          *reinterpret_cast<float4*>(dbuf + b * C::LDO + 4 * c4) = x;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- dP += dO (q0q1 x 16 q2) . G2s^T (16 q2 x r2) ----
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 4))
#pragma unroll
    for (int s = 0; s < 4 * Q2; ++s) {
      const int col = 4 * s + hi;
      const int b = col / Q2, kk = col % Q2;
      float av[C::MT2], bv[C::RT2];
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt) {
        const int m = 16 * mt + lo < C::M2 ? 16 * mt + lo : C::M2 - 1;  // rows past M2 are discarded
        av[mt] = dbuf[b * C::LDO + m * Q2 + kk];
      }
#pragma unroll
      for (int t = 0; t < C::RT2; ++t) bv[t] = 16 * t + lo < R2 ? bbuf[b * C::LDB + (16 * t + lo) * Q2 + kk] : 0.f;
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
        for (int t = 0; t < C::RT2; ++t)
          dp[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], dp[mt][t], 0, 0, 0);
    }

    // ---- dG2s = P^T (r2 x q0q1) . dO (q0q1 x 16 q2), scattered per id ----
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4 e[C::RT2][C::NT2];
#pragma unroll
      for (int t = 0; t < C::RT2; ++t)
#pragma unroll
        for (int nt = 0; nt < C::NT2; ++nt) e[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::M2 / 4; ++s) {
        const int m = 4 * s + hi;
        float av[C::RT2];
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) av[t] = 16 * t + lo < R2 ? pbuf[m * C::LDA + 16 * t + lo] : 0.f;
#pragma unroll
        for (int nt = 0; nt < C::NT2; ++nt) {
          const int col = 16 * nt + lo;
          const float bv = dbuf[(col / Q2) * C::LDO + m * Q2 + col % Q2];
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
            e[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, e[t][nt], 0, 0, 0);
        }
      }
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        const int col = 16 * nt + lo;
        const int b = col / Q2, kk = col % Q2;
        const uint32_t row2 = __shfl(i2, b, kWave);
        float* dst = G2LDS ? g2acc + row2 * C::LD2 + kk : dG2 + (size_t)row2 * C::ROW2 + kk;
#pragma unroll
        for (int t = 0; t < C::RT2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c2 = 16 * t + 4 * hi + r;
            if (c2 < R2 && b < len && !(dbg & 1)) atomicAdd(dst + c2 * Q2, e[t][nt][r]);
          }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    pos += len;
  }
  if (cur_group != 0xffffffffu) {
    flush_group();
    flush_g1();
  }
  if (G2LDS) {
    __syncthreads();
    for (int e = threadIdx.x; e < g2_floats; e += NW * 64) {
      const int row = e / C::ROW2;
      const float v = g2acc[row * C::LD2 + (e - row * C::ROW2)];
      if (v != 0.f) atomicAdd(dG2 + e, v);
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

int64_t fast3_workspace_bytes(const DevShape& s, int32_t op, int64_t nnz, int64_t B) {
  (void)op;
  (void)B;
  return 4 * align256(nnz * 4) + align256((num_groups(s) + 1) * 4) + 256;
}

struct SortedIds {
  uint32_t* keys;
  uint32_t* vals;
  uint32_t sentinel;
};

static int sort_ids(const DevShape& s, const int64_t* indices, const int64_t* rowidx, int64_t nnz,
                    const int32_t* nnz_dev, void* ws, int64_t ws_bytes, SortedIds* out, hipStream_t st) {
  if (ws == nullptr) return fail(TTEMB_E_WORKSPACE, "fast path needs a workspace");
  char* base = reinterpret_cast<char*>(ws);
  const int64_t seg = align256(nnz * 4);
  const int64_t G = num_groups(s);
  uint32_t* k_in = reinterpret_cast<uint32_t*>(base);
  uint32_t* v_in = reinterpret_cast<uint32_t*>(base + seg);
  uint32_t* k_out = reinterpret_cast<uint32_t*>(base + 2 * seg);
  uint32_t* v_out = reinterpret_cast<uint32_t*>(base + 3 * seg);
  uint32_t* counts = reinterpret_cast<uint32_t*>(base + 4 * seg);
  if (4 * seg + (G + 1) * 4 > ws_bytes)
    return fail(TTEMB_E_WORKSPACE, "fast path needs %lld workspace bytes, got %lld",
                (long long)(4 * seg + (G + 1) * 4), (long long)ws_bytes);
  const uint32_t sentinel = (uint32_t)((unsigned long long)s.L[0] * s.p[0]);
  int rc = check_hip(hipMemsetAsync(counts, 0, (size_t)(G + 1) * 4, st), "memset counts");
  if (rc) return rc;
  const int threads = 256;
  const unsigned blocks = (unsigned)((nnz + threads - 1) / threads);
  hipLaunchKernelGGL(fast3_keys_hist_kernel, dim3(blocks), dim3(threads), 0, st, indices, rowidx, nnz, nnz_dev,
                     sentinel, (uint32_t)s.p[0], (uint32_t)s.p[1], (uint32_t)s.p[2], k_in, v_in, counts);
  rc = check_hip(hipGetLastError(), "fast3_keys_hist_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(fast3_scan_kernel, dim3(1), dim3(1024), 0, st, counts, (int)G);
  rc = check_hip(hipGetLastError(), "fast3_scan_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(fast3_scatter_kernel, dim3(blocks), dim3(threads), 0, st, k_in, v_in, nnz, nnz_dev,
                     (uint32_t)s.p[2], counts, k_out, v_out);
  rc = check_hip(hipGetLastError(), "fast3_scatter_kernel");
  if (rc) return rc;
  out->keys = k_out;
  out->vals = v_out;
  out->sentinel = sentinel;
  return TTEMB_OK;
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward(const DevShape& s, const CorePtrs& cores, const SortedIds& ids, int64_t nnz,
                       const int32_t* nnz_dev, float* output, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const size_t lds = 4 * C::WAVE_FLOATS * sizeof(float);
  const int64_t waves = (nnz + kRange - 1) / kRange;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&fast3_forward_kernel<Q0, Q1, Q2, R1, R2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                       "hipFuncSetAttribute");
    if (rc) return rc;
    attr_set = true;
  }
  profile_begin(0, st);
  hipLaunchKernelGGL((fast3_forward_kernel<Q0, Q1, Q2, R1, R2>), dim3(blocks), dim3(256), lds, st, cores.c[0],
                     cores.c[1], cores.c[2], ids.keys, ids.vals, nnz, nnz_dev, (uint32_t)s.p[0],
                     (uint32_t)s.p[2], output);
  profile_end(0, st);
  return check_hip(hipGetLastError(), "fast3_forward_kernel");
}

int launch_forward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                         const int64_t* rowidx, int64_t nnz, const int32_t* nnz_dev, float* output,
                         void* ws, int64_t ws_bytes, hipStream_t st) {
  if (nnz <= 0) return TTEMB_OK;
  SortedIds ids;
  int rc = sort_ids(s, indices, rowidx, nnz, nnz_dev, ws, ws_bytes, &ids, st);
  if (rc) return rc;
  switch (classify(s)) {
    case kProducts: return run_forward<4, 5, 5, 16, 16>(s, cores, ids, nnz, nnz_dev, output, st);
    case kArxiv: return run_forward<4, 4, 8, 8, 8>(s, cores, ids, nnz, nnz_dev, output, st);
    case kPapers: return run_forward<8, 4, 4, 32, 32>(s, cores, ids, nnz, nnz_dev, output, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

template <int Q0, int Q1, int Q2, int R1, int R2, int NW, bool G2LDS>
static int run_backward_inst(const DevShape& s, const CorePtrs& cores, const SortedIds& ids, int64_t nnz,
                             const int32_t* nnz_dev, const float* d_output, const CorePtrsMut& d_cores, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const int g2_floats = s.p[2] * C::ROW2;
  const size_t lds = ((G2LDS ? ((s.p[2] * C::LD2 + 3) & ~3) : 0) + (size_t)NW * C::BWD_WAVE_FLOATS) * sizeof(float);
  // one workgroup per CU at most; every wave walks one contiguous slice of the sorted ids
  const int64_t max_waves = 256 * NW;
  int64_t waves = (nnz + kRange - 1) / kRange;
  if (waves > max_waves) waves = max_waves;
  const int64_t ids_per_wave = (nnz + waves - 1) / waves;
  const unsigned blocks = (unsigned)((waves + NW - 1) / NW);
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fast3_backward_kernel<Q0, Q1, Q2, R1, R2, NW, G2LDS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        "hipFuncSetAttribute");
    if (rc) return rc;
    attr_set = true;
  }
  profile_begin(1, st);
  hipLaunchKernelGGL((fast3_backward_kernel<Q0, Q1, Q2, R1, R2, NW, G2LDS>), dim3(blocks), dim3(NW * 64), lds, st,
                     cores.c[0], cores.c[1], cores.c[2], ids.keys, ids.vals, nnz, nnz_dev, (uint32_t)s.p[0],
                     (uint32_t)s.p[2], ids_per_wave, d_output, d_cores.c[0], d_cores.c[1], d_cores.c[2], g2_floats,
                     getenv("TTEMB_DBG") ? atoi(getenv("TTEMB_DBG")) : 0);
  profile_end(1, st);
  return check_hip(hipGetLastError(), "fast3_backward_kernel");
}

template <int Q0, int Q1, int Q2, int R1, int R2, int NW>
static int run_backward(const DevShape& s, const CorePtrs& cores, const SortedIds& ids, int64_t nnz,
                        const int32_t* nnz_dev, const float* d_output, const CorePtrsMut& d_cores, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  // a per-workgroup LDS copy of dG2 when the whole core fits beside the per-wave buffers
  const size_t with_g2 = ((size_t)s.p[2] * C::LD2 + 4 + (size_t)NW * C::BWD_WAVE_FLOATS) * sizeof(float);
  if (with_g2 <= 160 * 1024)
    return run_backward_inst<Q0, Q1, Q2, R1, R2, NW, true>(s, cores, ids, nnz, nnz_dev, d_output, d_cores, st);
  return run_backward_inst<Q0, Q1, Q2, R1, R2, NW, false>(s, cores, ids, nnz, nnz_dev, d_output, d_cores, st);
}

int launch_backward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                          const int64_t* rowidx, int64_t nnz, const int32_t* nnz_dev,
                          const float* d_output, const CorePtrsMut& d_cores, void* ws, int64_t ws_bytes,
                          hipStream_t st) {
  for (int t = 0; t < s.T; ++t) {
    int rc = check_hip(hipMemsetAsync(d_cores.c[t], 0, (size_t)s.p[t] * s.row_len[t] * 4, st), "memset d_core");
    if (rc) return rc;
  }
  if (nnz <= 0) return TTEMB_OK;
  SortedIds ids;
  int rc = sort_ids(s, indices, rowidx, nnz, nnz_dev, ws, ws_bytes, &ids, st);
  if (rc) return rc;
  switch (classify(s)) {
    case kProducts: return run_backward<4, 5, 5, 16, 16, 8>(s, cores, ids, nnz, nnz_dev, d_output, d_cores, st);
    case kArxiv: return run_backward<4, 4, 8, 8, 8, 8>(s, cores, ids, nnz, nnz_dev, d_output, d_cores, st);
    case kPapers: return run_backward<8, 4, 4, 32, 32, 4>(s, cores, ids, nnz, nnz_dev, d_output, d_cores, st);
    default: return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  }
}

}  // namespace ttemb
