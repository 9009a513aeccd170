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

#include <atomic>
#include <cstdlib>
#include <cstring>



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
  static constexpr int NT1 = (N1 + 15) / 16;     // (a last tile that q1 r2 does not fill is masked: q1 = 5 at rank 8)
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
  // d_output rows, read as [hi][m*q2+kk]; when q0 q1 is not a multiple of 4 the last K-step of the E product reads (and
  // discards) up to three rows past the end: the stride covers them
  static constexpr int DPAD = (M2 + 3) / 4 * 4 * Q2;
  static constexpr int LDOB = (((DPAD > D ? DPAD : D) + 15) / 32) * 32 + 16;
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
  static_assert(LDOB >= (M2 + 3) / 4 * 4 * Q2, "the padded K-step of the E product reads inside the staged row");
  static_assert(Q0 <= 16, "stage 1 pads q0 to one 16-row tile");
  static_assert(R1 % 4 == 0 && R2 % 4 == 0, "ranks must be multiples of the MFMA K");
  static_assert(N1 % 4 == 0, "rows of the prefix product move as float4");
  static_assert(D % 4 == 0 && ROW2 % 4 == 0, "rows move as float4");
  static_assert(ROW2 <= 256, "the dG2 reduce reads one E row per wavefront load");
};

// ---------------------------------------------------------------------------------
// Addressing helpers of the chain kernels.  Every global access of their inner loops goes through a buffer
// descriptor with a per-lane 32-bit byte offset: an offset past the buffer (kOob) makes a load return zeros and
// a store vanish, so ragged chunks need no branch around a memory instruction -- the instruction stream per
// chunk is fixed, and the compiler can count outstanding operations exactly (a skipped instruction would force
// every later wait to vmcnt(0), i.e. to drain the prefetch that was just issued).
// ---------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t kOob = 0xffffffffu;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(rsrc_t r, uint32_t voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ u32x4 buf_load4u(rsrc_t r, uint32_t voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
}
__device__ __forceinline__ uint32_t buf_load1u(rsrc_t r, uint32_t voff) {
  return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0);
}
__device__ __forceinline__ void buf_store4(rsrc_t r, uint32_t voff, const float4& x) {
  u32x4 v;
  v.x = __float_as_uint(x.x);
  v.y = __float_as_uint(x.y);
  v.z = __float_as_uint(x.z);
  v.w = __float_as_uint(x.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, 0);
}
// (experiment knob TTEMB_NT: bit mask of the stores issued non-temporally -- 1 the prefix-in-chain forward's P table, 2 its output
//  rows, 4 the plain forward's output rows, 8 the backward chunk kernel's E rows / dG0 parts)
#ifndef TTEMB_NT
#define TTEMB_NT 0
#endif
template <bool NT>
__device__ __forceinline__ void buf_store4_p(rsrc_t r, uint32_t voff, const float4& x) {
  u32x4 v;
  v.x = __float_as_uint(x.x);
  v.y = __float_as_uint(x.y);
  v.z = __float_as_uint(x.z);
  v.w = __float_as_uint(x.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, 0, NT ? 2 : 0);   // (aux 2 = nt on gfx94x / gfx950)
}
__device__ __forceinline__ void buf_store1(rsrc_t r, uint32_t voff, float x) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, (int)voff, 0, 0);
}

// Chunk descriptors are read through the constant address space with a wave-uniform index: scalar loads
// (s_load_dwordx4), which are counted by lgkmcnt, not vmcnt, and so never sit in the queue of the row loads.
typedef const __attribute__((address_space(4))) uint32_t* desc_ptr;
__device__ __forceinline__ uint4 load_desc(desc_ptr tab, uint32_t c, uint32_t nchunks) {
  const uint32_t at = 4u * (c < nchunks ? c : (nchunks ? nchunks - 1 : 0u));   // (the table always has an entry)
  const uint32_t keep = c < nchunks ? 0xffffffffu : 0u;   // past the table: an empty chunk
  return make_uint4(tab[at] & keep, tab[at + 1] & keep, tab[at + 2] & keep, tab[at + 3] & keep);
}

// ---------------------------------------------------------------------------------
// Grouping pass: a two-digit counting sort of the live ids by group' = i1 * p0 + i0, cut into chunks.
//   value = output row | kMultiBit when the bag holds several ids.
// ids of one (i0, i1) group end up adjacent, consecutive groups share i1.  THREE launches, no device-wide scan, no global
// atomic per id (one returning atomic per id capped the first version at ~16 G ids/s = 25 us; fire-and-forget ones run at
// the same ~16 G/s), and nothing that has to be cleared before the call (round 2: five launches, one of them a 4.5 us
// zero-fill of a few KB): every counter and flag the launches share is TAGGED with the call's epoch, a number kept on the
// device (so that a replayed HIP graph counts on) -- a word with another tag reads as "not written yet".
// The group space is cut into ranges of 2^shift groups (first digit), the id list into slices:
//   decode   every id -> (group, i2, row | multi); LDS histogram of the slice over the ranges; the slice's place inside
//            every range from one returning atomic per (slice, range) on the epoch-tagged range counters; the group is
//            stamped with the epoch ("holds an id")                                                  [one workgroup per slice]
//   spread   every id moves to its range (ranges in order, slices in arrival order inside a range): cursor[range] from a
//            scan of the range counters + the slice's place, one returning LDS atomic per id.  The prefix products of the
//            stamped groups ride in this launch (they depend on nothing the grouping computes)       [one workgroup per slice]
//   place    LDS histogram of a range's ids over its groups + scan (ids per group, starts inside the range), every id to its
//            final position with one returning LDS atomic, THEN a look back at the chunk totals the ranges before this one
//            published (epoch-tagged), then the chunk descriptors                                    [one workgroup per range]
// A chunk is <= 16 consecutive ids of one group; its 16-byte descriptor {position, group, length | flags, first
// chunk of the next group} is all the chain kernels need to walk the grouped ids -- they read descriptors with
// scalar loads and never decode a key, compare neighbours or shuffle.  The order of ids inside a group is slice
// arrival order, then arrival order inside a slice: every consumer is insensitive to it except for fp32 summation order in
// the backward.
// ---------------------------------------------------------------------------------
constexpr int kSortThreads = 1024;       // decode / spread: one workgroup per slice of the id list
#ifndef TTEMB_RANGE_THREADS
#define TTEMB_RANGE_THREADS 512
#endif
constexpr int kRangeThreads = TTEMB_RANGE_THREADS;   // place: one workgroup per range of groups
#ifndef TTEMB_SORT_SLICE
#define TTEMB_SORT_SLICE 2048
#endif
constexpr int kSliceIds = TTEMB_SORT_SLICE;   // ids per slice (target)
constexpr int kMaxSlices = 1024;
constexpr int kMaxRanges = 512;               // ranges of the group space (a power-of-two number of groups each)
constexpr int kSortBatch = 8;                 // ids per thread whose loads are in flight together
constexpr uint32_t kFirstBit = 0x100u, kLastBit = 0x200u;   // flags next to a chunk's length
constexpr int kCountBanks = 8;                               // banks of range counters (see GroupPlan::rcount)
// Epoch tag of a range counter: 38 bits above a 26-bit id count, the top bit always SET -- whatever a word held before
// (zeros, all-ones cache locations, floats), adding a count to it cannot produce a valid tag by carry.
__host__ __device__ __forceinline__ uint64_t counter_tag(uint64_t epoch) { return ((epoch & ((1ull << 37) - 1ull)) | (1ull << 37)) << 26; }
constexpr uint32_t kNoGroup = 0xffffffffu;

// A call whose output (or d_output) tensor does not fit one 32-bit window of byte offsets -- 2^24 rows or 2 GiB, what the
// chain kernels address through buffer descriptors -- or whose ids exceed what one grouping pass takes, runs as a sequence
// of PIECES: consecutive stretches of the id list, each grouped and looked up by the ordinary launches with rows counted
// from the piece's first bag (the reference chunks any call by batch_count, tt_embeddings_cuda.cu:1011-1027; SAGE.inference
// looks up all 111 M nodes of papers100M in one call, gnn_model.py:220-253).  The piece boundaries depend on `offsets`,
// which live on the device: they are computed THERE (plan_pieces_kernel), launches are sized for the largest possible
// piece, and a piece that turns out empty costs its launches and nothing else -- no host synchronisation.
struct Piece {
  long long pos0;           // first position of the id list
  long long count;          // ids of the piece (0: an unused slot of the table)
  long long rowbase;        // the bag of position pos0: rows are stored relative to it (< piece_rows by construction)
  long long zero0, zero1;   // bags whose output rows this piece clears when their length is not 1
  long long window_bytes;   // bytes of the [B][D] tensor from row `rowbase` on that the piece may address (< 2 GiB)
};

struct GroupPlan {           // device pointers into the caller's plan buffer / workspace
  const Piece* piece;        // null: the call is one piece (positions from 0, rows from 0)
  uint32_t* grp_in;          // [nnz] ungrouped: group of the id (kNoGroup past the live count)
  uint32_t* i2_in;           // [nnz] ungrouped: last index digit
  uint32_t* vals_in;         // [nnz] ungrouped: output row | kMultiBit
  uint32_t* grp_mid;         // [nnz] the same three, ordered by range
  uint32_t* i2_mid;
  uint32_t* vals_mid;
  uint32_t* shist;           // [slices][ranges] place of slice s inside range r (arrival order of the slices)
  uint64_t* rcount;          // [kCountBanks][kMaxRanges] (epoch tag | ids of the range so far, from the slices of the bank): the tag
                             //     makes a counter of an earlier call -- or whatever the memory held -- read as zero, so nothing
                             //     has to be cleared between calls.  Slices are dealt to the banks round-robin: the atomics of
                             //     200 slices on ONE counter per range were a serial chain of 200 memory-side operations
                             //     (decode 19-23 us); eight banks make it 25 (and the spread step adds eight words per range)
  uint32_t* rstart;          // [ranges + 1] first position of every range in the range-ordered arrays
  uint32_t* gstamp;          // [G] epoch (low word) of the last call that saw an id of the group (a stale or foreign word only
                             //     ever adds a prefix product nobody reads: it needs no clearing and no initial state)
  uint64_t* epochs;          // [2] the grouping pass's call counter, kept ON THE DEVICE so that a replayed HIP graph counts on:
                             //     [0] the last finished call (read by decode / spread, written by place),
                             //     [1] the running call (written by spread, read by place).  Any start value will do.
  uint64_t* rpub;            // [2 * ranges] (epoch << 24 | chunks of the range), then (epoch << 24 | its non-empty groups):
                             //     what a place workgroup publishes for the ranges after it
  uint32_t* i2s;             // [nnz] grouped: last index digit of the id
  uint32_t* vals;            // [nnz] grouped: output row | kMultiBit
  uint32_t* counts;          // [G+1] ids per group; entry G = number of groups that hold an id
  uint64_t* gpre;            // [G+3] low word: first grouped position of the group; high word: its first chunk.
                             //       entry G = (live ids, chunks), G+1 = the plan's tag, G+2 = its fault word (plan_poisoned)
  uint4* ctab;               // [max_chunks] chunk descriptors
  float* ptab;               // [G][M2*R2] prefix product P = G0[i0] . G1[i1] of every non-empty group
  float* etab;               // [nnz][ROW2] dG2 contribution rows, in grouped order
  float* dptab;              // [G][M2*R2] dP of every non-empty group
  float* g2part;             // [tiles][p2][ROW2] per-tile partial dG2
  float* g0part;             // [G][ROW0] per-group contribution to dG0: row i1 * p0 + i0 (the group's number) -- row i0 * p1 + i1 when the
                             //     chunk kernel that forms the group products wrote them (GroupFuse::parts_by_i0)
  float* g1part;             // [slices][p1][ROW1] per-slice partial dG1
  uint32_t* epi_live;        // [slices][p1] sparse form only: 1 when the slice of that i1 holds an id (its slab exists)
  uint32_t* wrows;           // wide-rank chain: [p1][stride] the rows (i0, a) of every i1 whose group holds an id (wide3_rows_kernel)
  uint32_t* wnrows;          // [p1] their number
  uint32_t* ticket;          // header word: the place step's range tickets (zeroed by the spread step of the same call)
  uint32_t use_ticket;       // the place step draws its ranges from the ticket (the id-only half of a two-phase forward: the launch
                             //     that runs beside the data-parallel step's all-reduce) instead of taking its workgroup indices
  uint32_t* fault_host;      // pinned host word (device address) a bounded wait that ran out reports to, or null
  uint32_t spin_limit;       // tries of the bounded waits of the grouping pass (ttemb_set_spin_limit; 0 = none: every wait expires)
  uint32_t keep_p;           // the caller keeps the plan (it passed a plan buffer): the forward that forms the prefix products in its
                             //     chain kernel stores them for the backward.  0 = the plan lives in the workspace and dies with the
                             //     call -- a forward nobody's backward reads (inference, a piece of a larger call): no stores
};

// ---------------------------------------------------------------------------------
// A bounded wait that runs out must not become plausible numbers (the reference checks its launches with AT_CUDA_CHECK,
// tt_embeddings_cuda.cu:1666,1742,1845; its kernels cannot time out, this grouping pass can).  Two words behind the plan's
// group table: gpre[G + 1] = the TAG of the call that built the plan (written by the place step with the chunk total),
// gpre[G + 2] = (tag | reason) of a wait that expired while it was built.  A plan whose two tags agree is POISONED: every
// kernel that would walk its chunk table leaves instead, the forward fills the output window with NaN, the finalize kernel
// emits NaN for every gradient element (fused modes: NaN weights) -- and the pinned host word makes the next API call of
// the process return TTEMB_E_HIP (ttemb_status() asks for it directly).  Tags carry the call's number with the top bit set,
// like every word of the grouping pass: what a recycled buffer held never reads as a fault.  The number is SALTED with the
// address of the workspace header it is counted in: a plan buffer may outlive a workspace, two workspaces count alike (both
// from whatever their memory held: zeros, often), and a fault word a faulted call left in the buffer must not meet a later
// call of ANOTHER workspace that happens to carry the same number.
// ---------------------------------------------------------------------------------
constexpr uint32_t kFaultTakeOver = 1u, kFaultLookBack = 2u;
__device__ __forceinline__ uint64_t plan_tag(const GroupPlan& plan, uint64_t call) {
  const uint64_t salt = (reinterpret_cast<uint64_t>(plan.epochs) >> 8) * 0x9e3779b97f4a7c15ull;
  return (((call + salt) & ((1ull << 39) - 1ull)) | (1ull << 39)) << 24;
}
__device__ __forceinline__ void report_fault(const GroupPlan& plan, uint32_t G, uint64_t call, uint32_t code) {
  __hip_atomic_store(&plan.gpre[G + 2u], plan_tag(plan, call) | code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (plan.fault_host != nullptr) __hip_atomic_store(plan.fault_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wave-uniform (scalar loads next to the chunk total every chain kernel reads first)
__device__ __forceinline__ bool plan_poisoned(const GroupPlan& plan, uint32_t G) {
  const desc_ptr w = (desc_ptr)plan.gpre;
  const uint32_t t_lo = w[2u * G + 2u], t_hi = w[2u * G + 3u], f_lo = w[2u * G + 4u], f_hi = w[2u * G + 5u];
  return t_hi == f_hi && ((t_lo ^ f_lo) >> 24) == 0u && (t_hi >> 31) != 0u;
}

// The forward of a poisoned plan: the call's whole output window reads NaN (every wavefront of the grid takes a strided
// share; the buffer descriptor clips).  `gw` of `nwaves` wavefronts.
__device__ __forceinline__ void poison_output(const GroupPlan& plan, float* out, uint32_t out_bytes, uint32_t row_floats, uint32_t gw,
                                              uint32_t nwaves, int lane) {
  if (plan.piece != nullptr) {
    out += plan.piece->rowbase * (long long)row_floats;
    out_bytes = (uint32_t)plan.piece->window_bytes;
  }
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)out_bytes, 0x00020000);
  const uint32_t qnan = 0x7fc00000u;
  u32x4 v;
  v.x = v.y = v.z = v.w = qnan;
  for (uint64_t off = ((uint64_t)gw * 64u + (uint32_t)lane) * 16u; off < out_bytes; off += (uint64_t)nwaves * 1024u)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(uint32_t)off, 0, 0);
}

__device__ __forceinline__ uint64_t pack_count(uint32_t c) {   // ids in the low word, chunks of <= kChunk ids in the high word
  return (uint64_t)c | ((uint64_t)((c + kChunk - 1) / kChunk) << 32);
}

// exclusive prefix of v over the workgroup's NT threads; also the total
template <typename T, int NT>
__device__ __forceinline__ T block_exclusive(T v, T* wave_sums, T& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T incl = v;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T up;
    if constexpr (sizeof(T) == 8) {
      const uint32_t lo_w = __shfl_up((uint32_t)incl, d, kWave), hi_w = __shfl_up((uint32_t)((uint64_t)incl >> 32), d, kWave);
      up = (T)((uint64_t)lo_w | ((uint64_t)hi_w << 32));
    } else {
      up = __shfl_up(incl, d, kWave);
    }
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) wave_sums[wave] = incl;
  __syncthreads();
  T before = 0, all = 0;
  for (int w = 0; w < NT / kWave; ++w) {
    const T x = wave_sums[w];
    if (w < wave) before += x;
    all += x;
  }
  __syncthreads();
  total = all;
  return before + incl - v;
}

__global__ __launch_bounds__(kSortThreads) void fast3_decode_kernel(
    const int64_t* __restrict__ indices, const int64_t* __restrict__ rowidx,
    const int64_t* __restrict__ offsets, uint32_t nnz, uint32_t per_slice,
    const int32_t* __restrict__ nnz_dev, int64_t B, int D, float* __restrict__ zero_out, uint32_t sentinel,
    uint32_t p0, uint32_t p1, uint32_t p2, uint32_t shift, uint32_t ranges, GroupPlan plan) {
  __shared__ uint32_t hist[kMaxRanges];
  for (uint32_t i = threadIdx.x; i < ranges; i += kSortThreads) hist[i] = 0u;
  const uint32_t epoch = (uint32_t)(plan.epochs[0] + 1ull);   // this call's number (nobody writes the word during this launch)
  __syncthreads();
  // a piece of a larger call: positions pos0 ..., rows relative to the piece's first bag, its own share of the bags to clear
  const Piece* pc = plan.piece;
  const int64_t pos0 = pc ? pc->pos0 : 0, rowbase = pc ? pc->rowbase : 0;
  const int64_t cnt = pc ? pc->count : live_count(nnz, nnz_dev);
  const int64_t cnt_all = pos0 + cnt;   // (end of the live ids as a position of the whole list: what bag_is_single compares with)
  const uint32_t s0 = blockIdx.x * per_slice;
  // every slice also owns a share of the bags: rows whose bag does not hold exactly one id are zeroed here (the
  // forward stores one-id bags and accumulates into the others; zero_out is null in the backward)
  if (zero_out != nullptr) {
    const int64_t z0 = pc ? pc->zero0 : 0, z1 = pc ? pc->zero1 : B;
    const int64_t per_b = (z1 - z0 + gridDim.x - 1) / gridDim.x;
    const int64_t b1 = z0 + (blockIdx.x + 1) * per_b < z1 ? z0 + (blockIdx.x + 1) * per_b : z1;
    for (int64_t b = z0 + blockIdx.x * per_b + threadIdx.x; b < b1; b += kSortThreads)
      if (offsets[b + 1] - offsets[b] != 1) {
        float4* o = reinterpret_cast<float4*>(zero_out + b * D);
        for (int c = 0; c * 4 < D; ++c) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  const uint32_t s1 = s0 + per_slice < nnz ? s0 + per_slice : nnz;
  constexpr int UD = 4;   // ids per thread whose loads are in flight together
  for (uint32_t base = s0 + threadIdx.x; base < s1; base += kSortThreads * UD) {
    int64_t idv[UD], o0[UD], o1[UD], rv[UD];
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const uint32_t n = base + u * kSortThreads;
      const int64_t g = pos0 + n;   // position in the whole id list
      const bool on = n < s1 && (int64_t)n < cnt;
      idv[u] = on ? indices[g] : 0;
      rv[u] = on && rowidx != nullptr ? rowidx[g] : -1;
      const bool direct = on && rowidx == nullptr && g < B;   // candidate for "bag g holds exactly id g"
      o0[u] = direct ? offsets[g] : -1;
      o1[u] = direct ? offsets[g + 1] : -1;
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const uint32_t n = base + u * kSortThreads;
      if (n >= s1) continue;
      if ((int64_t)n >= cnt) {
        plan.grp_in[n] = kNoGroup;
        continue;
      }
      int64_t id = idv[u];
      id = id < 0 ? 0 : (id >= (int64_t)sentinel ? (int64_t)sentinel - 1 : id);
      const int64_t g = pos0 + n;
      int64_t row;
      bool multi;
      if (rowidx != nullptr) {
        row = rv[u];
        multi = !bag_is_single(rowidx, offsets, g, cnt_all, row);
      } else if (o0[u] >= 0 && o0[u] <= g && g < o1[u]) {
        row = g;  // the usual case (every bag holds one id) costs two coalesced reads
        multi = o1[u] - o0[u] != 1;
      } else {    // bag of position g: the last b with offsets[b] <= g  (tt_embeddings_cuda.cu:1349-1365)
        int64_t lo = 0, hi = B;  // invariant: offsets[lo] <= g < offsets[hi]
        while (hi - lo > 1) {
          const int64_t mid = (lo + hi) >> 1;
          if (offsets[mid] <= g) lo = mid; else hi = mid;
        }
        row = lo;
        multi = offsets[row + 1] - offsets[row] != 1;
      }
      row -= rowbase;   // (0 for a one-piece call; < 2^24 and < 2 GiB / (4 D) inside a piece, by the piece table's construction)
      const uint32_t uu = (uint32_t)id;
      const uint32_t i0 = uu / (p1 * p2);
      const uint32_t rem = uu - i0 * (p1 * p2);
      const uint32_t i1 = rem / p2;
      const uint32_t group = i1 * p0 + i0;
      plan.grp_in[n] = group;
      plan.i2_in[n] = rem - i1 * p2;
      plan.vals_in[n] = (uint32_t)row | (multi ? kMultiBit : 0u);
      plan.gstamp[group] = epoch;   // "this group holds an id" (every writer stores the same word)
      atomicAdd(&hist[group >> shift], 1u);
    }
  }
  __syncthreads();
  // this slice's place inside every range: arrival order of the slices, ONE returning atomic per slice and range on a
  // counter tagged with the call's epoch.  The place step of the call before (same workspace layout) left every counter at
  // (this call's tag | 0), so the add returns the place at once.  A counter with another tag -- the first call on this
  // memory, a layout that moved -- is taken over instead: look, then compare-and-swap to (tag | own count) or, once the
  // tag is there, add.  (The blind add that found the wrong tag changed a word that held nothing of value; the tag cannot
  // change back during the launch, so an add after a matching look is safe, and a lost swap just looks again.)
  const uint64_t tag = counter_tag(plan.epochs[0] + 1ull);
  uint32_t* dst = plan.shist + (size_t)blockIdx.x * ranges;
  for (uint32_t i = threadIdx.x; i < ranges; i += kSortThreads) {
    unsigned long long* ctr = reinterpret_cast<unsigned long long*>(&plan.rcount[(blockIdx.x % kCountBanks) * kMaxRanges + i]);
    unsigned long long old = atomicAdd(ctr, (unsigned long long)hist[i]);
    uint32_t place = (uint32_t)(old & 0x3ffffffull);
    if ((old & ~0x3ffffffull) != tag) {
      place = 0;
      bool taken = false;
      for (uint32_t tries = 0; tries < plan.spin_limit; ++tries) {   // (ends after at most one round per slice of the bank; bounded all the same)
        old = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & ~0x3ffffffull) == tag) {
          place = (uint32_t)(atomicAdd(ctr, (unsigned long long)hist[i]) & 0x3ffffffull);
          taken = true;
          break;
        }
        if (atomicCAS(ctr, old, tag | hist[i]) == old) {
          place = 0;   // the first slice of this call in the range
          taken = true;
          break;
        }
      }
      if (!taken) report_fault(plan, p0 * p1, plan.epochs[0] + 1ull, kFaultTakeOver);   // the slice has no place: the plan is poisoned
    }
    dst[i] = place;
  }
}

// One slice of the spread step.  cursor[r] = ids of the ranges before r (a scan of the range counters the decode step left)
// + this slice's place inside r.
__device__ __forceinline__ void spread_slice(uint32_t s, uint32_t slices, uint32_t nnz, uint32_t per_slice, uint32_t shift,
                                             uint32_t ranges, const GroupPlan& plan, uint32_t* cursor, uint32_t* wave_sums) {
  // (every slice added to its bank's counter of every range -- zeros included -- so a bank that got a slice carries this
  // call's tag; one that got none -- fewer slices than banks -- still holds the tag the place step before left, or anything:
  // its count only counts under this call's tag)
  uint32_t tot = 0, bef = 0;
  if (threadIdx.x < ranges) {
    const uint64_t tag = counter_tag(plan.epochs[0] + 1ull);
    uint64_t w[kCountBanks];
#pragma unroll
    for (int b = 0; b < kCountBanks; ++b) w[b] = plan.rcount[b * kMaxRanges + threadIdx.x];
    bef = plan.shist[(size_t)s * ranges + threadIdx.x];
#pragma unroll
    for (int b = 0; b < kCountBanks; ++b) {
      const uint32_t c = (w[b] & ~0x3ffffffull) == tag ? (uint32_t)(w[b] & 0x3ffffffull) : 0u;
      tot += c;
      if ((uint32_t)b < s % kCountBanks) bef += c;   // the banks before this slice's come first inside the range
    }
  }
  uint32_t all;
  const uint32_t excl = block_exclusive<uint32_t, kSortThreads>(tot, wave_sums, all);
  if (threadIdx.x < ranges) {
    cursor[threadIdx.x] = excl + bef;
    if (s == 0) plan.rstart[threadIdx.x] = excl;
  }
  if (s == 0 && threadIdx.x == 0) {
    plan.rstart[ranges] = all;
    plan.epochs[1] = plan.epochs[0] + 1ull;   // the running call's number, for the place step
    *plan.ticket = 0u;                        // the place step's workgroups draw their ranges from it (nobody reads it during this launch)
  }
  __syncthreads();
  const uint32_t s0 = s * per_slice;
  const uint32_t s1 = s0 + per_slice < nnz ? s0 + per_slice : nnz;
  for (uint32_t base = s0 + threadIdx.x; base < s1; base += kSortThreads * kSortBatch) {
    uint32_t g[kSortBatch], i2v[kSortBatch], vv[kSortBatch];   // all loads of the batch first
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const uint32_t n = base + u * kSortThreads;
      g[u] = n < s1 ? plan.grp_in[n] : kNoGroup;
    }
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const uint32_t n = base + u * kSortThreads;
      const bool on = g[u] != kNoGroup;
      i2v[u] = on ? plan.i2_in[n] : 0u;
      vv[u] = on ? plan.vals_in[n] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      if (g[u] == kNoGroup) continue;
      const uint32_t dst = atomicAdd(&cursor[g[u] >> shift], 1u);
      if (dst >= nnz) continue;   // cannot happen with a consistent table; a bad counter must not become a wild store
      plan.grp_mid[dst] = g[u];
      plan.i2_mid[dst] = i2v[u];
      plan.vals_mid[dst] = vv[u];
    }
  }
}

__global__ __launch_bounds__(kSortThreads) void fast3_spread_kernel(uint32_t nnz, uint32_t per_slice, uint32_t shift,
                                                                   uint32_t ranges, GroupPlan plan) {
  __shared__ uint32_t cursor[kMaxRanges];
  __shared__ uint32_t wave_sums[kSortThreads / kWave];
  spread_slice(blockIdx.x, gridDim.x, nnz, per_slice, shift, ranges, plan, cursor, wave_sums);
}

// The place step of one range: count, place and describe in ONE launch.
//   1. LDS histogram of the range's ids over its groups, scan -> ids per group (plan.counts), first position / first
//      chunk of every group INSIDE the range; the range's totals (chunks, non-empty groups) are PUBLISHED, tagged with the
//      call's epoch, for the ranges after it;
//   2. every id takes its final position with one returning LDS atomic (positions need nothing from other ranges: the
//      range's first position is a column sum the spread step already formed);
//   3. only now the workgroup looks back: it sums what the ranges before it published -- long done by then, the scatter
//      of step 2 sits between a workgroup's own publication and its first look (a separate count launch in front of the
//      place launch cost ~5 us + the launch gap; a look-back BEFORE the scatter waited for the slowest histogram);
//   4. chunk descriptors and group starts, one thread per group (a chunk's descriptor depends on its group's numbers only).
// Which range a workgroup takes.  In a WHOLE forward (and a backward that regroups) it is the workgroup's index: workgroups
// are dispatched in index order per XCD, so the lowest-numbered unfinished workgroup is always running or next in line for a
// slot of its XCD, and it waits only on finished ones -- progress does not need the launch to be resident (it is NOT at rank
// 32, where the prefix units' registers leave one 512-thread workgroup per CU: 300 ranges on 256 slots; a validation build
// that rotates the ranges against the dispatch order expires there at once).  In the ID-ONLY HALF of a two-phase forward --
// the launch the data-parallel step runs BESIDE its RCCL all-reduce kernel (TTDataParallel.step(overlap=True)), where CUs
// are held by a kernel this library does not control -- the range is a TICKET the workgroup draws when it starts (one atomic
// on a header word the spread step zeroed): it then waits only for workgroups that have started, whatever the dispatcher did
// (rocPRIM's look-back scan draws its tile ids the same way).  The ticket is not free -- the ~270 atomics of a launch queue
// up on one address: +3 us on the grouping pass at 409 600 ids, A/B in one call; requesting the likely range's bounds next to
// the ticket instead of behind it changed nothing -- which is why the whole forward, whose launches share the GPU with nothing
// of this process, does not pay it (-DTTEMB_PLACE_TICKET_ALWAYS does).  Either way the wait is bounded: one that runs out
// poisons the plan and reports (report_fault) -- never a hung device, never a plausible wrong table.
constexpr uint64_t kEpochMask = (1ull << 40) - 1ull;
static_assert(kMaxRanges <= kRangeThreads, "the look-back reads one published word per thread");
__device__ __forceinline__ void place_range(const uint32_t ranges, uint32_t G, uint32_t shift,
                                            uint32_t nnz, uint32_t max_chunks, const GroupPlan& plan, uint32_t* lds_s,
                                            uint64_t* red) {
  // lds_s: [span] cursor | [span] first position | [span] count | [span] first chunk (inside the range)
  const uint32_t span = 1u << shift;
  uint32_t* cursor = lds_s;
  uint32_t* gfirst = cursor + span;
  uint32_t* gcount = gfirst + span;
  uint32_t* gchunk = gcount + span;
  constexpr int NWV = kRangeThreads / kWave;
  // (the ticket's round trip runs next to the epoch load and the clearing of the counters: neither needs the range)
  const uint64_t epoch = (plan.epochs[1] & (kEpochMask >> 1)) | (1ull << 39);   // (top bit set: zeros / all-ones never match)
  const bool ticketed = plan.use_ticket != 0u;   // (workgroup-uniform: a kernel argument)
  if (ticketed && threadIdx.x == 0) reinterpret_cast<uint32_t*>(red)[0] = atomicAdd(plan.ticket, 1u);
  for (uint32_t i = threadIdx.x; i < span; i += kRangeThreads) gcount[i] = 0u;
  __syncthreads();
  uint32_t range = blockIdx.x;
  if (ticketed) {
#ifdef TTEMB_TICKET_ROTATE   // (validation build: every workgroup's range differs from its index.  NOT a valid order: range 0 is
    range = (reinterpret_cast<uint32_t*>(red)[0] + 1u) % ranges;   // drawn last, so a launch whose workgroups are not all resident expires)
#else
    range = reinterpret_cast<uint32_t*>(red)[0];   // (red is the scan's scratch next: two barriers further down)
#endif
    if (range >= ranges) return;   // (cannot happen: `ranges` workgroups draw from a counter that started at 0)
  }
  const uint32_t g0 = range << shift;
  const uint32_t n0 = plan.rstart[range], n1 = plan.rstart[range + 1];
  // ---- 1. histogram, scan, publication ----
  for (uint32_t b0 = n0 + threadIdx.x; b0 < n1; b0 += kRangeThreads * kSortBatch) {
    uint32_t g[kSortBatch];
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const uint32_t n = b0 + u * kRangeThreads;
      g[u] = n < n1 ? plan.grp_mid[n] - g0 : kNoGroup;
    }
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u)
      if (g[u] < span) atomicAdd(&gcount[g[u]], 1u);
  }
  __syncthreads();
  uint64_t carry = 0;
  uint32_t live_here = 0;
  for (uint32_t b0 = 0; b0 < span; b0 += kRangeThreads) {
    const uint32_t i = b0 + threadIdx.x;
    const bool on = i < span && g0 + i < G;
    const uint32_t c = i < span ? gcount[i] : 0u;
    uint64_t total;
    const uint64_t pre = carry + block_exclusive<uint64_t, kRangeThreads>(pack_count(c), red, total);
    if (i < span) {
      gfirst[i] = cursor[i] = n0 + (uint32_t)pre;
      gchunk[i] = (uint32_t)(pre >> 32);
    }
    if (on) plan.counts[g0 + i] = c;
    carry += total;
    live_here += (uint32_t)__syncthreads_count(c != 0u);
  }
  const uint32_t chunks_here = (uint32_t)(carry >> 32);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&plan.rpub[range], (epoch << 24) | (uint64_t)chunks_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&plan.rpub[ranges + range], (epoch << 24) | (uint64_t)live_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  // ---- 2. final positions ----
  for (uint32_t b0 = n0 + threadIdx.x; b0 < n1; b0 += kRangeThreads * kSortBatch) {
    uint32_t gl[kSortBatch], i2v[kSortBatch], vv[kSortBatch];   // all loads of the batch first
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const uint32_t n = b0 + u * kRangeThreads;
      const bool on = n < n1;
      gl[u] = on ? plan.grp_mid[n] - g0 : kNoGroup;
      i2v[u] = on ? plan.i2_mid[n] : 0u;
      vv[u] = on ? plan.vals_mid[n] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      if (gl[u] >= span) continue;   // (kNoGroup, or a word that is not of this range: never trust a table with an LDS address)
      const uint32_t dst = atomicAdd(&cursor[gl[u]], 1u);
      if (dst >= nnz) continue;   // as in the spread step: never trust a counter with an address
      plan.i2s[dst] = i2v[u];
      plan.vals[dst] = vv[u];
    }
  }
  // ---- 3. look back: chunks (and, for the last range, non-empty groups) of the ranges before this one ----
  uint64_t before = 0, live_before = 0;
  if (threadIdx.x < range) {
    const bool want_live = range == ranges - 1;
    uint64_t v = 0, w = epoch << 24;
    for (uint32_t spin = 0; spin < plan.spin_limit; ++spin) {
      v = __hip_atomic_load(&plan.rpub[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (want_live) w = __hip_atomic_load(&plan.rpub[ranges + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 24) == epoch && (w >> 24) == epoch) break;
      __builtin_amdgcn_s_sleep(8);
    }
    const bool seen = (v >> 24) == epoch && (w >> 24) == epoch;
    before = seen ? (v & 0xffffffull) : 0ull;
    live_before = seen ? (w & 0xffffffull) : 0ull;
    // the wait ran out: this range's chunk numbers are wrong.  The plan is poisoned (nobody walks its chunk table) and
    // the host hears of it -- never a plausible table
    if (!seen) report_fault(plan, G, plan.epochs[1], kFaultLookBack);
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    before += __shfl_down((uint32_t)before, d, kWave);        // (< 2^32: at most 2^26 chunks in a call)
    live_before += __shfl_down((uint32_t)live_before, d, kWave);
  }
  __syncthreads();   // (red was the scan's scratch)
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = before;
    red[NWV + 1 + (threadIdx.x >> 6)] = live_before;
  }
  __syncthreads();
  uint64_t chunk_base = 0, live_all = 0;
  for (int w = 0; w < NWV; ++w) {
    chunk_base += red[w];
    live_all += red[NWV + 1 + w];
  }
  // ---- 4. group starts and chunk descriptors ----
  for (uint32_t i = threadIdx.x; i < span; i += kRangeThreads) {
    const uint32_t g = g0 + i;
    if (g >= G) continue;
    const uint32_t c = gcount[i], first = gfirst[i];
    const uint32_t ch0 = (uint32_t)chunk_base + gchunk[i];
    plan.gpre[g] = (uint64_t)first | ((uint64_t)ch0 << 32);
    const uint32_t chunks = (c + kChunk - 1) / kChunk;
    for (uint32_t k = 0; k < chunks; ++k) {
      const uint32_t rest = c - k * kChunk;
      const uint32_t len = rest < (uint32_t)kChunk ? rest : (uint32_t)kChunk;
      if (ch0 + k < max_chunks)
        plan.ctab[ch0 + k] = make_uint4(first + k * kChunk, g, len | (k == 0 ? kFirstBit : 0u) | (k + 1 == chunks ? kLastBit : 0u), ch0 + chunks);
    }
  }
  if (threadIdx.x == 0) {
    if (range == ranges - 1) {
      plan.gpre[G] = (uint64_t)n1 | ((chunk_base + chunks_here) << 32);   // (live ids, chunks)
      const uint64_t ptag = plan_tag(plan, plan.epochs[1]);
      plan.gpre[G + 1u] = ptag;   // the plan's tag: a fault word with this tag poisons it
      // a fault word of another call (a buffer that once held a faulted plan; whatever a recycled buffer held) is cleared on
      // the way.  Compare-and-swap: a report of THIS call that lands in between stays.
      unsigned long long* fw = reinterpret_cast<unsigned long long*>(&plan.gpre[G + 2u]);
      const unsigned long long seen = __hip_atomic_load(fw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((seen >> 24) != (ptag >> 24) && seen != 0ull) atomicCAS(fw, seen, 0ull);
      // how many groups hold an id at all: the backward's epilogue / finalize pick their sparse or dense form by it
      plan.counts[G] = (uint32_t)live_all + live_here;
    }
    if (range == 0) plan.epochs[0] = plan.epochs[1];   // the call is counted (nobody reads this word during this launch)
    // this range's id counter, read for the last time by the spread step: left at (the NEXT call's tag | 0), so that
    // the next decode step's first add already counts
    for (int b = 0; b < kCountBanks; ++b) plan.rcount[b * kMaxRanges + range] = counter_tag(plan.epochs[1] + 1ull);
  }
}

__global__ __launch_bounds__(kRangeThreads) void fast3_place_kernel(uint32_t nnz, uint32_t max_chunks, uint32_t G, uint32_t shift,
                                                                    GroupPlan plan) {
  extern __shared__ uint32_t lds_s[];
  __shared__ uint64_t red[2 * (kRangeThreads / kWave + 1)];
  place_range(gridDim.x, G, shift, nnz, max_chunks, plan, lds_s, red);
}

// The piece table of a call past one 32-bit row window (struct Piece): a greedy walk over the id list on the device.  A
// piece ends after `max_ids` ids or in front of the first id whose bag lies `max_rows` bags past the piece's first bag,
// whichever comes first -- so every row of a piece is < max_rows when counted from its first bag, whatever the bag lengths
// (empty bags included).  At most ceil(nnz / max_ids) + ceil(B / max_rows) pieces: the host launches that many, unused
// slots have count 0.  One wavefront: a few dozen steps of two 64-ary searches each (four round trips at 16 M bags).
// bag of position `pos` (the last b with offsets[b] <= pos), searched by a whole wavefront: 64 probes per round trip
__device__ __forceinline__ long long wave_bag_of_position(const int64_t* __restrict__ offsets, long long B, long long pos, int lane) {
  long long lo = 0, hi = B;   // invariant: offsets[lo] <= pos < offsets[hi]
  while (hi - lo > 1) {
    const long long span = hi - lo, step = (span + kWave - 1) / kWave;
    long long probe = lo + (lane + 1) * step;
    probe = probe > hi ? hi : probe;
    const bool ok = probe < hi && (long long)offsets[probe] <= pos;   // true for a prefix of the lanes (offsets do not decrease)
    const int k = __popcll(__ballot(ok));
    const long long nlo = k > 0 ? lo + k * step : lo;           // the last probe that passed (k <= 63 here: probe k = 64 is hi or past it)
    long long nhi = lo + (long long)(k + 1) * step;             // the first that did not
    nhi = nhi > hi ? hi : nhi;
    lo = nlo;
    hi = nhi;
  }
  return lo;
}

__global__ __launch_bounds__(kWave) void plan_pieces_kernel(const int64_t* __restrict__ offsets, int64_t B, int64_t nnz,
                                                            const int32_t* __restrict__ nnz_dev, long long max_ids, long long max_rows,
                                                            int D, int slots, Piece* __restrict__ tab) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  const long long total = live_count(nnz, nnz_dev);
  long long pos = 0, zr = 0;
  for (int k = 0; k < slots; ++k) {
    Piece pc;
    pc.pos0 = pos;
    pc.count = 0;
    pc.rowbase = 0;
    pc.zero0 = pc.zero1 = zr;
    pc.window_bytes = 0;
    if (pos < total) {   // (wave-uniform)
      const long long rb = wave_bag_of_position(offsets, B, pos, lane);
      const long long row_end = rb + max_rows;   // first bag this piece must not reach
      long long end = row_end < B ? (long long)offsets[row_end] : total;   // first position of that bag
      end = end > total ? total : end;
      end = end > pos + max_ids ? pos + max_ids : end;
      if (end <= pos) end = pos + 1;   // (cannot happen: the bag of `pos` starts at or before it and ends after it)
      pc.count = end - pos;
      pc.rowbase = rb;
      // bags to clear: from where the piece before stopped up to the bag the next piece starts in -- inclusive when that
      // bag began inside this piece (it is split between the two, and this one runs first)
      if (end < total) {
        const long long nb = wave_bag_of_position(offsets, B, end, lane);
        pc.zero1 = (long long)offsets[nb] == end ? nb : nb + 1;
      } else {
        pc.zero1 = B;
      }
      if (pc.zero1 < pc.zero0) pc.zero1 = pc.zero0;
      const long long rows_left = B - rb < max_rows ? B - rb : max_rows;
      pc.window_bytes = rows_left * (long long)D * 4;
      zr = pc.zero1;
      pos = end;
    }
    if (lane == 0) tab[k] = pc;
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
#ifndef TTEMB_PREFIX_GROUPS
#define TTEMB_PREFIX_GROUPS 8
#endif
constexpr int kPrefixGroups = TTEMB_PREFIX_GROUPS;   // values of i0 per wavefront
template <int Q0, int Q1, int Q2, int R1, int R2>
__device__ __forceinline__ void prefix_unit(const float* __restrict__ G0, const float* __restrict__ G1, uint32_t p0,
                                            const GroupPlan& plan, uint32_t i0_block, uint32_t i1, int lane, int stamps) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  constexpr int GM = 16 / Q0;  // whole groups per 16-row MFMA tile (q0 = 5: three groups, the last tile row idles)
  const int hi = lane >> 4, lo = lane & 15;
  const uint32_t i0_begin = i0_block * kPrefixGroups;
  const uint32_t i0_end = i0_begin + kPrefixGroups < p0 ? i0_begin + kPrefixGroups : p0;
  // any work at all?  (one lane per i0 of the slice)
  const uint32_t my = i0_begin + lane;
  // which groups hold an id: their counters when the grouping is complete (stamps == 0), else the stamps the decode step of
  // this call left -- the unit then runs next to the spread step (1: the call's number is epochs[0] + 1, stable until the
  // place step) or next to the place step (2: epochs[1], which the spread step wrote), before the counters exist
  const uint32_t want = stamps == 0 ? 0u : (uint32_t)(stamps == 1 ? plan.epochs[0] + 1ull : plan.epochs[1]);
  const bool mine = lane < kPrefixGroups && my < i0_end &&
                    (stamps != 0 ? plan.gstamp[i1 * p0 + my] == want : plan.counts[i1 * p0 + my] != 0u);
  const unsigned long long live = __ballot(mine);
  if (!live) return;
  const float* g1 = G1 + (size_t)i1 * C::ROW1;
  float bv[C::KS1][C::NT1];
#pragma unroll
  for (int s = 0; s < C::KS1; ++s)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt)
      bv[s][nt] = (C::N1 % 16 == 0 || 16 * nt + lo < C::N1) ? g1[(4 * s + hi) * C::N1 + 16 * nt + lo] : 0.f;
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
      if (row / Q0 >= GM || !((live >> gi) & 1ull)) continue;
      float* dst = plan.ptab + (size_t)(i1 * p0 + i0_begin + gi) * (C::M2 * R2);
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) {
        const int n = 16 * nt + lo;
        if (C::N1 % 16 == 0 || n < C::N1) dst[(a * Q1 + n / R2) * R2 + n % R2] = acc[nt][r];
      }
    }
  }
}

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(64) void fast3_prefix_kernel(const float* __restrict__ G0, const float* __restrict__ G1,
                                                          uint32_t p0, uint32_t stamps, GroupPlan plan) {
  prefix_unit<Q0, Q1, Q2, R1, R2>(G0, G1, p0, plan, blockIdx.x, blockIdx.y, (int)threadIdx.x, (int)stamps);
}

// The prefix products ride in the spread AND the place launch, half of the units in each (they need the cores and the decode
// step's stamps, nothing the grouping computes later): latency-bound kernels share the machine instead of queueing.  (All
// of them next to the spread step: 17.7 us for a launch whose two halves take 11 and 8.6 us alone.)  At rank 32 a unit
// keeps 64-80 registers of G1 row -- more than a 1024-thread workgroup has next to the spread step (q = 4,5,5 / 5,5,4 /
// 4,4,8 spilled 20-72 bytes per lane there; q = 8,4,4, the papers100M shape, has room): those shapes put every unit into
// the place launch (512-thread workgroups).
template <int Q0, int Q1, int Q2, int R1, int R2>
struct PrefixRides {
  static constexpr bool value = R1 < 32 || Q0 >= 8;
};

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(kSortThreads) void fast3_spread_prefix_kernel(uint32_t slices, uint32_t nnz, uint32_t per_slice,
                                                                          uint32_t shift, uint32_t ranges, uint32_t unit_end,
                                                                          const float* __restrict__ G0,
                                                                          const float* __restrict__ G1, uint32_t p0,
                                                                          uint32_t p1, GroupPlan plan) {
  __shared__ uint32_t cursor[kMaxRanges];
  __shared__ uint32_t wave_sums[kSortThreads / kWave];
  if (blockIdx.x < slices) {
    spread_slice(blockIdx.x, slices, nnz, per_slice, shift, ranges, plan, cursor, wave_sums);
    return;
  }
  const uint32_t blocks0 = (p0 + kPrefixGroups - 1) / kPrefixGroups;
  const uint32_t unit = (blockIdx.x - slices) * (kSortThreads / kWave) + (threadIdx.x >> 6);
  if (unit >= unit_end) return;
  if constexpr (PrefixRides<Q0, Q1, Q2, R1, R2>::value)
    prefix_unit<Q0, Q1, Q2, R1, R2>(G0, G1, p0, plan, unit % blocks0, unit / blocks0, (int)(threadIdx.x & 63), 1);
}

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(kRangeThreads) void fast3_place_prefix_kernel(uint32_t ranges, uint32_t nnz, uint32_t max_chunks,
                                                                          uint32_t G, uint32_t shift, uint32_t unit_begin,
                                                                          const float* __restrict__ G0,
                                                                          const float* __restrict__ G1, uint32_t p0,
                                                                          uint32_t p1, GroupPlan plan) {
  extern __shared__ uint32_t lds_s[];
  __shared__ uint64_t red[2 * (kRangeThreads / kWave + 1)];
  if (blockIdx.x < ranges) {
    place_range(ranges, G, shift, nnz, max_chunks, plan, lds_s, red);
    return;
  }
  const uint32_t blocks0 = (p0 + kPrefixGroups - 1) / kPrefixGroups;
  const uint32_t unit = unit_begin + (blockIdx.x - ranges) * (kRangeThreads / kWave) + (threadIdx.x >> 6);
  if (unit >= blocks0 * p1) return;
  prefix_unit<Q0, Q1, Q2, R1, R2>(G0, G1, p0, plan, unit % blocks0, unit / blocks0, (int)(threadIdx.x & 63), 2);
}

// ---------------------------------------------------------------------------------
// forward
//
// One wavefront takes a contiguous share of the chunk descriptors.  Everything it needs for a chunk is known two
// steps ahead, so the loop is a software pipeline with no data-dependent control flow: while chunk c is
// multiplied, the G2 rows and the prefix product of chunk c+1 are in flight into registers and the (i2, row)
// pairs of chunk c+2 are being fetched.  Lanes are tied to ids four by four (lane = 4 * id + piece): a lane
// loads the pieces j, j+4, ... of "its" id's rows and later stores the same pieces of its output row, so no
// lane ever needs another lane's index.
// ---------------------------------------------------------------------------------
// The chain kernels are persistent: a workgroup is kChainWaves INDEPENDENT wavefronts (one per SIMD: they share
// nothing and never meet at a barrier), the grid is sized to the machine and every wavefront takes an equal,
// contiguous share of the chunk table.  On gfx950 an fp32 MFMA does not co-issue with other vector work of ANY
// wavefront on its SIMD (tools/micro/mfma_valu.hip: an MFMA-only and a VALU-only partner wave take the SUM of their
// times), so a SIMD's time is the sum of everything its waves issue.  The kernels are therefore built to issue
// little besides MFMAs:
//   * per-chunk tables (the (i2, row) pairs, E rows, P / dP of the group) are addressed through buffer descriptors
//     re-based on the chunk in SCALAR registers: the per-lane offsets are kernel-lifetime constants, the descriptor's
//     size does the clipping (rows past the chunk's length read zeros / are not stored), no vector instruction
//     computes or predicates an address;
//   * rows of the caller's tensors need one multiply-add per lane and chunk (row number -> byte offset), their
//     pieces ride in the instructions' immediate offsets;
//   * MFMA tiles / K-steps that hold no id of a short chunk are skipped (wave-uniform branches around MFMA-only code).
#ifndef TTEMB_CHAIN_WAVES
#define TTEMB_CHAIN_WAVES 4
#endif
constexpr int kChainWaves = TTEMB_CHAIN_WAVES;       // wavefronts per workgroup of the chain kernels
constexpr uint32_t kOobBase = 0x80000000u;           // a byte offset past every table (fast3_fits keeps them < 2 GiB)

__device__ __forceinline__ uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
// A wavefront in its multiply phase streams MFMAs; its SIMD partner's loads, stores and LDS traffic then queue behind
// them (the oldest wave wins the issue port).  Everything that is not an MFMA section runs at raised priority, so a
// memory instruction waits for at most the MFMA in flight.
#ifdef TTEMB_NO_PRIO
#define TTEMB_PRIO(x)
#else
#define TTEMB_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(kChainWaves * 64) void fast3_forward_kernel(const float* __restrict__ G2, GroupPlan plan, uint32_t G,
                                                                         uint32_t p2, uint32_t nnz, float* __restrict__ out,
                                                                         uint32_t out_bytes) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = uniform(threadIdx.x >> 6);
  const int hi = lane >> 4, lo = lane & 15;
  const int b_l = lane >> 2, j_l = lane & 3;   // this lane's id inside a chunk, and which pieces of its rows
  float* pbuf = smem + wave * C::WAVE_FLOATS;
  float* bbuf = pbuf + C::P_FLOATS;
  float* obuf = bbuf;

  const desc_ptr ctab = (desc_ptr)plan.ctab;
  const uint32_t nchunks = ((desc_ptr)plan.gpre)[2u * G + 1u];
  const uint32_t nwaves = gridDim.x * kChainWaves, gw = blockIdx.x * kChainWaves + wave;
  if (plan_poisoned(plan, G)) {   // a bounded wait of the grouping pass ran out: NaN rows, never a walk over a wrong table
    poison_output(plan, out, out_bytes, (uint32_t)C::D, gw, nwaves, lane);
    return;
  }
  const uint32_t per = (nchunks + nwaves - 1) / nwaves;
  const uint32_t c0 = gw * per;
  if (c0 >= nchunks) return;
  const uint32_t c1 = c0 + per < nchunks ? c0 + per : nchunks;

  constexpr int F4G = C::ROW2 / 4, NLG = (F4G + 3) / 4;     // float4 pieces of a G2 row / per lane
  constexpr int D4 = C::D / 4, NLO = (D4 + 3) / 4;          // float4 pieces of an output row / per lane
  constexpr int PF = C::M2 * R2, PF4 = PF / 4, NLP = (PF4 + kWave - 1) / kWave;
  const rsrc_t r_g2 = make_rsrc(G2, p2 * (uint32_t)C::ROW2 * 4u);
  if (plan.piece != nullptr) {   // a piece of a larger call: rows count from the piece's first bag
    out += plan.piece->rowbase * (long long)C::D;
    out_bytes = (uint32_t)plan.piece->window_bytes;
  }
  const rsrc_t r_out = make_rsrc(out, out_bytes);
  const uint32_t rowpiece = 16u * (uint32_t)j_l;
  const uint32_t g_last = (F4G % 4 == 0 || j_l + 4 * (NLG - 1) < F4G) ? rowpiece + 64u * (NLG - 1) : kOobBase;
  const bool o_has_last = D4 % 4 == 0 || j_l + 4 * (NLO - 1) < D4;

  // Pipeline of one wavefront (k = the chunk being multiplied):
  //     multiply chunk k out of LDS, rows -> LDS -> registers | G2 rows of k+1: registers -> LDS | offsets of k+2 from
  //     its i2 | store the rows of k | load the G2 rows (and P) of k+2 | load the i2 of k+3 and the output rows of k+1
  // Stores are issued only after everything the next steps wait for has been consumed, so no wait sits behind a store.
  float4 pre_g[NLG], pre_p[NLP];
  auto request = [&](uint32_t row, const uint4& d) {   // row: byte offset of the lane's G2 row
#pragma unroll
    for (int k = 0; k < NLG; ++k)
      pre_g[k] = k + 1 < NLG ? buf_load4(r_g2, row + rowpiece + 64u * k) : buf_load4(r_g2, row + g_last);
    // P of a new group: the descriptor is re-based on the group's slot, empty when the chunk continues a group
    const rsrc_t r_p = make_rsrc(plan.ptab + (size_t)uniform(d.y) * PF, uniform((d.z & kFirstBit) ? (uint32_t)(PF * 4) : 0u));
#pragma unroll
    for (int it = 0; it < NLP; ++it) pre_p[it] = buf_load4(r_p, 16u * (uint32_t)lane + 1024u * it);
  };
  auto stage = [&](bool with_p) {   // registers -> LDS
    if (with_p) {   // the prefix product of a new group, as a (q0 q1) x r2 matrix (P rows are padded: b32 writes)
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;
        if (PF4 % kWave == 0 || e < PF4) {
          float* dst = pbuf + (4 * e / R2) * C::LDA + (4 * e) % R2;
          dst[0] = pre_p[it].x;
          dst[1] = pre_p[it].y;
          dst[2] = pre_p[it].z;
          dst[3] = pre_p[it].w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      if (F4G % 4 == 0 || idx < F4G) *reinterpret_cast<float4*>(bbuf + b_l * C::LDB + 4 * idx) = pre_g[k];
    }
  };

  // ---- prologue: chunk 0 into LDS, rows of chunk 1 and the i2 of chunk 2 in flight, descriptors three chunks ahead ----
  // Nothing a step requests outlives the next step: i2 of chunk c+3 and the output rows of chunk c+1 are requested at the end
  // of step c and consumed in step c+1.  (With (i2, row) pairs requested two chunks ahead the row word lived two steps, the
  // rotation of its registers copied a freshly loaded value, and that copy -- s_waitcnt vmcnt(0) at the top of every step --
  // drained the loads the step before had just issued; a descriptor loaded and used in the same step was a second wait.)
  auto fetch_i2 = [&](const uint4& d) {   // lanes past the chunk's length read 0: G2 row 0 stands in, nothing of theirs is stored
    return buf_load1u(make_rsrc(plan.i2s + uniform(d.x), uniform((d.z & 0xffu) * 4u)), 4u * (uint32_t)b_l);
  };
  auto fetch_val = [&](const uint4& d) {
    return buf_load1u(make_rsrc(plan.vals + uniform(d.x), uniform((d.z & 0xffu) * 4u)), 4u * (uint32_t)b_l);
  };
  uint4 d_cur = load_desc(ctab, c0, c1);
  uint4 d_nxt = load_desc(ctab, c0 + 1, c1);   // past c1: an empty chunk
  uint4 d_nn = load_desc(ctab, c0 + 2, c1);
  uint4 d_n3 = load_desc(ctab, c0 + 3, c1);
  uint32_t i2_nn, val_cur;
  {
#if defined(TTEMB_ABL) && (TTEMB_ABL & 256)   // (ablation 256: the first two chunks' rows are requested without waiting for their i2 -- wrong rows, timing only)
    const uint32_t i2_a = (uint32_t)b_l, i2_b = (uint32_t)b_l + 16u;
#else
    const uint32_t i2_a = fetch_i2(d_cur), i2_b = fetch_i2(d_nxt);
#endif
    uint4 d0 = d_cur;   // a share may begin inside a group: its first chunk needs P whatever its flags say
    d0.z |= kFirstBit;
    request(__umul24(i2_a, (uint32_t)(C::ROW2 * 4)), d0);   // i2 < p2 <= 4096
    stage(true);
    __builtin_amdgcn_sched_barrier(0);
    request(__umul24(i2_b, (uint32_t)(C::ROW2 * 4)), d_nxt);
  }
  i2_nn = fetch_i2(d_nn);
  val_cur = fetch_val(d_cur);

  for (uint32_t c = c0;; ++c) {
    const uint32_t len = d_cur.z & 0xffu;
    const uint32_t val = val_cur;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);

    TTEMB_PRIO(0);
    // ---- stage 2: (q0 q1 x r2) . (r2 x 16 q2): the A operand (P) is read once, then one column tile (16 of the
    //      16 q2 columns = ids 16 nt / q2 ...) at a time; tiles past the chunk's last id are skipped ----
    float av[C::MT2][C::KS2];
#pragma unroll
    for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
      for (int s = 0; s < C::KS2; ++s) av[mt][s] = pbuf[(16 * mt + lo) * C::LDA + 4 * s + hi];
    float bv[C::NT2][C::KS2];
#pragma unroll
    for (int nt = 0; nt < C::NT2; ++nt) {
      if ((uint32_t)(16 * nt) < len * Q2) {
        const int n = 16 * nt + lo;
#pragma unroll
        for (int s = 0; s < C::KS2; ++s) bv[nt][s] = bbuf[(n / Q2) * C::LDB + (4 * s + hi) * Q2 + n % Q2];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every staged-G2 read is done: the region becomes the row buffer
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < C::NT2; ++nt) {
      if ((uint32_t)(16 * nt) < len * Q2) {
        const int n = 16 * nt + lo;
        const int b = n / Q2, kk = n % Q2;
        f32x4 acc[C::MT2];   // the row tiles' accumulation chains alternate
#pragma unroll
        for (int s = 0; s < C::KS2; ++s)
#pragma unroll
          for (int mt = 0; mt < C::MT2; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], bv[nt][s], s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt], 0, 0, 0);
        // rows -> LDS, id-major.  When q0 q1 is a multiple of 4 a lane's four rows 16 mt + 4 hi + r are in or out together
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt) {
          if (16 * mt + 4 * hi + (C::M2 % 4 == 0 ? 3 : 0) < C::M2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (C::M2 % 4 == 0 || 16 * mt + 4 * hi + r < C::M2) obuf[b * C::LDO + (16 * mt + 4 * hi + r) * Q2 + kk] = acc[mt][r];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    TTEMB_PRIO(2);
    // ---- this lane's pieces of its output row: LDS -> registers ----
    float4 x[NLO];
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int idx = 4 * k + j_l < D4 ? 4 * k + j_l : D4 - 1;
      x[k] = *reinterpret_cast<const float4*>(obuf + b_l * C::LDO + 4 * idx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the row buffer has been read: the next chunk's G2 rows may land in it
    __builtin_amdgcn_sched_barrier(0);

    // ---- G2 rows (and P) of the next chunk: registers -> LDS; then the row offset of the chunk after (consumes its
    //      pair here, in front of the stores: behind them its wait would cover every store as well) ----
    stage((d_nxt.z & kFirstBit) != 0u);
    const uint32_t row_nn = __umul24(i2_nn, (uint32_t)(C::ROW2 * 4));
    asm volatile("" ::"v"(row_nn));
    __builtin_amdgcn_sched_barrier(0);

    // ---- 16-byte global stores, four lanes per row: no load is waited for behind these stores ----
    {
      const bool row_ok = (uint32_t)b_l < len;
      const bool multi = (val & kMultiBit) != 0u;
      const uint32_t row_off = __umul24(val & 0x00ffffffu, (uint32_t)(C::D * 4)) + rowpiece;   // rows < 2^24 (fast3_fits)
      const uint32_t off = row_ok && !multi ? row_off : kOobBase;   // bags with several ids accumulate below
#pragma unroll
      for (int k = 0; k < NLO; ++k) buf_store4_p<(TTEMB_NT & 4) != 0>(r_out, (k + 1 < NLO || o_has_last) ? off + 64u * k : kOobBase, x[k]);
      if (__ballot(row_ok && multi) != 0ull) {  // rare: float atomics (buffer atomics fault on an out-of-range offset
        if (row_ok && multi) {                  // instead of vanishing, so they sit in a branch)
          float* dst = out + (size_t)(val & 0x00ffffffu) * (uint32_t)C::D;
#pragma unroll
          for (int k = 0; k < NLO; ++k) {
            const int idx = 4 * k + j_l;
            if (k + 1 < NLO || o_has_last) {
              atomicAdd(dst + 4 * idx + 0, x[k].x);
              atomicAdd(dst + 4 * idx + 1, x[k].y);
              atomicAdd(dst + 4 * idx + 2, x[k].z);
              atomicAdd(dst + 4 * idx + 3, x[k].w);
            }
          }
        }
      }
    }
    if (c + 1 >= c1) break;
    // ---- loads of the chunk after next, i2 of the one after that, output rows of the next ----
    request(row_nn, d_nn);
    i2_nn = fetch_i2(d_n3);
    val_cur = fetch_val(d_nxt);
    d_cur = d_nxt;
    d_nxt = d_nn;
    d_nn = d_n3;
    d_n3 = load_desc(ctab, c + 4, c1);
  }
}


// ---------------------------------------------------------------------------------
// forward, prefix products formed IN the chain kernel: frontiers with few ids per group
//
// With 23 ids per group (the products frontier) forming P = G0[i0] . G1[i1] once per group in a launch of its own and
// reading it back is cheap; with 3 ids per group (819 200 ids on the papers100M table: 280 000 groups) that table is a
// 1.1 GB write and a 1.1 GB read around 0.4 GB of output rows -- the prefix launch took 363 us at 48 TFLOP/s and the chain
// kernel 333 us on 60 us of MFMA work.  Here a wavefront forms the P of a BATCH of 16 / q0 consecutive groups (one full
// 16-row MFMA tile: the rows of G0[i0 .. i0 + 16/q0), the B operand is G1[i1] held in registers for as long as the
// wavefront stays inside one i1 -- its share of the chunk table covers a fraction of one) when it meets the batch's first
// chunk, keeps the batch's products in LDS for the chunks that follow, and stores them to the plan's table on the way (the
// backward of the same call reads them there: fire-and-forget stores).  Groups are numbered i1 * p0 + i0, so a batch is
// p0-aligned by construction; a batch whose other groups hold no id computes their P for nothing (<= half the tile on a
// uniform papers100M frontier at 95 % occupancy: 5 %).  The A operand of a batch (q0 r1 floats per group, L2-resident) is
// requested with the G2 rows of the batch's first chunk, two steps ahead.
// ---------------------------------------------------------------------------------
template <int Q0, int Q1, int Q2, int R1, int R2>
struct PFuseCfg {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  static constexpr int GM = 16 / Q0;          // groups per batch
  // q0 > 8: one group per tile -- nothing to batch; at rank 32 only q0 = 8 (the papers100M shape: G1[i1] is 64 registers) fits
  // the two-wavefront bound: q = 4,4,8 / 5,5,4 spill 64-76 bytes per lane there, q = 4,5,5 (80 registers of G1 row) 20 bytes.
  // Those shapes keep the prefix launch (no shipped kernel has scratch: tools/kres.py --fail-on-scratch, run by the CPU tests).
#ifdef TTEMB_PFUSE_455_R32   // (A/B: the spilling q = 4,5,5 rank-32 instance, shipped in round 4)
  static constexpr bool ok = GM >= 2 && !(R1 == 32 && ((Q0 == 4 && Q1 == 4 && Q2 == 8) || (Q0 == 5 && Q1 == 5 && Q2 == 4)));
#else
  static constexpr bool ok = GM >= 2 && !(R1 == 32 && Q0 != 8);
#endif
  // A staged P is M2 rows of r2 floats, UNPADDED (a padded row cost the q = 4,4,8 rank-16 shape its third workgroup per CU):
  // the 16-byte quads of row m are XOR-swizzled by swz(m) instead, so that the A-operand reads (lane lo = row, 4 s + hi =
  // column) of eight consecutive rows fall on eight different bank groups.  Rows past M2 of the last row tile are read
  // from whatever follows the slot (inside the wavefront's region) and discarded.
  static constexpr int QR = R2 / 4;                              // quads per row
  static constexpr int RPB = R2 >= 32 ? 1 : 32 / R2;             // rows per sweep of the 32 banks
  static constexpr int SLOT_FLOATS = C::M2 * R2;
  static constexpr int WAVE_FLOATS = GM * SLOT_FLOATS + C::BO_FLOATS;
  static_assert(C::BO_FLOATS >= (C::MT2 * 16 - C::M2) * R2, "the last slot's padded rows are read inside the wavefront's region");
  __device__ static constexpr int swz(int m) { return 4 * ((m / RPB) % QR); }
};

// (two wavefronts per SIMD at least: left to itself the compiler gave the rank-32 instances 256 VGPRs + accumulation
// registers -- ONE wavefront per SIMD, 695 us on the papers100M frontier)
template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__(kChainWaves * 64, 2) void fast3_forward_pfuse_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, const float* __restrict__ G2, GroupPlan plan, uint32_t G, uint32_t p0,
    uint64_t p0_magic, uint32_t p2, uint32_t nnz, float* __restrict__ out, uint32_t out_bytes) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  using PC = PFuseCfg<Q0, Q1, Q2, R1, R2>;
  constexpr int GM = PC::GM;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = uniform(threadIdx.x >> 6);
  const int hi = lane >> 4, lo = lane & 15;
  const int b_l = lane >> 2, j_l = lane & 3;   // this lane's id inside a chunk, and which pieces of its rows
  float* pbuf = smem + wave * PC::WAVE_FLOATS;   // GM slots of SLOT_FLOATS: the prefix products of the current batch
  float* bbuf = pbuf + GM * PC::SLOT_FLOATS;
  float* obuf = bbuf;

  const desc_ptr ctab = (desc_ptr)plan.ctab;
  const uint32_t nchunks = ((desc_ptr)plan.gpre)[2u * G + 1u];
  const uint32_t nwaves = gridDim.x * kChainWaves, gw = blockIdx.x * kChainWaves + wave;
  if (plan_poisoned(plan, G)) {   // a bounded wait of the grouping pass ran out: NaN rows, never a walk over a wrong table
    poison_output(plan, out, out_bytes, (uint32_t)C::D, gw, nwaves, lane);
    return;
  }
  const uint32_t per = (nchunks + nwaves - 1) / nwaves;
  const uint32_t c0 = gw * per;
  if (c0 >= nchunks) return;
  const uint32_t c1 = c0 + per < nchunks ? c0 + per : nchunks;

  constexpr int F4G = C::ROW2 / 4, NLG = (F4G + 3) / 4;     // float4 pieces of a G2 row / per lane
  constexpr int D4 = C::D / 4, NLO = (D4 + 3) / 4;          // float4 pieces of an output row / per lane
  constexpr int PF = C::M2 * R2;
  const rsrc_t r_g2 = make_rsrc(G2, p2 * (uint32_t)C::ROW2 * 4u);
  const rsrc_t r_g0 = make_rsrc(G0, p0 * (uint32_t)C::ROW0 * 4u);
  if (plan.piece != nullptr) {   // a piece of a larger call: rows count from the piece's first bag
    out += plan.piece->rowbase * (long long)C::D;
    out_bytes = (uint32_t)plan.piece->window_bytes;
  }
  const rsrc_t r_out = make_rsrc(out, out_bytes);
  const uint32_t rowpiece = 16u * (uint32_t)j_l;
  const uint32_t g_last = (F4G % 4 == 0 || j_l + 4 * (NLG - 1) < F4G) ? rowpiece + 64u * (NLG - 1) : kOobBase;
  const bool o_has_last = D4 % 4 == 0 || j_l + 4 * (NLO - 1) < D4;
  // the P product of a batch: row lo = (group lo / q0 of the batch, core row lo % q0).  The K order is permuted: lane group
  // hi owns k = KPH hi + {0 .. KPH-1} (KPH = r1 / 4 consecutive k), MFMA step s multiplies k = KPH hi + s -- so that a
  // lane's KS1 values of its G0 row are KPH / 4 16-byte loads instead of KS1 dword loads; the G1 operand is loaded to match
  constexpr int KPH = R1 / 4;
  const uint32_t a_grp = (uint32_t)lo / Q0;                                   // (q0 = 5: row 15 belongs to no group)
  const uint32_t a_off = (uint32_t)((lo % Q0) * R1 + KPH * hi) * 4u;

  // where a chunk's group sits: its i1, the first i0 of its batch, its slot in the batch, and whether the chunk OPENS a batch
  // (the batch differs from the one requested last; chunks are met in group order, so a batch is opened once per wavefront)
  struct Grp {
    uint32_t i1, i0b, slot, first_group;
    bool fresh;
  };
  uint32_t last_batch = 0xffffffffu;
  auto locate = [&](const uint4& d) {
    const uint32_t g = uniform(d.y);
    Grp r;
    r.i1 = (uint32_t)(((uint64_t)g * p0_magic) >> 40);   // g / p0: exact for g p0 < 2^40 (the host's magic = 2^40 / p0 + 1)
    const uint32_t i0 = g - r.i1 * p0;
    r.i0b = i0 / GM * GM;
    r.slot = i0 - r.i0b;
    r.first_group = g - r.slot;
    r.fresh = uniform(d.z & 0xffu) != 0u && r.first_group != last_batch;   // (an empty chunk past the share opens nothing)
    if (r.fresh) last_batch = r.first_group;
    return r;
  };

  float4 pre_g[NLG];
  float pre_a[C::KS1];
  uint32_t pre_cnt = 0u;   // ids of "this lane's" group of the batch: the product of a sibling group WITHOUT ids is formed with the tile, not stored
  const rsrc_t r_cnt = make_rsrc(plan.counts, G * 4u);
#ifdef TTEMB_PFABL   // (ablation: the row traffic of a chunk -- G2 row loads, staging, row reads and stores, i2 / row words -- only for
                     //  every TTEMB_PFABL-th chunk: what serving 16 ids per trip instead of one group's could save; wrong rows, timing only)
  bool pf_io = true;
#else
  constexpr bool pf_io = true;
#endif
  auto request = [&](uint32_t row, const Grp& gp) {   // row: byte offset of the lane's G2 row
    if (pf_io) {
#pragma unroll
    for (int k = 0; k < NLG; ++k)
      pre_g[k] = k + 1 < NLG ? buf_load4(r_g2, row + rowpiece + 64u * k) : buf_load4(r_g2, row + g_last);
    }
    // the batch's rows of G0 when the chunk opens one (else the loads fall off the buffer: the instruction stream is fixed)
    const uint32_t i0a = gp.i0b + a_grp;
    const uint32_t base = (gp.fresh && a_grp < (uint32_t)GM && i0a < p0) ? i0a * (uint32_t)(C::ROW0 * 4) + a_off : kOobBase;
#ifndef TTEMB_PSTORE_ALL   // (A/B: the P of every group of a batch stored, with or without ids)
    pre_cnt = buf_load1u(r_cnt, (gp.fresh && a_grp < (uint32_t)GM && i0a < p0) ? (gp.first_group + a_grp) * 4u : kOobBase);
#else
    pre_cnt = 1u;
#endif
    if constexpr (KPH % 4 == 0) {
#pragma unroll
      for (int v = 0; v < KPH / 4; ++v) {
        const float4 x = buf_load4(r_g0, base + 16u * v);
        pre_a[4 * v] = x.x; pre_a[4 * v + 1] = x.y; pre_a[4 * v + 2] = x.z; pre_a[4 * v + 3] = x.w;
      }
    } else {   // rank 8: two consecutive k per lane group
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) pre_a[s] = __uint_as_float(buf_load1u(r_g0, base + 4u * s));
    }
  };
  auto stage_rows = [&]() {   // G2 rows: registers -> LDS
    if (!pf_io) return;
#pragma unroll
    for (int k = 0; k < NLG; ++k) {
      const int idx = j_l + 4 * k;
      if (F4G % 4 == 0 || idx < F4G) *reinterpret_cast<float4*>(bbuf + b_l * C::LDB + 4 * idx) = pre_g[k];
    }
  };
  float bvp[C::KS1][C::NT1];   // G1[i1] as the B operand of the P product
  uint32_t cur_i1 = 0xffffffffu;
  auto form_p = [&](const Grp& gp) {   // the chunk opens a batch: its GM prefix products -> LDS slots and the plan's table
    if (gp.i1 != cur_i1) {   // (once per i1 of the wavefront's share: a share covers a fraction of one)
      const float* g1 = G1 + (size_t)gp.i1 * C::ROW1;
#pragma unroll
      for (int s = 0; s < C::KS1; ++s)
#pragma unroll
        for (int nt = 0; nt < C::NT1; ++nt)
          bvp[s][nt] = (C::N1 % 16 == 0 || 16 * nt + lo < C::N1) ? g1[(KPH * hi + s) * C::N1 + 16 * nt + lo] : 0.f;   // k = KPH hi + s
      cur_i1 = gp.i1;
    }
    // The product is formed TRANSPOSED -- G1[i1]^T (rows n = (j, c2)) times the batch's G0 rows (columns (group, a)) -- with
    // the very same operand registers, handed to the MFMA the other way round: the accumulator of lane (hi, lo) then holds
    // P[group lo / q0][a = lo % q0][n = 16 nt + 4 hi + r], r = 0..3 -- four CONSECUTIVE c2 of one row of P, one 16-byte LDS
    // write and one 16-byte store to the plan's table per column tile (the plain product leaves four different rows per
    // lane: 4 x as many of both).
    f32x4 acc[C::NT1];
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) {
      acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < C::KS1; ++s) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bvp[s][nt], pre_a[s], acc[nt], 0, 0, 0);
    }
    const rsrc_t r_p = make_rsrc(plan.ptab + (size_t)gp.first_group * PF, plan.keep_p ? (uint32_t)(GM * PF * 4) : 0u);   // (0: every store falls off)
    const int gi = lo / Q0, a = lo % Q0;
    const bool on = gi < GM && gp.i0b + (uint32_t)gi < p0;   // (q0 = 5: column 15 is no group's; a batch at the end of an i1 may be short)
    float* slot = pbuf + (gi < GM ? gi : 0) * PC::SLOT_FLOATS;
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) {
      const int n = 16 * nt + 4 * hi;   // (N1 and r2 are multiples of 4: the four columns stay inside one row of P)
      const bool col = C::N1 % 16 == 0 || n < C::N1;
      const int m = a * Q1 + n / R2, c2 = n % R2;
      if (on && col) *reinterpret_cast<f32x4*>(slot + m * R2 + (c2 ^ PC::swz(m))) = acc[nt];
      buf_store4_p<(TTEMB_NT & 1) != 0>(r_p, (on && col && pre_cnt != 0u) ? (uint32_t)((gi * PF + m * R2 + c2) * 4) : kOobBase, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
    }
  };
  auto fetch_i2 = [&](const uint4& d) {   // lanes past the chunk's length read 0: G2 row 0 stands in, nothing of theirs is stored
    return buf_load1u(make_rsrc(plan.i2s + uniform(d.x), uniform((d.z & 0xffu) * 4u)), 4u * (uint32_t)b_l);
  };
  auto fetch_val = [&](const uint4& d) {
    return buf_load1u(make_rsrc(plan.vals + uniform(d.x), uniform((d.z & 0xffu) * 4u)), 4u * (uint32_t)b_l);
  };

  // ---- prologue: chunk 0 (its batch's products formed) in LDS, rows of chunk 1 and the i2 of chunk 2 in flight ----
  uint4 d_cur = load_desc(ctab, c0, c1);
  uint4 d_nxt = load_desc(ctab, c0 + 1, c1);   // past c1: an empty chunk
  uint4 d_nn = load_desc(ctab, c0 + 2, c1);
  uint4 d_n3 = load_desc(ctab, c0 + 3, c1);
  Grp g_cur = locate(d_cur), g_nxt = locate(d_nxt);
  uint32_t i2_nn, val_cur;
  {
    const uint32_t i2_a = fetch_i2(d_cur), i2_b = fetch_i2(d_nxt);
    request(__umul24(i2_a, (uint32_t)(C::ROW2 * 4)), g_cur);   // i2 < p2 <= 4096
    stage_rows();
    form_p(g_cur);   // (the share's first chunk always opens a batch)
    __builtin_amdgcn_sched_barrier(0);
    request(__umul24(i2_b, (uint32_t)(C::ROW2 * 4)), g_nxt);
  }
  i2_nn = fetch_i2(d_nn);
  val_cur = fetch_val(d_cur);
  Grp g_nn = locate(d_nn);

  for (uint32_t c = c0;; ++c) {
    const uint32_t len = d_cur.z & 0xffu;
    const uint32_t val = val_cur;
#ifdef TTEMB_PFABL
    pf_io = (c - c0) % TTEMB_PFABL == 0;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);

    TTEMB_PRIO(0);
    // ---- stage 2: (q0 q1 x r2) . (r2 x 16 q2), P from the group's slot of the batch ----
    // (the K order of this product is permuted as well: lane group hi owns c2 = KPH2 hi + {0 .. KPH2-1}, so that a lane's
    //  KS2 values of its row of P are 16-byte LDS reads -- 4 instead of 16 per chunk at rank 32 -- and the G2 operand follows)
    constexpr int KPH2 = R2 / 4;
    const float* pslot = pbuf + g_cur.slot * PC::SLOT_FLOATS;
    float av[C::MT2][C::KS2];
#pragma unroll
    for (int mt = 0; mt < C::MT2; ++mt) {
      const int m = 16 * mt + lo;
      if constexpr (KPH2 % 4 == 0) {
#pragma unroll
        for (int v = 0; v < KPH2 / 4; ++v) {
          const float4 x = *reinterpret_cast<const float4*>(pslot + m * R2 + ((KPH2 * hi + 4 * v) ^ PC::swz(m)));
          av[mt][4 * v] = x.x; av[mt][4 * v + 1] = x.y; av[mt][4 * v + 2] = x.z; av[mt][4 * v + 3] = x.w;
        }
      } else {   // rank 8: two consecutive c2 per lane group, inside one 16-byte quad
#pragma unroll
        for (int s = 0; s < C::KS2; ++s) av[mt][s] = pslot[m * R2 + (((KPH2 * hi + s) & ~3) ^ PC::swz(m)) + ((KPH2 * hi + s) & 3)];
      }
    }
    float bv[C::NT2][C::KS2];
#pragma unroll
    for (int nt = 0; nt < C::NT2; ++nt) {
      if ((uint32_t)(16 * nt) < len * Q2) {
        const int n = 16 * nt + lo;
#pragma unroll
        for (int s = 0; s < C::KS2; ++s) bv[nt][s] = bbuf[(n / Q2) * C::LDB + (KPH2 * hi + s) * Q2 + n % Q2];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every staged-G2 read is done: the region becomes the row buffer
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < C::NT2; ++nt) {
      if ((uint32_t)(16 * nt) < len * Q2) {
        const int n = 16 * nt + lo;
        const int b = n / Q2, kk = n % Q2;
        f32x4 acc[C::MT2];
#pragma unroll
        for (int s = 0; s < C::KS2; ++s)
#pragma unroll
          for (int mt = 0; mt < C::MT2; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][s], bv[nt][s], s == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt) {
          if (16 * mt + 4 * hi + (C::M2 % 4 == 0 ? 3 : 0) < C::M2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (C::M2 % 4 == 0 || 16 * mt + 4 * hi + r < C::M2) obuf[b * C::LDO + (16 * mt + 4 * hi + r) * Q2 + kk] = acc[mt][r];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    TTEMB_PRIO(2);
    // ---- this lane's pieces of its output row: LDS -> registers ----
    float4 x[NLO];
    if (pf_io) {
#pragma unroll
    for (int k = 0; k < NLO; ++k) {
      const int idx = 4 * k + j_l < D4 ? 4 * k + j_l : D4 - 1;
      x[k] = *reinterpret_cast<const float4*>(obuf + b_l * C::LDO + 4 * idx);
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // the row buffer has been read: the next chunk's G2 rows may land in it
    __builtin_amdgcn_sched_barrier(0);

    // ---- G2 rows of the next chunk: registers -> LDS; the row offset of the chunk after and this chunk's row word are
    //      consumed HERE, so that no load is outstanding across the branch below (its stores would sit in front of their wait) ----
    stage_rows();
    const uint32_t row_nn = __umul24(i2_nn, (uint32_t)(C::ROW2 * 4));
    asm volatile("" ::"v"(row_nn), "v"(val));
    __builtin_amdgcn_sched_barrier(0);

    // ---- 16-byte global stores, four lanes per row ----
    if (pf_io) {
      const bool row_ok = (uint32_t)b_l < len;
      const bool multi = (val & kMultiBit) != 0u;
      const uint32_t row_off = __umul24(val & 0x00ffffffu, (uint32_t)(C::D * 4)) + rowpiece;   // rows < 2^24 (fast3_fits)
      const uint32_t off = row_ok && !multi ? row_off : kOobBase;   // bags with several ids accumulate below
#pragma unroll
      for (int k = 0; k < NLO; ++k) buf_store4_p<(TTEMB_NT & 2) != 0>(r_out, (k + 1 < NLO || o_has_last) ? off + 64u * k : kOobBase, x[k]);
      if (__ballot(row_ok && multi) != 0ull) {  // rare: float atomics
        if (row_ok && multi) {
          float* dst = out + (size_t)(val & 0x00ffffffu) * (uint32_t)C::D;
#pragma unroll
          for (int k = 0; k < NLO; ++k) {
            const int idx = 4 * k + j_l;
            if (k + 1 < NLO || o_has_last) {
              atomicAdd(dst + 4 * idx + 0, x[k].x);
              atomicAdd(dst + 4 * idx + 1, x[k].y);
              atomicAdd(dst + 4 * idx + 2, x[k].z);
              atomicAdd(dst + 4 * idx + 3, x[k].w);
            }
          }
        }
      }
    }
    // ---- the next chunk opens a batch: its prefix products (after the row stores: the rows' registers are free by now) ----
    __builtin_amdgcn_sched_barrier(0);
    if (g_nxt.fresh) {   // wave-uniform: the next chunk opens a batch
      asm volatile("; a batch begins" ::: "memory");
      form_p(g_nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 >= c1) break;
    // ---- loads of the chunk after next, i2 of the one after that, output rows of the next ----
    request(row_nn, g_nn);
    if (pf_io) {
    i2_nn = fetch_i2(d_n3);
    val_cur = fetch_val(d_nxt);
    }
    d_cur = d_nxt;
    g_cur = g_nxt;
    d_nxt = d_nn;
    g_nxt = g_nn;
    d_nn = d_n3;
    g_nn = locate(d_nn);
    d_n3 = load_desc(ctab, c + 4, c1);
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
//        dG1[i1] += G0[i0]^T . dP  (registers over a slice of i0, per-slice slabs),  dG0[i0] += dP . G1[i1]^T.
//  D. finalize: dG2 / dG1 = sum of slabs, dG0 = sum of per-group parts.
// LDS float atomics cost ~160 LDS cycles per wave-instruction on gfx950 (measured), global
// ones ~1.3 TB/s chip-wide; a store pass + per-destination sum pass is several times cheaper.
// ---------------------------------------------------------------------------------
// FUSE: the dG2 reduction happens inside this kernel and the E table never exists.  The workgroup is then kFuseWaves = 8
// wavefronts that walk their shares in lock step: every round each wavefront multiplies one chunk and leaves its E rows
// (16 x r2 q2 floats) in LDS, where its staged G2 rows were, and links every row into the list of its i2 (one LDS
// exchange per row: head[i2] <- row, next[row] <- old head).  After a barrier wavefront w adds the rows whose
// i2 = w (mod 8) into "its" rows of the workgroup's dG2 slab, which it keeps in REGISTERS for the whole kernel: row
// i2 = 8 k + w sits in register quad k / GPW of lane group k % GPW (a row is r2 q2 / 4 lanes wide, GPW rows side by
// side).  A lane group walks the lists of its own rows -- quad by quad, so every register index is static -- and adds
// plain: no float atomics, no slab in LDS, no scalar dispatch, and a repeated id is simply a list of two rows.  At the
// end every wavefront stores its rows of the workgroup's slab (what the reduce kernel writes per tile in the unfused
// form); fast3_finalize_kernel adds the slabs as before.
#ifndef TTEMB_FUSE_WAVES
#define TTEMB_FUSE_WAVES 4
#endif
// Four wavefronts (one per SIMD) and two such workgroups per CU: the lock step then never binds two wavefronts of one
// SIMD -- with eight in one workgroup both partners multiplied, waited for memory and reduced at the same moments and
// the kernel took twice as long -- while the two workgroups drift into complementary phases as independent waves do.
constexpr int kFuseWaves = TTEMB_FUSE_WAVES;   // a power of two
constexpr int kFuseQuads = 48 / kFuseWaves;    // register quads of slab rows per wavefront: p2 <= kFuseWaves * GPW * kFuseQuads
#ifndef TTEMB_FUSE_BATCH
#define TTEMB_FUSE_BATCH 6
#endif
constexpr int kFuseBatch = TTEMB_FUSE_BATCH;   // quads whose list heads / first rows are read together
constexpr int kFuseSub = 2;                    // chunks a wavefront multiplies between two reductions (one more E region each)
static_assert(kFuseQuads % kFuseBatch == 0 && kFuseBatch % 6 == 0, "quads are handled in whole batches");
// Shapes whose dP is ONE row tile (q0 q1 <= 16) hold half the accumulators and A operands of the others: their wavefronts
// have registers for half as many slab quads again, which is what a wide row (r2 q2 = 128: two rows per register quad
// across the lanes) needs to keep 140 slab rows in four wavefronts (q = 4,4,8 at rank 16, the arxiv shape of the scripts).
constexpr int fuse_quads(int m2, int row2) { return (m2 <= 16 && row2 >= 128) ? kFuseQuads + kFuseBatch : kFuseQuads; }
// Which (q, rank) shapes have a fused form at all (the rest of the rule -- p2 against the register quads, the CU's LDS --
// depends on the table and is checked per call by fused_dg2()): rank <= 16 (at rank 32 operands + slab rows spill), and
// at least two slab rows side by side in a wavefront (a row of r2 q2 > 128 floats takes more than half the lanes: one row
// per register quad, and the kernel spilled 176-208 bytes per lane on the lifted 2-core shapes q = 8,16 / 10,10).
// FUSE = true is instantiated for these shapes only.
constexpr bool fuse_shape(int r2, int row2) { return r2 <= 16 && row2 % 4 == 0 && row2 / 4 <= kWave / 2; }

// GF ("group products fused", frontiers with few ids per group): the two per-group products of fast3_group_epilogue_kernel
//        dG0 part[g] = dP[g] (q0 x q1 r2) . G1[i1]^T,      dG1[i1] += G0[i0]^T (r1 x q0) . dP[g]
// are formed HERE, at the moment a group's last chunk has been multiplied and its dP is still in this wavefront's
// accumulators -- the dP table (q0 q1 r2 floats per group: 1.08 GB written and read back around 0.43 GB of algorithmic bytes
// on the papers100M frontier of 819 200 ids, where three ids share a group) never exists, and the epilogue launch is gone.
// What leaves the kernel per group is its dG0 part (q0 r1 floats instead of q0 q1 r2); dG1[i1] accumulates in MFMA registers
// for as long as the wavefront stays inside one i1 (its share of the chunk table covers a fraction of one: the fact
// fast3_forward_pfuse_kernel uses to keep G1[i1] in registers) and is added to ONE zero-filled slab with float atomics when
// the i1 changes -- a few thousand flushes per launch.  Groups are batched 16 / q0 at a time (consecutive i0 of one i1: one
// full 16-row MFMA tile), their dP rows meet in an LDS slot; a batch is multiplied when the wavefront's next chunk belongs to
// another batch.  G1[i1] as the B operand of the dG0 product stays in registers, K-permuted so that it is loaded with
// 16-byte pieces (lane group hi owns n = (N1/4) hi + s); the batch's G0 rows are requested at the top of the iteration
// that multiplies the batch's last chunk (an instruction of every iteration: off the buffer when no batch ends).
#ifndef TTEMB_G0T
#define TTEMB_G0T 1   // (0: parts in group order, the round-5 first form -- finalize 78.4 against 74.8 us at papers100M, chunk kernel equal)
#endif
constexpr bool kPartsByI0 = TTEMB_G0T != 0;
struct GroupFuse {
  const float* G0;
  const float* G1;
  float* dg1;          // [p1][ROW1], zero-filled by the host: dG1 (added with float atomics, one flush per (wavefront, i1))
  uint32_t p0;
  uint64_t p0_magic;   // g / p0 = (g * magic) >> 40
  uint32_t p1;
  uint32_t parts_by_i0;   // the dG0 part of group (i1, i0) is stored at row i0 * p1 + i1 (the p1 parts the finalize kernel sums for one
                          // i0 are then consecutive kilobytes) instead of i1 * p0 + i0
};

template <int Q0, int Q1, int Q2, int R1, int R2, bool FUSE, bool GF = false>
__global__ __launch_bounds__((FUSE ? kFuseWaves : kChainWaves) * 64, FUSE ? 8 / kFuseWaves : 1) void fast3_bwd_chunk_kernel(
    const float* __restrict__ G2, uint32_t G, uint32_t p2, uint32_t nnz, const float* __restrict__ d_out, uint32_t dout_bytes,
    GroupPlan plan, GroupFuse gf) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  static_assert(!(FUSE && GF), "the in-kernel dG2 reduction and the in-kernel group products are separate forms");
  constexpr int NW = FUSE ? kFuseWaves : kChainWaves;
  constexpr int GM = 16 / Q0;             // GF: groups per batch
  constexpr int LDD = C::N1 + 4;          // GF: row stride of the batch's stacked dP rows [rho][n] (16-byte aligned rows)
  constexpr int GF_FLOATS = GF ? 16 * LDD : 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const uint32_t wave = uniform(threadIdx.x >> 6);
  const int hi = lane >> 4, lo = lane & 15;
  const int b_l = lane >> 2, j_l = lane & 3;   // this lane's id inside a chunk, and which pieces of its rows
  // FUSE: a second region for E rows behind the staging regions -- a wavefront multiplies kFuseSub = 2 chunks between two
  // reductions (the barriers and the fixed part of the list walk are paid once per two chunks); the first chunk's rows
  // wait there, the second's take the place of its staged G2 rows
  constexpr int WF = C::PB_FLOATS + C::BB2_FLOATS + C::OB_FLOATS + (FUSE ? kChunk * C::ROW2 : 0) + GF_FLOATS;
  float* pbuf = smem + wave * WF;
  // FUSE: [kFuseWaves] "has a chunk after this one" | [kFuseWaves * 16] list entry of every row of the round: {byte offset
  // of the row in LDS, byte offset of the next entry of its list} | [heads] byte offset of the first entry of every i2's
  // list.  Lists are chased by LDS byte offsets (relative to smem), so walking one costs no address arithmetic.
  // An empty list is the list of the NIL entry: {a row of zeros, NIL itself}.  Walking a list therefore needs no test at
  // all -- lanes whose list is empty or has ended add zeros -- and a head is reset by storing NIL.
  float* f_zero = smem + NW * WF;                                      // [ROW2] zeros
  uint32_t* f_more = reinterpret_cast<uint32_t*>(f_zero + C::ROW2);
  constexpr int kRoundRows = kFuseWaves * kFuseSub * 16;
  uint2* f_tab = reinterpret_cast<uint2*>(f_more + kFuseWaves);        // [kRoundRows + 1]: the last one is NIL
  uint32_t* f_head = reinterpret_cast<uint32_t*>(f_tab + kRoundRows + 1);   // [heads + 1]: the last one is for idle lanes
  char* const lds0 = reinterpret_cast<char*>(smem);
  const uint32_t f_nil = (uint32_t)(reinterpret_cast<char*>(f_tab + kRoundRows) - lds0);
  // this lane's row of the first / second chunk of a round: its list entry and its place (E region behind the staging
  // regions / over the staged G2 rows)
  const uint32_t my_entry0 = (uint32_t)(reinterpret_cast<char*>(f_tab + (wave * kFuseSub + 0) * 16 + b_l) - lds0);
  const uint32_t my_entry1 = (uint32_t)(reinterpret_cast<char*>(f_tab + (wave * kFuseSub + 1) * 16 + b_l) - lds0);
  const uint32_t my_row0 = (uint32_t)((wave * WF + C::PB_FLOATS + C::BB2_FLOATS + C::OB_FLOATS + b_l * C::ROW2) * sizeof(float));
  const uint32_t my_row1 = (uint32_t)((wave * WF + C::PB_FLOATS + b_l * C::ROW2) * sizeof(float));
  float* ebuf = pbuf + C::PB_FLOATS + C::BB2_FLOATS + C::OB_FLOATS;
  float* const gslot = ebuf;   // GF: the batch's dP rows (FUSE and GF exclude each other: the region behind the staging regions)
  uint32_t sub = 0;   // which chunk of the round is being multiplied
  float* bbuf = pbuf + C::PB_FLOATS;   // staged G2 rows
  float* dbuf = bbuf + C::BB2_FLOATS;  // staged d_output rows

  // This wavefront's share of the chunk table.  It owns the groups whose FIRST chunk lies in its share: it skips
  // the tail of a group that began earlier and follows its last group to the end, so every group is handled by
  // exactly one wavefront and its dP needs no partial sums.
  const desc_ptr ctab = (desc_ptr)plan.ctab;
  const uint32_t nchunks = ((desc_ptr)plan.gpre)[2u * G + 1u];
  const uint32_t nwaves = gridDim.x * NW, gw = blockIdx.x * NW + wave;
  if (plan_poisoned(plan, G)) return;   // (the whole grid: the finalize kernel then emits NaN for every gradient element)
  const uint32_t per = (nchunks + nwaves - 1) / nwaves;
  const uint32_t c0 = gw * per;
  bool have = c0 < nchunks;   // a wavefront without work leaves (FUSE: keeps meeting the barriers with empty chunks)
  const uint32_t c1 = c0 + per < nchunks ? c0 + per : nchunks;
  uint32_t c = c0;
  if (have && !(ctab[4u * c0 + 2u] & kFirstBit)) {
    c = ctab[4u * c0 + 3u];  // first chunk of the next group
    have = c < c1;
  }
  if (!FUSE && !have) return;

  constexpr int F4G = C::ROW2 / 4, NLG = (F4G + 3) / 4;   // float4 pieces of a G2 row / per lane (4 lanes per id)
  constexpr int F4D = C::D / 4, NLD = (F4D + 3) / 4;      // float4 pieces of a d_output row / per lane
  constexpr int PF = C::M2 * R2, PF4 = PF / 4, NLP = (PF4 + kWave - 1) / kWave;
  const rsrc_t r_g2 = make_rsrc(G2, p2 * (uint32_t)C::ROW2 * 4u);
  if (plan.piece != nullptr) {   // a piece of a larger call: rows count from the piece's first bag
    d_out += plan.piece->rowbase * (long long)C::D;
    dout_bytes = (uint32_t)plan.piece->window_bytes;
  }
  const rsrc_t r_do = make_rsrc(d_out, dout_bytes);

  // ---- kernel-lifetime lane constants ----
  // LDS offsets of the MFMA operands.  dP step (b4, kk): lane group `hi` contributes id b = 4 b4 + hi, column kk of it
  // When the last row tile of dP holds at most four rows (q0 q1 = 20: rows 16..19) it is multiplied by the 16-block
  // 4x4x1 MFMA instead of a padded 16x16x4 one: 8 cycles instead of 32 for the same K-step.  Block (hi, lo / 4) of that
  // instruction takes A = dO[id hi][row 16 (MT2-1) + lo % 4] and B = the SAME register the wide tiles use
  // (G2[id hi][c2 = lo]), and leaves in lane (hi, lo) the sum over "its" id of every K-step for dP[row][c2 = lo]: the
  // four lane groups are added when the group's dP is stored.
  constexpr int kTailRows = C::M2 - 16 * (C::MT2 - 1);
  constexpr bool kNarrowTail = C::MT2 > 1 && kTailRows <= 4;
  int offA[C::MT2], offB[C::RT2], offE[C::NT2];
#pragma unroll
  for (int mt = 0; mt < C::MT2; ++mt) {
    int m = 16 * mt + lo < C::M2 ? 16 * mt + lo : C::M2 - 1;  // rows past M2 are discarded
    if (kNarrowTail && mt == C::MT2 - 1) m = 16 * mt + (lo & 3) < C::M2 ? 16 * mt + (lo & 3) : C::M2 - 1;
    offA[mt] = hi * C::LDOB + m * Q2;
  }
#pragma unroll
  for (int t = 0; t < C::RT2; ++t) offB[t] = hi * C::LDBB + ((16 * t + lo) % R2) * Q2;
#pragma unroll
  for (int nt = 0; nt < C::NT2; ++nt) {
    const int col = 16 * nt + lo;
    offE[nt] = (col / Q2) * C::LDOB + col % Q2 + hi * Q2;
  }
  // byte offsets inside a chunk's E rows ([id][kk][c2]: col * r2 + c2, 16-byte pieces) and inside a group's dP
  // ([m][c2]); a lane that holds no element of the table gets an offset past it
  uint32_t eoff[C::RT2], dpoff[C::MT2][C::RT2];
#pragma unroll
  for (int t = 0; t < C::RT2; ++t) eoff[t] = 16 * t + 4 * hi < R2 ? (uint32_t)(lo * R2 + 16 * t + 4 * hi) * 4u : kOobBase;
#pragma unroll
  for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
    for (int t = 0; t < C::RT2; ++t)
      dpoff[mt][t] = 16 * t + lo < R2 ? (uint32_t)((16 * mt + 4 * hi) * R2 + 16 * t + lo) * 4u : kOobBase;   // rows >= M2 fall off the group's slot
  const uint32_t rowpiece = 16u * (uint32_t)j_l;
  // the last piece round of a row whose float4 count is not a multiple of 4 exists for some lanes only
  const uint32_t g_last = (F4G % 4 == 0 || j_l + 4 * (NLG - 1) < F4G) ? rowpiece + 64u * (NLG - 1) : kOobBase;
  const bool d_has_last = F4D % 4 == 0 || j_l + 4 * (NLD - 1) < F4D;
  // FUSE: an E row in LDS is [kk][c2] like a row of the E table, its 16-byte quads XOR-swizzled by kk (the accumulator
  // layout would otherwise put four lanes of a write on one bank group).  Producer: where this lane's accumulators go;
  // consumer: which quad of a row this lane adds (lane group = which of the GPW rows of a register quad)
  constexpr int LPR = C::ROW2 / 4, GPW = kWave / LPR, QK = R2 / 4;   // lanes per row, rows side by side, quads per kk
  int ewr[C::NT2][C::RT2];
#pragma unroll
  for (int nt = 0; nt < C::NT2; ++nt)
#pragma unroll
    for (int t = 0; t < C::RT2; ++t) {
      const int col = 16 * nt + lo, kk = col % Q2, quad = 4 * t + hi;
      ewr[nt][t] = quad < QK ? (col / Q2) * C::ROW2 + kk * R2 + 4 * (quad ^ (kk & (QK - 1))) : -1;
    }
  const int s_grp = lane / LPR, s_piece = lane % LPR;
  const int s_kk = s_piece / QK;
  const int rd_off = s_kk * R2 + 4 * ((s_piece % QK) ^ (s_kk & (QK - 1)));   // float offset of this lane's quad inside a row
  constexpr int FQ = fuse_quads(C::M2, C::ROW2);
  constexpr int FB = FQ % kFuseBatch == 0 ? kFuseBatch : 6;   // quads walked together (fuse_quads() is a multiple of 6)
  f32x4 slab[FQ];
  if constexpr (FUSE) {
#pragma unroll
    for (int q = 0; q < FQ; ++q) slab[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i <= kFuseWaves * GPW * FQ; i += kFuseWaves * kWave) f_head[i] = f_nil;
    for (int i = threadIdx.x; i < C::ROW2; i += kFuseWaves * kWave) f_zero[i] = 0.f;
    if (threadIdx.x == 0) f_tab[kRoundRows] = make_uint2((uint32_t)(reinterpret_cast<char*>(f_zero) - lds0), f_nil);
    __syncthreads();
  }

  // Pipeline of one wavefront (k = the chunk being multiplied):
  //     multiply chunk k out of LDS | rows of k+1: registers -> LDS | offsets of k+2 from its (i2, row) pairs |
  //     store E / dP of k | load rows of k+2 | load the (i2, row) pairs of k+3
  // Stores are issued only after everything the next steps wait for has been consumed, so no wait ever sits
  // behind a store; rows travel through registers and are in flight during one whole multiply.
  struct Offs {
    uint32_t row, grow;   // byte offsets of this lane's first pieces: G2 row, d_output row (past the table = no id)
  };
  auto fetch_meta = [&](const uint4& d, uint32_t& i2, uint32_t& val) {
    const uint32_t len = d.z & 0xffu;   // lanes past the chunk's length read 0: G2 row 0 stands in, its d_output row is zeros
    const uint32_t at = uniform(d.x), bytes = uniform(len * 4u);   // (loop-carried words may sit in vector registers)
    i2 = buf_load1u(make_rsrc(plan.i2s + at, bytes), 4u * (uint32_t)b_l);
    val = buf_load1u(make_rsrc(plan.vals + at, bytes), 4u * (uint32_t)b_l);
  };
  auto offsets = [&](const uint4& d, uint32_t i2, uint32_t val) {
    Offs o;
    o.row = __umul24(i2, (uint32_t)(C::ROW2 * 4)) + rowpiece;                                 // i2 < p2 <= 4096
    const uint32_t g = __umul24(val & 0x00ffffffu, (uint32_t)(C::D * 4));                     // rows < 2^24 (fast3_fits); drops kMultiBit
#if defined(TTEMB_ABL) && (TTEMB_ABL & 2)
    o.grow = (uint32_t)b_l < (d.z & 0xffu) ? __umul24(val & 1023u, (uint32_t)(C::D * 4)) : kOobBase;   // ablation: cache-resident rows
#else
    o.grow = (uint32_t)b_l < (d.z & 0xffu) ? g : kOobBase;
#endif
    return o;
  };
  float4 pre_g[NLG], pre_d[NLD], pre_p[NLP];
  auto request = [&](const Offs& o, const uint4& d) {
#pragma unroll
    for (int k = 0; k < NLG; ++k)
      pre_g[k] = k + 1 < NLG ? buf_load4(r_g2, o.row + 64u * k) : buf_load4(r_g2, o.row - rowpiece + g_last);
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      pre_d[k] = buf_load4(r_do, (k + 1 < NLD || d_has_last) ? o.grow + rowpiece + 64u * k : kOobBase);   // o.grow may be kOobBase itself
    // P of a new group: the descriptor is re-based on the group's slot, empty when the chunk continues a group
#if defined(TTEMB_ABL) && (TTEMB_ABL & 1024)   // (ablation 1024: P is not read -- zeros; timing only)
    const rsrc_t r_p = make_rsrc(plan.ptab + (size_t)uniform(d.y) * PF, 0u);
#else
    const rsrc_t r_p = make_rsrc(plan.ptab + (size_t)uniform(d.y) * PF, uniform((d.z & kFirstBit) ? (uint32_t)(PF * 4) : 0u));
#endif
#pragma unroll
    for (int it = 0; it < NLP; ++it) pre_p[it] = buf_load4(r_p, 16u * (uint32_t)lane + 1024u * it);
  };
  auto stage = [&](bool with_p) {   // registers -> LDS
    if (with_p) {
#pragma unroll
      for (int it = 0; it < NLP; ++it) {
        const int e = it * kWave + lane;  // float4 number e of the (q0 q1) x r2 matrix
        if (PF4 % kWave == 0 || e < PF4)
          *reinterpret_cast<float4*>(pbuf + (4 * e / R2) * C::LDPB + (4 * e) % R2) = pre_p[it];
      }
    }
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
  };

  f32x4 dp[C::MT2][C::RT2];
  // ---- GF: state of the group products ----
  constexpr int KS3 = C::N1 / 4, KPH3 = C::N1 / 4;          // k-steps of the dG0 product; lane group hi owns n = KPH3 hi + s
  f32x4 g1acc[GF ? C::RT1 : 1][GF ? C::NT1 : 1];            // dG1[cur_i1] of this wavefront's groups so far
  float gf_g1[GF ? KS3 : 1][GF ? C::RT1 : 1];               // G1[cur_i1] as the B operand of the dG0 product
  float gf_g0[4][GF ? C::RT1 : 1];                          // the G0 rows of the batch that ends in this iteration (A of the dG1 product)
  uint32_t gf_i1 = 0xffffffffu, gf_batch = 0xffffffffu, gf_mask = 0u;   // i1 of g1acc / gf_g1; first group of the open batch; its staged groups
  const rsrc_t r_g0 = make_rsrc(gf.G0, GF ? gf.p0 * (uint32_t)C::ROW0 * 4u : 0u);
  const rsrc_t r_part = make_rsrc(plan.g0part, GF ? G * (uint32_t)C::ROW0 * 4u : 0u);
  struct Batch {
    uint32_t i1, i0b, first_group;
  };
  auto batch_of = [&](uint32_t g) {   // (scalar arithmetic: g is wave-uniform)
    Batch b;
    b.i1 = (uint32_t)(((uint64_t)g * gf.p0_magic) >> 40);   // g / p0: exact for g p0 < 2^40
    const uint32_t i0 = g - b.i1 * gf.p0;
    b.i0b = i0 / GM * GM;
    b.first_group = g - (i0 - b.i0b);
    return b;
  };
  auto flush_g1 = [&]() {   // dG1[gf_i1] += this wavefront's sums (row c1 = 16 t + 4 hi + r, column n = 16 nt + lo), then zeros
    if constexpr (GF) {
      float* dst = gf.dg1 + (size_t)gf_i1 * C::ROW1;
#pragma unroll
      for (int t = 0; t < C::RT1; ++t)
#pragma unroll
        for (int nt = 0; nt < C::NT1; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c1 = 16 * t + 4 * hi + r;
            if (c1 < R1 && (C::N1 % 16 == 0 || 16 * nt + lo < C::N1)) atomicAdd(dst + c1 * C::N1 + 16 * nt + lo, g1acc[t][nt][r]);
          }
          g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
  };
  if constexpr (GF) {
#pragma unroll
    for (int t = 0; t < C::RT1; ++t)
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt) g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // which chunk follows `cc` for this wavefront: the next one, unless `cc` closed a group at or past the range end
  auto has_next = [&](uint32_t cc, const uint4& d) { return cc + 1 < nchunks && !((d.z & kLastBit) && cc + 1 >= c1); };
  const uint4 none = make_uint4(0u, 0u, 0u, 0u);

  // MFMA operands of the dP product, two sets (block b4 in set b4 & 1).  !FUSE (short iterations, ONE wavefront per SIMD at rank
  // 32: nothing hides an LDS round trip): the reads of a chunk's first block are issued right after its rows were staged, an
  // iteration ahead, and the blocks / column tiles past the chunk's length are not read at all (three ids per chunk on the
  // papers100M frontier: 70 of the ~100 operand reads of an iteration fed nothing).
  float av2[2][Q2][C::MT2], bv2[2][Q2][C::RT2];
  auto load_block = [&](int b4, int slot) {
#pragma unroll
    for (int kk = 0; kk < Q2; ++kk) {
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt) av2[slot][kk][mt] = dbuf[offA[mt] + b4 * 4 * C::LDOB + kk];
#pragma unroll
      for (int t = 0; t < C::RT2; ++t) {
        bv2[slot][kk][t] = bbuf[offB[t] + b4 * 4 * C::LDBB + kk];
        if (16 * t + lo >= R2) bv2[slot][kk][t] = 0.f;
      }
    }
  };
#if defined(TTEMB_NO_LDS_AHEAD) || defined(TTEMB_NO_PIPE_LDS) || defined(TTEMB_ABL)   // (A/B: the round-4 order -- every read of an iteration issued inside it, whatever the chunk's length)
  constexpr bool kAhead = false;
#else
  constexpr bool kAhead = !FUSE;
#endif
  // ---- prologue: chunk 0 into LDS, rows of chunk 1 and the pairs of chunk 2 in flight ----
  uint4 d_cur = load_desc(ctab, c, nchunks);
  if (!have) d_cur = none;
  bool more1 = have && has_next(c, d_cur);
  uint4 d_nxt = load_desc(ctab, c + 1, nchunks);
  if (!more1) d_nxt = none;
  bool more2 = more1 && has_next(c + 1, d_nxt);
  uint4 d_nn = load_desc(ctab, c + 2, nchunks);
  if (!more2) d_nn = none;
  uint32_t i2_a, val_a, i2_b, val_b;
#if defined(TTEMB_ABL) && (TTEMB_ABL & 128)   // (ablation 128: the first two chunks' rows are requested without waiting for their (i2, row) pairs -- wrong rows, timing only)
  i2_a = i2_b = (uint32_t)b_l;
  val_a = val_b = (uint32_t)(gw * 16u + (uint32_t)b_l);
#else
  fetch_meta(d_cur, i2_a, val_a);
  fetch_meta(d_nxt, i2_b, val_b);
#endif
  request(offsets(d_cur, i2_a, val_a), d_cur);
  stage(true);
  if constexpr (kAhead) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    load_block(0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  request(offsets(d_nxt, i2_b, val_b), d_nxt);
  uint32_t i2_nn, val_nn;
  fetch_meta(d_nn, i2_nn, val_nn);
  uint32_t i2_cur = i2_a, i2_nxt = i2_b;   // FUSE: the i2 of the rows being multiplied travel with them
  uint4 d_n3 = FUSE ? load_desc(ctab, c + 3, nchunks) : none;   // FUSE: descriptors run one chunk ahead of the pairs they locate

#ifdef TTEMB_STAMPS
  long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_n = 0, st_t[11];
  for (int i = 0; i < 11; ++i) st_t[i] = 0;
#define TTEMB_STAMP(i) st_t[i] = clock64(); __builtin_amdgcn_sched_barrier(0)
  const long long st_begin = clock64();
#else
#define TTEMB_STAMP(i)
#endif
  for (;;) {
    // FUSE: the pairs of chunk c + 3 are requested FIRST in the iteration, with a descriptor loaded an iteration ago, and
    // take their place in the rotation at its end.  Requested at the end, next to the rotation (the other form, below):
    // the loop-carried registers of the old pairs are still live there, the loads go into temporaries, and the copy back --
    // a wait for a load issued a few instructions earlier, vmcnt(1), behind a scalar load that is waited for on the spot
    // -- closes every iteration: 106 -> 100 us on the products shape.  (The unfused form with its short iterations --
    // papers100M: 1.5 ids per chunk -- measured 10 % slower this way and keeps the old order.)
    uint32_t i2_n3 = 0u, val_n3 = 0u;
    uint4 d_n4 = none;
    if constexpr (FUSE) {
      fetch_meta(d_n3, i2_n3, val_n3);
      d_n4 = load_desc(ctab, c + 4, nchunks);
    }
    const uint32_t len = d_cur.z & 0xffu;
    // GF: does the batch of this chunk's group end here?  (the group ends, and the wavefront's next chunk -- if it has one --
    // belongs to another batch.)  Its G0 rows are requested now and used after this chunk's products.
    Batch b_cur = {0u, 0u, 0u};
    bool batch_ends = false;
    if constexpr (GF) {
      b_cur = batch_of(uniform(d_cur.y));
      batch_ends = (d_cur.z & kLastBit) != 0u && (!more1 || batch_of(uniform(d_nxt.y)).first_group != b_cur.first_group);
      // A operand of the dG1 product: row c1 = 16 t + lo of G0^T, column rho = 4 s + hi = (group rho / q0, core row rho % q0)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int rho = 4 * s4 + hi;
        const uint32_t i0a = b_cur.i0b + (uint32_t)(rho / Q0);
        const bool on = batch_ends && rho / Q0 < GM && i0a < gf.p0;
#pragma unroll
        for (int t = 0; t < C::RT1; ++t)
          gf_g0[s4][t] = __uint_as_float(buf_load1u(r_g0, (on && 16 * t + lo < R1) ? i0a * (uint32_t)(C::ROW0 * 4) + 4u * ((rho % Q0) * R1 + 16 * t + lo) : kOob));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
    TTEMB_PRIO(0);
    TTEMB_STAMP(0);
    if (d_cur.z & kFirstBit) {
      asm volatile("; a group begins" ::: "memory");   // keeps this a branch (as selects it costs a vector op per register and chunk)
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) dp[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // operands of the E product: A = P^T (read once per chunk), B = the staged d_output rows, one column tile at a time
    constexpr int KS = (C::M2 + 3) / 4;
    float ap[KS][C::RT2];
    float be[2][KS];
    auto load_ap = [&]() {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        // q0 q1 not a multiple of the MFMA K: the last step's rows past M2 are read (inside the buffers) and zeroed
        const bool k_ok = 4 * s + 3 < C::M2 || 4 * s + hi < C::M2;
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) {
          ap[s][t] = pbuf[(4 * s + hi) * C::LDPB + (16 * t + lo) % R2];
          if (16 * t + lo >= R2 || !k_ok) ap[s][t] = 0.f;
        }
      }
    };
    auto load_tile = [&](int nt, int slot) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const bool k_ok = 4 * s + 3 < C::M2 || 4 * s + hi < C::M2;
        be[slot][s] = dbuf[offE[nt] + 4 * s * Q2];
        if (!k_ok) be[slot][s] = 0.f;
      }
    };
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 4))
    // ---- dP += dO (q0q1 x 16 q2) . G2s^T (16 q2 x r2); K-steps whose four ids lie past the chunk's length are skipped ----
#ifndef TTEMB_NO_PIPE_LDS
    // The LDS reads of block b4 + 1 are issued BEFORE the MFMAs of block b4 (two operand sets): a block no longer opens
    // with a wait for its own reads (~100 cycles, three or four times per block).  The reads of a block past the chunk's
    // length are issued all the same (they hit rows of an older chunk and are not used).
    {
      if constexpr (kAhead) {   // the E product's first operands ride in front of the dP MFMAs (they need nothing of them)
        load_ap();
        load_tile(0, 0);
      } else {
        load_block(0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b4 = 0; b4 < kChunk / 4; ++b4) {
        if (b4 + 1 < kChunk / 4 && (!kAhead || (uint32_t)(4 * (b4 + 1)) < len)) load_block(b4 + 1, (b4 + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if ((uint32_t)(4 * b4) < len) {
#pragma unroll
          for (int kk = 0; kk < Q2; ++kk)
#pragma unroll
            for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
              for (int t = 0; t < C::RT2; ++t) {
                if (kNarrowTail && mt == C::MT2 - 1)
                  dp[mt][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(av2[b4 & 1][kk][mt], bv2[b4 & 1][kk][t], dp[mt][t], 0, 0, 0);
                else
                  dp[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2[b4 & 1][kk][mt], bv2[b4 & 1][kk][t], dp[mt][t], 0, 0, 0);
              }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#else
#pragma unroll
    for (int b4 = 0; b4 < kChunk / 4; ++b4) {
      if ((uint32_t)(4 * b4) < len) {
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
            for (int t = 0; t < C::RT2; ++t) {
              if (kNarrowTail && mt == C::MT2 - 1)
                dp[mt][t] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[mt], bv[t], dp[mt][t], 0, 0, 0);
              else
                dp[mt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[t], dp[mt][t], 0, 0, 0);
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keeps operand loads from piling up in registers
    }
#endif
#endif
    TTEMB_STAMP(1);
    // ---- E = P^T (r2 x q0q1) . dO (q0q1 x 16 q2): the A operand (P^T) is read once, column tiles without an id are skipped ----
    f32x4 e[C::RT2][C::NT2];
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 8))
    if constexpr (!kAhead) load_ap();
#ifndef TTEMB_NO_PIPE_LDS   // (the same two-set scheme for the B operand of the E product: -2.5 % on the fused kernel, A/B in one call)
    {
      if constexpr (!kAhead) load_tile(0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        if (nt + 1 < C::NT2 && (!kAhead || (uint32_t)(16 * (nt + 1)) < len * Q2)) load_tile(nt + 1, (nt + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if ((uint32_t)(16 * nt) < len * Q2) {
#pragma unroll
          for (int t = 0; t < C::RT2; ++t) e[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < C::RT2; ++t)
              e[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s][t], be[nt & 1][s], e[t][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#else
#pragma unroll
    for (int nt = 0; nt < C::NT2; ++nt) {
      if ((uint32_t)(16 * nt) < len * Q2) {
#pragma unroll
        for (int t = 0; t < C::RT2; ++t) e[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const bool k_ok = 4 * s + 3 < C::M2 || 4 * s + hi < C::M2;
          float bv = dbuf[offE[nt] + 4 * s * Q2];
          if (!k_ok) bv = 0.f;
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
            e[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s][t], bv, e[t][nt], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every LDS read of this chunk is done: the next chunk's rows may land
    __builtin_amdgcn_sched_barrier(0);
    TTEMB_PRIO(2);
    TTEMB_STAMP(2);

    // ---- GF: a finished group's dP joins its batch; a finished batch is multiplied HERE, in front of the wait for the next
    //      chunk's rows (requested an iteration ago: they arrive under these MFMAs) ----
    f32x4 gf_out[GF ? C::RT1 : 1];
    bool gf_store = false;
    uint32_t gf_store_mask = 0u, gf_store_group = 0u;
    if constexpr (GF) {
      gf_store = false;
      if (d_cur.z & kLastBit) {   // a group is complete: its dP joins the batch's slot; a complete batch is multiplied
        asm volatile("; a group ends" ::: "memory");
        if (b_cur.first_group != gf_batch) {   // the batch's first group (of this wavefront): the rows of the others read zero
          constexpr int Z4 = 16 * LDD / 4;
#pragma unroll
          for (int it = 0; it < (Z4 + kWave - 1) / kWave; ++it)
            if (Z4 % kWave == 0 || it * kWave + lane < Z4) *reinterpret_cast<float4*>(gslot + 4 * (it * kWave + lane)) = make_float4(0.f, 0.f, 0.f, 0.f);
          gf_batch = b_cur.first_group;
          gf_mask = 0u;
        }
        const uint32_t sl = uniform(d_cur.y) - b_cur.first_group;   // the group's place in its batch (< GM)
        gf_mask |= 1u << sl;
        float* const rows = gslot + sl * (uint32_t)(Q0 * LDD);
        // dP[m = (a, j)][c2] -> slot row (sl q0 + a), column n = j r2 + c2
#pragma unroll
        for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (kNarrowTail && mt == C::MT2 - 1) {   // narrow tile: register r of lane (hi, lo) is lane group hi's share of dP[16 mt + r][16 t + lo]
                float v = dp[mt][t][r];
                v += __shfl_xor(v, 16, kWave);
                v += __shfl_xor(v, 32, kWave);
                const int m = 16 * mt + r;
                if (hi == 0 && 16 * t + lo < R2 && r < kTailRows) rows[(m / Q1) * LDD + (m % Q1) * R2 + 16 * t + lo] = v;
              } else {
                const int m = 16 * mt + 4 * hi + r;
                if (m < C::M2 && 16 * t + lo < R2) rows[(m / Q1) * LDD + (m % Q1) * R2 + 16 * t + lo] = dp[mt][t][r];
              }
            }
        if (batch_ends) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (b_cur.i1 != gf_i1) {   // rare (a share covers a fraction of one i1): the sums of the i1 before leave, G1[i1] is loaded
            asm volatile("; another i1" ::: "memory");
            if (gf_i1 != 0xffffffffu) flush_g1();
            const float* g1 = gf.G1 + (size_t)b_cur.i1 * C::ROW1;
#pragma unroll
            for (int t = 0; t < C::RT1; ++t)
#pragma unroll
              for (int v = 0; v < KPH3 / 4; ++v) {   // B[k = n = KPH3 hi + s][c1 = 16 t + lo] = G1[i1][c1][n]
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (16 * t + lo < R1) x = *reinterpret_cast<const float4*>(g1 + (16 * t + lo) * C::N1 + KPH3 * hi + 4 * v);
                gf_g1[4 * v][t] = x.x; gf_g1[4 * v + 1][t] = x.y; gf_g1[4 * v + 2][t] = x.z; gf_g1[4 * v + 3][t] = x.w;
              }
            gf_i1 = b_cur.i1;
          }
          // operands of both products out of the slot, then the MFMAs back to back
          float b1[4][C::NT1];   // dG1: B[k = rho = 4 s + hi][n = 16 nt + lo]
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int nt = 0; nt < C::NT1; ++nt) b1[s4][nt] = gslot[(4 * s4 + hi) * LDD + 16 * nt + lo];
          float a3[KS3];         // dG0: A[rho = lo][k = n = KPH3 hi + s]
#pragma unroll
          for (int v = 0; v < KPH3 / 4; ++v) {
            const float4 x = *reinterpret_cast<const float4*>(gslot + lo * LDD + KPH3 * hi + 4 * v);
            a3[4 * v] = x.x; a3[4 * v + 1] = x.y; a3[4 * v + 2] = x.z; a3[4 * v + 3] = x.w;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();   // every read of the slot is done before the next batch's rows land
          TTEMB_PRIO(0);
          // dG1[i1] += [G0 rows]^T (r1 x 16) . [dP] (16 x q1 r2)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int nt = 0; nt < C::NT1; ++nt)
#pragma unroll
              for (int t = 0; t < C::RT1; ++t)
                g1acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gf_g0[s4][t], b1[s4][nt], g1acc[t][nt], 0, 0, 0);
          // dG0 parts = [dP] (16 x q1 r2) . G1[i1]^T (q1 r2 x r1); four interleaved accumulation chains
          f32x4 g0p[4][C::RT1];
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
            for (int t = 0; t < C::RT1; ++t) g0p[c4][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s3 = 0; s3 < KS3; ++s3)
#pragma unroll
            for (int t = 0; t < C::RT1; ++t)
              g0p[s3 & 3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[s3], gf_g1[s3][t], g0p[s3 & 3][t], 0, 0, 0);
          TTEMB_PRIO(2);
          // accumulator row 4 hi + r = rho = (group rho / q0 of the batch, core row rho % q0), column c1 = 16 t + lo: the part of
          // every group this wavefront staged (the finalize kernel sums the parts of the non-empty groups over i1).  Stored
          // further down, behind this chunk's E rows: no wait of the pipeline sits behind a store
#pragma unroll
          for (int t = 0; t < C::RT1; ++t) gf_out[t] = (g0p[0][t] + g0p[1][t]) + (g0p[2][t] + g0p[3][t]);
          gf_store = true;
          gf_store_mask = gf_mask;
          gf_store_group = gf.parts_by_i0 ? b_cur.i0b * gf.p1 + b_cur.i1 : b_cur.first_group;
          gf_batch = 0xffffffffu;
          gf_mask = 0u;
        }
      }
    }
#ifdef TTEMB_STAMPS
    if constexpr (GF) {   // (the group products' share of the "stage" interval, and how many iterations multiplied a batch)
      __builtin_amdgcn_sched_barrier(0);
      const long long now = clock64();
      st_acc[9] += now - st_t[2];
      st_acc[8] += gf_store ? 1 : 0;
      st_t[2] = now;
    }
#endif
    // ---- rows of the next chunk: registers -> LDS; then the offsets of the chunk after (consumes its pairs) ----
    Offs o_nn;
    uint4 d_n3x = none;
    if constexpr (!FUSE) {
      stage((d_nxt.z & kFirstBit) != 0u);
      if constexpr (kAhead) {   // the next chunk's first dP operands: its rows have just landed, the stores and requests below hide the trip
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        load_block(0, 0);
      }
#ifndef TTEMB_DESC_LATE
      // the descriptor of chunk c + 3 is requested HERE and used at the end of the iteration (loaded there, its scalar round
      // trip -- a wait of its own, ~600 cycles -- closed every iteration).  From here to the end of the iteration no LDS
      // operation is waited for, so the outstanding scalar load turns no partial LDS wait into a full one.
      d_n3x = load_desc(ctab, c + 3, nchunks);
#endif
      o_nn = offsets(d_nn, i2_nn, val_nn);
      // the offsets are needed only after the stores; computed there, their wait for the (i2, row) pairs would become a
      // wait for every store in front of it as well (vmcnt counts in order): ~3 000 cycles per chunk.  Pin them here.
      asm volatile("" ::"v"(o_nn.row), "v"(o_nn.grow));
    }
    __builtin_amdgcn_sched_barrier(0);
    TTEMB_STAMP(3);

    // ---- this chunk's results leave now: no load is waited for behind these stores ----
    // Lane (hi, lo) holds E[c2 = 16 t + 4 hi + r][col = 16 nt + lo] with col = id * q2 + kk, and the E table keeps
    // an id's row as [kk][c2] (the reduce kernel sums rows element by element, the finalize kernel puts dG2 back
    // into [c2][kk]): the chunk's rows are one contiguous block indexed col * r2 + c2, written in 16-byte pieces
    // through a descriptor that covers exactly this chunk's rows.
    if constexpr (FUSE) {
      // (the wave barrier above: every staged-G2 read of this chunk is done; the second chunk's E rows take that region)
      float* const eb = sub == 0 ? ebuf : bbuf;
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        if ((uint32_t)(16 * nt) < len * Q2) {
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
            if (ewr[nt][t] >= 0) *reinterpret_cast<f32x4*>(eb + ewr[nt][t]) = e[t][nt];
        }
      }
      if (j_l == 0 && (uint32_t)b_l < len)   // the row joins the list of its i2
        f_tab[(wave * kFuseSub + sub) * 16 + b_l] =
            make_uint2(sub == 0 ? my_row0 : my_row1, atomicExch(&f_head[i2_cur], sub == 0 ? my_entry0 : my_entry1));
      if (lane == 0) f_more[wave] = more1 ? 1u : 0u;
    }
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 1))
    if constexpr (!FUSE) {
      const rsrc_t r_e = make_rsrc(plan.etab + (size_t)uniform(d_cur.x) * C::ROW2, uniform(len) * (uint32_t)(C::ROW2 * 4));
#pragma unroll
      for (int nt = 0; nt < C::NT2; ++nt) {
        if ((uint32_t)(16 * nt) < len * Q2) {
#pragma unroll
          for (int t = 0; t < C::RT2; ++t)
            buf_store4_p<(TTEMB_NT & 8) != 0>(r_e, eoff[t] + (uint32_t)(16 * nt * R2 * 4), make_float4(e[t][nt][0], e[t][nt][1], e[t][nt][2], e[t][nt][3]));
        }
      }
    }
#endif
    if constexpr (GF) {
      // (the group products ran before the next chunk's rows were staged; their dG0 parts leave now, behind the E rows)
      if (gf_store) {
        const uint32_t gf_store_step = gf.parts_by_i0 ? gf.p1 : 1u;   // rows between the parts of two neighbouring groups of the batch
#pragma unroll
        for (int t = 0; t < C::RT1; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rho = 4 * hi + r;
            const bool on = rho / Q0 < GM && ((gf_store_mask >> (rho / Q0)) & 1u) != 0u && 16 * t + lo < R1;
            buf_store1(r_part, on ? (gf_store_group + (uint32_t)(rho / Q0) * gf_store_step) * (uint32_t)(C::ROW0 * 4) + 4u * ((rho % Q0) * R1 + 16 * t + lo) : kOob, gf_out[t][r]);
          }
      }
    } else
#if defined(TTEMB_ABL) && (TTEMB_ABL & 512)   // (ablation 512: the dP table is not written -- with the epilogue launch skipped: what folding the epilogue into this kernel can save at most; timing only)
    if (!FUSE && nnz == 0xffffffffu)
#else
    if (d_cur.z & kLastBit)
#endif
    {  // dP of a finished group, into its own slot
      const rsrc_t r_dp = make_rsrc(plan.dptab + (size_t)uniform(d_cur.y) * PF, (uint32_t)(PF * 4));
#pragma unroll
      for (int mt = 0; mt < C::MT2; ++mt)
#pragma unroll
        for (int t = 0; t < C::RT2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (kNarrowTail && mt == C::MT2 - 1) {
              // narrow tile: register r of lane (hi, lo) is lane group hi's share of dP[16 mt + r][c2 = 16 t + lo]
              float v = dp[mt][t][r];
              v += __shfl_xor(v, 16, kWave);
              v += __shfl_xor(v, 32, kWave);
              const bool mine = hi == 0 && 16 * t + lo < R2 && r < kTailRows;
              buf_store1(r_dp, mine ? (uint32_t)((16 * mt + r) * R2 + 16 * t + lo) * 4u : kOobBase, v);
            } else {
              buf_store1(r_dp, dpoff[mt][t] + (uint32_t)(r * R2 * 4), dp[mt][t][r]);
            }
          }
    }
    TTEMB_STAMP(4);
    bool reduce_now = false;
    if constexpr (FUSE) {
      reduce_now = sub == kFuseSub - 1;
      sub = reduce_now ? 0u : sub + 1u;
      if (!reduce_now) {   // first chunk of the round: its E rows wait in their own region, the staging regions are free
        stage((d_nxt.z & kFirstBit) != 0u);
        o_nn = offsets(d_nn, i2_nn, val_nn);
        asm volatile("" ::"v"(o_nn.row), "v"(o_nn.grow));
      }
    }
    if (FUSE && reduce_now) {
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 32))   // (ablation 32: no barriers -- wrong sums, timing only)
      __syncthreads();   // the round's E rows and their lists are in LDS
#endif
      TTEMB_STAMP(5);
#if !(defined(TTEMB_ABL) && (TTEMB_ABL & 16))   // (ablation 16: no list walk)
      {
        // slab row of this lane in quad q: i2 = 8 (q GPW + lane group) + wave.  First the heads of all quads, then the
        // first row of every list (reads in flight together), then whatever is left of longer lists
        // head of this lane's slab row in quad q: f_head[(q GPW + lane group) kFuseWaves + wave]; idle lanes use the spare word
        uint32_t* const my_head = s_grp < GPW ? f_head + ((uint32_t)s_grp * kFuseWaves + wave) : f_head + kFuseWaves * GPW * FQ;
        const int qstep = s_grp < GPW ? GPW * kFuseWaves : 0;
        const char* const rows = lds0 + (uint32_t)rd_off * 4u;
#pragma unroll
        for (int q0 = 0; q0 < FQ; q0 += FB) {
          if ((uint32_t)(q0 * GPW * kFuseWaves) >= p2) break;   // wave-uniform: no slab row in the remaining quads
          uint32_t en[FB];   // entry being visited
#pragma unroll
          for (int u = 0; u < FB; ++u) en[u] = my_head[(q0 + u) * qstep];   // heads past p2 hold NIL for ever
#pragma unroll
          for (int u = 0; u < FB; ++u) my_head[(q0 + u) * qstep] = f_nil;   // (every lane of the group stores the same word)
          // The lists of the batch are walked TOGETHER, one entry of each per step: entries, then rows, in flight together; a
          // list that has ended (or was empty) keeps visiting NIL and adds zeros.  The number of steps is the longest list of
          // the batch (~3: 128 rows over ~140 values of i2) -- walking the tails list by list cost one dependent LDS round
          // trip pair per extra entry of every list: 3 600 of the 9 700 cycles of an iteration.
          for (;;) {
            uint2 ent[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) ent[u] = *reinterpret_cast<const uint2*>(lds0 + en[u]);
            f32x4 x[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) x[u] = *reinterpret_cast<const f32x4*>(rows + ent[u].x);
            bool left = false;
#pragma unroll
            for (int u = 0; u < FB; ++u) {
              slab[q0 + u] += x[u];
              en[u] = ent[u].y;
              left = left || en[u] != f_nil;
            }
            if (__ballot(left) == 0ull) break;
          }
        }
      }
#endif
#if defined(TTEMB_ABL) && (TTEMB_ABL & 32)
      const bool any = more1;   // every wavefront for itself (the loop ends with the wavefront's own share)
      TTEMB_STAMP(6);
#else
      const bool any = __ballot(lane < kFuseWaves && f_more[lane & (kFuseWaves - 1)] != 0u) != 0ull;
      TTEMB_STAMP(6);
      __syncthreads();   // every E row has been added: the regions take the next chunks' rows
#endif
      TTEMB_STAMP(7);
      stage((d_nxt.z & kFirstBit) != 0u);
      o_nn = offsets(d_nn, i2_nn, val_nn);
      asm volatile("" ::"v"(o_nn.row), "v"(o_nn.grow));
      TTEMB_STAMP(8);
      if (!any) break;
    } else if (!FUSE) {
      if (!more1) break;
    }
    // ---- loads of the chunk after next, pairs of the one after that ----
    request(o_nn, d_nn);
    ++c;
    i2_cur = i2_nxt;
    i2_nxt = i2_nn;
    d_cur = d_nxt;
    d_nxt = d_nn;
    more1 = more2;
    more2 = more1 && has_next(c + 1, d_nxt);
    if constexpr (FUSE) {
      d_nn = d_n3;
      if (!more2) d_nn = none;
      d_n3 = d_n4;
      i2_nn = i2_n3;
      val_nn = val_n3;
    } else {
#ifndef TTEMB_DESC_LATE
      d_nn = d_n3x;   // (c has advanced: this is chunk c + 2)
#else
      d_nn = load_desc(ctab, c + 2, nchunks);
#endif
      if (!more2) d_nn = none;
      fetch_meta(d_nn, i2_nn, val_nn);
    }
#ifdef TTEMB_STAMPS
    __builtin_amdgcn_sched_barrier(0);
    st_t[FUSE ? 9 : 5] = clock64();
    for (int i = 0; i < (FUSE ? 9 : 5); ++i) st_acc[i] += st_t[i + 1] - st_t[i];
    ++st_n;
#endif
  }
  if constexpr (GF) {
    if (gf_i1 != 0xffffffffu) flush_g1();   // what this wavefront still holds of dG1
  }
  if constexpr (FUSE) {
    // this wavefront's rows of the workgroup's slab: i2 = 8 (q GPW + lane group) + wave, every row of the slab is written by
    // one wavefront (zeros where no id came by); the lane's quad goes back to its place in the [kk][c2] row
    float* tile = plan.g2part + (size_t)blockIdx.x * p2 * C::ROW2;
#if defined(TTEMB_ABL) && (TTEMB_ABL & 64)   // (ablation 64: the slab rows are not stored -- timing only; the sums stay live)
    if (nnz == 0xffffffffu)
#endif
#pragma unroll
    for (int q = 0; q < FQ; ++q) {
      const uint32_t i2 = (uint32_t)(q * GPW + s_grp) * kFuseWaves + wave;
      if (s_grp < GPW && i2 < p2) *reinterpret_cast<f32x4*>(tile + (size_t)i2 * C::ROW2 + 4 * s_piece) = slab[q];
    }
  }
#ifdef TTEMB_STAMPS
  if (lane == 0 && wave == 0 && (blockIdx.x % 97) == 0 && st_n > 0)
    printf("chunk kernel wg %u: %lld iterations, total %lld cycles; per iteration: dP %lld | E %lld | stage(+wait) %lld | stores %lld | request/B1 %lld | reduce %lld | B2 %lld | stage %lld | request %lld%s | GF: group products %lld per iteration, %lld batches\n",
           blockIdx.x, st_n, (long long)(clock64() - st_begin), st_acc[0] / st_n, st_acc[1] / st_n, st_acc[2] / st_n, st_acc[3] / st_n,
           st_acc[4] / st_n, st_acc[5] / st_n, st_acc[6] / st_n, st_acc[7] / st_n, GF ? 0ll : st_acc[8] / st_n, "", GF ? st_acc[9] / st_n : 0ll, GF ? st_acc[8] : 0ll);
#endif
}

#ifndef TTEMB_REDUCE_NB
#define TTEMB_REDUCE_NB 1   // (2 / 3 / 4 buckets in flight per wavefront measured: 103-105 us at papers100M whichever -- see below)
#endif
#ifndef TTEMB_REDUCE_U
#define TTEMB_REDUCE_U 6
#endif
constexpr int NWB = 16;                       // wavefronts per workgroup of the dG2 reduce
constexpr int kReduceNB = TTEMB_REDUCE_NB;    // buckets a wavefront sums together
constexpr int kReduceU = TTEMB_REDUCE_U;      // row loads in flight per bucket and lane group
// B. dG2 reduce.  A workgroup takes kRowsB consecutive E rows, buckets them by i2 inside LDS
// (a tile-local counting sort of row numbers), then each wave sums the rows of "its" i2 values
// in registers and stores one (r2 q2)-float row per i2 into the tile's slab of partial sums
// (plain stores: atomics from every tile onto the 45 KB of dG2 ran at ~0.1 TB/s).  E rows are
// read exactly once, 16 bytes per lane.  fast3_finalize_kernel adds the slabs up.
// SHARED: tables whose slab (p2 rows of r2 q2 floats) is large keep ONE zero-filled slab that every tile adds into
// with float atomics, a whole row (contiguous floats) per non-empty bucket, instead of a slab per tile (a 4-core
// table with a merged last pair has 0.9 MB slabs: 256 of them were 236 MB to write and read back).
template <int ROW2, int kRowsMax, int NWB, bool SHARED>
__global__ __launch_bounds__(NWB * 64) void fast3_dg2_reduce_kernel(GroupPlan plan, int G, uint32_t p2, uint32_t kRowsB, uint32_t stride) {
  extern __shared__ uint32_t lds_u[];   // [p2 + 1] bucket starts | [p2] cursors | [kRowsMax] row list (uint16)
  uint32_t* bstart = lds_u;
  uint32_t* cursor = lds_u + p2 + 1;
  unsigned short* rows = reinterpret_cast<unsigned short*>(cursor + p2);
  constexpr int F4 = ROW2 / 4;      // float4 per row
  constexpr int SUB = kWave / F4;   // rows handled per load instruction
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (plan_poisoned(plan, (uint32_t)G)) return;
  const uint32_t total = (uint32_t)plan.gpre[G];
  const uint32_t s0 = blockIdx.x * kRowsB;
  const uint32_t n_rows = s0 >= total ? 0u : (s0 + kRowsB < total ? kRowsB : total - s0);
  // rows wider than a wavefront load (the wide-rank chain) are reduced ROW2 floats at a time: launch row blockIdx.y
  // takes columns [ROW2 y, ROW2 (y + 1)) of every E row (`stride` floats apart) and of the slab
  const uint32_t col0 = blockIdx.y * ROW2;
  float* slab = plan.g2part + (SHARED ? (size_t)0 : (size_t)blockIdx.x * p2 * stride) + col0;  // this tile's partial dG2, every row written
  if constexpr (SHARED) {
    // Far more buckets than rows in a tile (a 4-core table with a merged last pair: 3 600 values of i2, <= 2 048 rows): hardly
    // any two rows of a tile share their i2, so bucketing them costs a scan and a walk over every (mostly empty) bucket
    // per tile and saves no atomic.  Every row is added to its slab row as it is: SUB rows per 16-byte load, then one
    // atomic instruction of 64 consecutive floats per row.
    if (p2 > (uint32_t)kRowsMax && ROW2 == 4 * (ROW2 / 4) && kWave % F4 == 0) {
      const int sub = lane / F4, c4 = lane - sub * F4;
      for (uint32_t j0 = wave * SUB; j0 < n_rows; j0 += NWB * SUB) {
        const uint32_t r = j0 + sub;
        const bool on = r < n_rows;
        const uint32_t i2 = on ? plan.i2s[s0 + r] : 0u;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) acc = *reinterpret_cast<const float4*>(plan.etab + (size_t)(s0 + r) * stride + col0 + 4 * c4);
#pragma unroll
        for (int k = 0; k < SUB; ++k) {   // row j0 + k: its floats sit in lanes k F4 .. k F4 + F4 - 1, four per lane
          const uint32_t i2k = __shfl(i2, k * F4, kWave);
          const bool onk = j0 + k < n_rows;
          for (int e0 = 0; e0 < ROW2; e0 += kWave) {
            const int e = e0 + lane, src = k * F4 + ((e >> 2) % F4);
            const float x = __shfl(acc.x, src, kWave), y = __shfl(acc.y, src, kWave);
            const float z = __shfl(acc.z, src, kWave), w = __shfl(acc.w, src, kWave);
            const float v = (e & 3) == 0 ? x : ((e & 3) == 1 ? y : ((e & 3) == 2 ? z : w));
            if (onk && e < ROW2) atomicAdd(slab + (size_t)i2k * stride + e, v);
          }
        }
      }
      return;
    }
  }
  for (uint32_t e = tid; e <= p2; e += NWB * 64) bstart[e] = 0;
  __syncthreads();
  // histogram of i2 over the tile (integer LDS atomics; 8 ids per thread)
  uint32_t my_i2[kRowsMax / (NWB * 64)], my_rank[kRowsMax / (NWB * 64)];
#pragma unroll
  for (int k = 0; k < kRowsMax / (NWB * 64); ++k) {
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
  for (int k = 0; k < kRowsMax / (NWB * 64); ++k)
    if (my_i2[k] != 0xffffffffu) rows[bstart[my_i2[k]] + my_rank[k]] = (unsigned short)(k * NWB * 64 + tid);
  __syncthreads();
  // wave w sums the buckets i2 = w, w + NWB, ... -- kReduceNB of them at a time.  A bucket of a tile is a handful of rows (8 at
  // papers100M: 3 200 rows over 400 values of i2) and the kernel reads its 512-byte rows at 4.6 TB/s there; two, three or four
  // buckets in flight per wavefront change nothing (105.2 / 104.4 / 104.9 against 103.2 us, one call, round 5): the rate is
  // what gathered 512-byte rows get (the guide: 5.5 TB/s for 1 152-byte rows, 6.3 streaming), not a latency chain.
  const int sub = lane / F4, c4 = lane - sub * F4;
  for (uint32_t i2a = wave; i2a < p2; i2a += kReduceNB * NWB) {
    uint32_t b0[kReduceNB], b1[kReduceNB];
    float4 acc[kReduceNB];
    uint32_t longest = 0;
#pragma unroll
    for (int n = 0; n < kReduceNB; ++n) {
      const uint32_t i2 = i2a + n * NWB;
      b0[n] = i2 < p2 ? bstart[i2] : 0u;
      b1[n] = i2 < p2 ? bstart[i2 + 1] : 0u;
      acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
      longest = b1[n] - b0[n] > longest ? b1[n] - b0[n] : longest;
    }
    if (sub < SUB) {
      // several independent row loads in flight per lane group (the loop is latency-bound otherwise)
      constexpr int U = kReduceU;
      for (uint32_t t = sub; t - sub < longest; t += U * SUB) {
        uint32_t rr[kReduceNB][U];
        float4 v[kReduceNB][U];
#pragma unroll
        for (int n = 0; n < kReduceNB; ++n)
#pragma unroll
          for (int u = 0; u < U; ++u) rr[n][u] = b0[n] + t + u * SUB < b1[n] ? rows[b0[n] + t + u * SUB] : 0xffffffffu;
#pragma unroll
        for (int n = 0; n < kReduceNB; ++n)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            v[n][u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr[n][u] != 0xffffffffu)
              v[n][u] = *reinterpret_cast<const float4*>(plan.etab + (size_t)(s0 + rr[n][u]) * stride + col0 + 4 * c4);
          }
#pragma unroll
        for (int n = 0; n < kReduceNB; ++n)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc[n].x += v[n][u].x; acc[n].y += v[n][u].y; acc[n].z += v[n][u].z; acc[n].w += v[n][u].w;
          }
      }
    }
#pragma unroll
    for (int n = 0; n < kReduceNB; ++n) {
      const uint32_t i2 = i2a + n * NWB;
      if (i2 >= p2) break;   // wave-uniform
      if (SHARED && b0[n] == b1[n]) continue;   // nothing to add
#pragma unroll
      for (int k = 1; k < SUB; ++k) {
        const int src = (lane + k * F4) & 63;
        const float x = __shfl(acc[n].x, src, kWave), y = __shfl(acc[n].y, src, kWave);
        const float z = __shfl(acc[n].z, src, kWave), w = __shfl(acc[n].w, src, kWave);
        if (lane < F4) { acc[n].x += x; acc[n].y += y; acc[n].z += z; acc[n].w += w; }
      }
      if constexpr (SHARED) {
        // element e of the row sits in lane e / 4, component e % 4: hand it to lane e so that an atomic instruction adds
        // 64 consecutive floats (the shape the memory system takes at full rate; 16 lanes x 4 strided scalars did not)
        for (int e0 = 0; e0 < ROW2; e0 += kWave) {
          const int e = e0 + lane, src = (e >> 2) & 63;
          const float x = __shfl(acc[n].x, src, kWave), y = __shfl(acc[n].y, src, kWave);
          const float z = __shfl(acc[n].z, src, kWave), w = __shfl(acc[n].w, src, kWave);
          const float v = (e & 3) == 0 ? x : ((e & 3) == 1 ? y : ((e & 3) == 2 ? z : w));
          if (e < ROW2) atomicAdd(slab + (size_t)i2 * stride + e, v);
        }
      } else {
        if (lane < F4) *reinterpret_cast<float4*>(slab + (size_t)i2 * stride + 4 * lane) = acc[n];
      }
    }
  }
}

// C. group epilogue: one wavefront per (i1, slice of `gpw` consecutive i0) -- groups are numbered i1 * p0 + i0.
//    The slice is walked 16/q0 groups at a time: their dP (q0 x q1 r2 each) stacked are one 16-row MFMA operand, so
//        dG0 parts  = [dP of 16/q0 groups] (16 x q1r2) . G1[i1]^T (q1r2 x r1)      -- all 16 rows of the tile are real
//        dG1[i1]   += [G0 rows of those groups]^T (r1 x 16) . [dP] (16 x q1r2)     -- K = 16 instead of q0
//    (group by group the first product used q0 of 16 tile rows: 25 MFMAs per group against 10 now).
//    dG1 accumulates in MFMA registers over the slice and leaves as the slice's own slab (plain stores; the first
//    version flushed with float atomics: 2.8 M of them per launch bounded the kernel), the dG0 contribution is stored
//    per group.  The finalize kernel adds slabs and contributions.  The loop body is the same instruction stream for
//    every batch: an empty group's dP load falls off its buffer (zeros), so it multiplies zeros and stores a zero
//    contribution -- no branch around a memory instruction; the next batch's operands are requested before this
//    batch's MFMAs and its stores are issued after them.
#ifndef TTEMB_EPI_SLICES
#define TTEMB_EPI_SLICES 16
#endif
constexpr int kEpiSlices = TTEMB_EPI_SLICES;   // target number of i0 slices (dG1 slabs)
// fewer than three quarters of the groups hold an id (count left in counts[G] by the grouping pass): wave-uniform
__device__ __forceinline__ bool sparse_groups(const GroupPlan& plan, uint32_t G) {
  return __builtin_amdgcn_readfirstlane(plan.counts[G]) * 4u < 3u * G;
}

// SKIP: walk only the batches that hold an id and leave the dG0 parts of the others unwritten (the finalize kernel
// then skips them by their counts) -- the form for frontiers that touch a small share of the groups.  Without SKIP
// the loop is the plain counted one, and every group gets a part (zeros for an empty one).
// Wavefronts per workgroup of the epilogue: they take consecutive slices of one i1 and share its G1 row in LDS.  One is
// enough while that row is small; at rank 32 (16-20 KB next to a wavefront's 8-10 KB of dP) one-wave workgroups filled
// the CU's LDS with six wavefronts, and the kernel is a chain of round trips per batch: four share the row then.
template <int Q0, int Q1, int Q2, int R1, int R2>
struct EpiCfg {
  static constexpr int G1_BYTES = R1 * (Q1 * R2 + 1) * 4;
#ifndef TTEMB_EPI_SHARE_BYTES
#define TTEMB_EPI_SHARE_BYTES (8 * 1024)
#endif
  static constexpr int WAVES = G1_BYTES > TTEMB_EPI_SHARE_BYTES ? 4 : 1;
};

template <int Q0, int Q1, int Q2, int R1, int R2, bool SKIP>
__device__ __forceinline__ void epilogue_unit(
    const float* __restrict__ G0, const float* __restrict__ G1, uint32_t p0, uint32_t p1, uint32_t gpw, const GroupPlan& plan,
    float* dpbuf, float* g1buf, uint32_t slice, uint32_t slices, uint32_t i1) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  constexpr int GM = 16 / Q0;            // groups per batch: their q0 rows fill one 16-row MFMA tile
  constexpr int LDD = C::N1 + 4;         // row stride of the staged dP rows (16-byte aligned rows)
  constexpr int PF = C::M2 * R2;         // floats of one group's dP = q0 * N1
  constexpr int BF4 = 16 * C::N1 / 4;    // float4 pieces of a batch's stacked dP
  constexpr int NLB = (BF4 + kWave - 1) / kWave;
  constexpr int KS3 = C::N1 / 4;
  static_assert(C::N1 % 4 == 0, "dP rows move as float4");
  constexpr int EW = EpiCfg<Q0, Q1, Q2, R1, R2>::WAVES;
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 4, lo = lane & 15;
  const bool active = slice < slices;   // (a workgroup's last wavefronts may have no slice: they only help with G1)
  const uint32_t i0_begin = active ? slice * gpw : p0;
  const uint32_t i0_end = i0_begin + gpw < p0 ? i0_begin + gpw : p0;
  const uint32_t g_begin = i1 * p0 + (active ? i0_begin : 0u);
  const uint32_t G = p0 * p1;
  const rsrc_t r_dp = make_rsrc(plan.dptab, G * (uint32_t)PF * 4u);
  const rsrc_t r_g0 = make_rsrc(G0, p0 * (uint32_t)C::ROW0 * 4u);
  const rsrc_t r_part = make_rsrc(plan.g0part, G * (uint32_t)C::ROW0 * 4u);

  f32x4 g1acc[C::RT1][C::NT1];
#pragma unroll
  for (int t = 0; t < C::RT1; ++t)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt) g1acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // which of this wave's groups hold ids (one lane per group; gpw <= 64)
  const uint32_t cnt_l = (i0_begin + lane < i0_end) ? plan.counts[g_begin + lane] : 0u;   // (an inactive wavefront: none)
  const unsigned long long live = __ballot(cnt_l != 0);
  float4 nxt[NLB];
  float nxt_g0[4][C::RT1];
  auto request = [&](uint32_t k0) {   // operands of the batch of groups k0 .. k0 + GM - 1; an empty or missing group reads zeros
#pragma unroll
    for (int it = 0; it < NLB; ++it) {
      const int e = it * kWave + lane;                  // float4 number e of the stacked [16][N1] block
      const uint32_t gi = k0 + (uint32_t)(4 * e / PF);  // the group it belongs to
      const bool on = (BF4 % kWave == 0 || e < BF4) && 4 * e / PF < GM && gi < gpw && ((live >> gi) & 1ull);   // tile rows past the
      nxt[it] = buf_load4(r_dp, on ? (g_begin + k0) * (uint32_t)(PF * 4) + 16u * e : kOob);                    // GM whole groups stay zero
    }
    // A operand of the dG1 product: row c = 16 t + lo of G0^T, column rho = 4 s + hi = (group rho / q0, core row rho % q0)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int rho = 4 * s + hi;
      const uint32_t gi = k0 + (uint32_t)(rho / Q0);
      const bool on = rho / Q0 < GM && gi < gpw && ((live >> gi) & 1ull);
#pragma unroll
      for (int t = 0; t < C::RT1; ++t)
        nxt_g0[s][t] = __uint_as_float(buf_load1u(
            r_g0, (on && 16 * t + lo < R1) ? (i0_begin + gi) * (uint32_t)(C::ROW0 * 4) + 4u * ((rho % Q0) * R1 + 16 * t + lo) : kOob));
    }
  };
  // only batches that hold an id are walked (a METIS-ordered frontier touches a few per cent of the groups; their
  // dG0 parts are neither computed nor stored, and the finalize kernel skips them by their counts)
  auto next_live = [&](uint32_t k) {   // first batch start >= k with a non-empty group, or past the slice
    if constexpr (SKIP) {
      while (k < gpw && !((live >> k) & ((1ull << GM) - 1ull))) k += GM;
    }
    return k;
  };
  if constexpr (SKIP) {   // a slice without ids leaves no slab and no parts: one flag says so
    if (active && lane == 0) plan.epi_live[(size_t)slice * p1 + i1] = live != 0ull ? 1u : 0u;
    if (EW == 1 && live == 0ull) return;
  }
  const uint32_t k_first = next_live(0);
  request(k_first);
  {
    // G1[i1] -> LDS (also for a slice without ids: 0 x stale LDS could be NaN): all loads first -- a load / wait /
    // store per piece would serialise ~20 L2 round trips.  The workgroup's wavefronts take every EW-th piece.
    const float* g1 = G1 + (size_t)i1 * C::ROW1;
    constexpr int NG1 = (C::ROW1 + EW * kWave - 1) / (EW * kWave);
    const int t0 = (int)threadIdx.x;   // 0 .. EW * 64 - 1
    float g1v[NG1];
#pragma unroll
    for (int it = 0; it < NG1; ++it) {
      const int e = it * EW * kWave + t0;
      g1v[it] = (C::ROW1 % (EW * kWave) == 0 || e < C::ROW1) ? g1[e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NG1; ++it) {
      const int e = it * EW * kWave + t0;
      if (C::ROW1 % (EW * kWave) == 0 || e < C::ROW1) g1buf[(e / C::N1) * C::LDG + e % C::N1] = g1v[it];
    }
    if constexpr (EW > 1) {
      __syncthreads();   // every wavefront of the workgroup is here: none has left yet
      if (!active || (SKIP && live == 0ull)) return;
    }
  }
  const uint32_t n_here = i0_end - i0_begin;
  for (uint32_t k0 = k_first; k0 < n_here;) {
    // the batch's stacked dP -> LDS [rho][n]
    float g0v[4][C::RT1];
#pragma unroll
    for (int it = 0; it < NLB; ++it) {
      const int e = it * kWave + lane;
      if (BF4 % kWave == 0 || e < BF4) *reinterpret_cast<float4*>(dpbuf + (4 * e / C::N1) * LDD + (4 * e) % C::N1) = nxt[it];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) g0v[s][t] = nxt_g0[s][t];
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t k_next = next_live(k0 + GM);
    request(k_next);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // one wavefront per workgroup: LDS hand-over only (a __syncthreads would also drain the prefetch)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // operands of both products are read from LDS first, then the MFMAs run back to back
    float b1[4][C::NT1];     // dG1: B[k = rho = 4 s + hi][n = 16 nt + lo]
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt)
        b1[s][nt] = (C::N1 % 16 == 0 || 16 * nt + lo < C::N1) ? dpbuf[(4 * s + hi) * LDD + 16 * nt + lo] : 0.f;
    float a3[KS3], b3[KS3][C::RT1];   // dG0: A[rho = lo][k = n = 4 s + hi], B[k = n][c = 16 t + lo] = G1[c][n]
#pragma unroll
    for (int s = 0; s < KS3; ++s) {
      const int n = 4 * s + hi;
      a3[s] = dpbuf[lo * LDD + n];
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) b3[s][t] = 16 * t + lo < R1 ? g1buf[(16 * t + lo) * C::LDG + n] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // every LDS read of this batch is done before the next dP lands
    // dG1[i1] += [G0 rows]^T (r1 x 16) . [dP] (16 x q1 r2)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int nt = 0; nt < C::NT1; ++nt)
#pragma unroll
        for (int t = 0; t < C::RT1; ++t)
          g1acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(g0v[s][t], b1[s][nt], g1acc[t][nt], 0, 0, 0);
    // dG0 parts = [dP] (16 x q1 r2) . G1[i1]^T (q1 r2 x r1); four interleaved accumulation chains
    f32x4 g0part[4][C::RT1];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t) g0part[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS3; ++s)
#pragma unroll
      for (int t = 0; t < C::RT1; ++t)
        g0part[s & 3][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[s], b3[s][t], g0part[s & 3][t], 0, 0, 0);
    // the contribution of every group of this batch to dG0 (zeros for an empty group next to a live one): summed over i1 by fast3_finalize_kernel.
    // accumulator row 4 hi + r = rho = (group rho / q0, core row rho % q0), column c = 16 t + lo
#pragma unroll
    for (int t = 0; t < C::RT1; ++t) {
      const f32x4 sum = (g0part[0][t] + g0part[1][t]) + (g0part[2][t] + g0part[3][t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rho = 4 * hi + r;
        const uint32_t gi = k0 + (uint32_t)(rho / Q0);
        const bool on = rho / Q0 < GM && gi < n_here && 16 * t + lo < R1;
        buf_store1(r_part, on ? (g_begin + gi) * (uint32_t)(C::ROW0 * 4) + 4u * ((rho % Q0) * R1 + 16 * t + lo) : kOob, sum[r]);
      }
    }
    k0 = k_next;
  }
  // this slice's dG1[i1] slab: every element is written (zeros when the slice holds no id)
  float* dst = plan.g1part + ((size_t)slice * p1 + i1) * C::ROW1;
#pragma unroll
  for (int t = 0; t < C::RT1; ++t)
#pragma unroll
    for (int nt = 0; nt < C::NT1; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * t + 4 * hi + r;
        if (c < R1 && (C::N1 % 16 == 0 || 16 * nt + lo < C::N1)) dst[c * C::N1 + 16 * nt + lo] = g1acc[t][nt][r];
      }
}

template <int Q0, int Q1, int Q2, int R1, int R2>
__global__ __launch_bounds__((EpiCfg<Q0, Q1, Q2, R1, R2>::WAVES * 64)) void fast3_group_epilogue_kernel(
    const float* __restrict__ G0, const float* __restrict__ G1, uint32_t p0, uint32_t p1, uint32_t gpw, uint32_t slices, GroupPlan plan) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  constexpr int EW = EpiCfg<Q0, Q1, Q2, R1, R2>::WAVES;
  __shared__ __attribute__((aligned(16))) float dpbuf_all[EW][16 * (C::N1 + 4)];
  __shared__ __attribute__((aligned(16))) float g1buf[R1 * C::LDG];
  const uint32_t wave = threadIdx.x >> 6;
  float* dpbuf = dpbuf_all[wave];
  const uint32_t nsb = (slices + EW - 1u) / EW;   // workgroups per i1 (1-D grid: consecutive workgroups share an i1)
  const uint32_t i1_blk = blockIdx.x / nsb;
  const uint32_t slice = (blockIdx.x - i1_blk * nsb) * EW + wave;
  if (sparse_groups(plan, p0 * p1))
    epilogue_unit<Q0, Q1, Q2, R1, R2, true>(G0, G1, p0, p1, gpw, plan, dpbuf, g1buf, slice, slices, i1_blk);
  else
    epilogue_unit<Q0, Q1, Q2, R1, R2, false>(G0, G1, p0, p1, gpw, plan, dpbuf, g1buf, slice, slices, i1_blk);
}

// D. finalize (and, in the fused modes, the optimiser step: every core element is produced exactly once here, so
// the update needs no gradient buffer and no extra launch): dG2 = sum of the per-tile slabs; dG0[i0] = sum over i1 of the per-group contributions of the
// non-empty groups; dG1 = sum of the per-slice slabs.  A workgroup owns 32 consecutive outputs; its 8 lane
// rows split the terms, so every load instruction reads 128 contiguous bytes per row and many
// are in flight; the 8 partial sums meet in LDS.  Every output is written exactly once.
__device__ __forceinline__ void finalize_emit(const FusedUpdate& upd, int t, float* __restrict__ grad, int idx, float g) {
  if (upd.w[0] == nullptr) {   // dense mode: the gradient itself (eps = 1: added to what an earlier piece of the call left)
    grad[idx] = upd.eps != 0.f ? grad[idx] + g : g;
  } else if (upd.st[0] == nullptr) {   // fused SGD (tt_embeddings_cuda.cu:381-397), every row
    upd.w[t][idx] -= upd.lr * g;
  } else {                             // fused Adagrad (tt_embeddings_cuda.cu:399-419)
    const float s2 = upd.st[t][idx] + g * g;
    upd.st[t][idx] = s2;
    upd.w[t][idx] -= upd.lr * g / (sqrtf(s2) + upd.eps);
  }
}

__global__ __launch_bounds__(256) void fast3_finalize_kernel(GroupPlan plan, int tiles, int slices, int p0, int p1,
                                                             int g2_floats, int row0, int g1_floats, int q2, int r2,
                                                             float* __restrict__ dG0, float* __restrict__ dG1,
                                                             float* __restrict__ dG2, FusedUpdate upd, int all_parts) {
  __shared__ float part[8][33];
  const int x = threadIdx.x & 31, y = threadIdx.x >> 5;
  const int n0 = p0 * row0;
  // the form the epilogue kernel took; the wide-rank chain (all_parts) writes every dG1 slab and the dG0 parts of the non-empty groups
  const bool sparse_any = sparse_groups(plan, (uint32_t)(p0 * p1));
  // a poisoned plan (a bounded wait of the grouping pass ran out): every gradient element this kernel produces is NaN.  In
  // the fused modes NaN would go into the parameters and the Adagrad state, with nothing left to rerun: when the host hears
  // of the fault anyway (the pinned word exists: ttemb_status / the next call return TTEMB_E_HIP) the update is SKIPPED and
  // parameters and state stay as they were -- the NaN forward rows and the error are loud enough.  Without the host word (a
  // graph captured before ttemb_init) NaN parameters remain the only signal.
  const bool poisoned = plan_poisoned(plan, (uint32_t)(p0 * p1));
  const bool loud = poisoned && plan.fault_host != nullptr;
  if (upd.poison_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && (loud || !upd.sticky)) *upd.poison_out = loud ? 1u : 0u;
  if (loud && upd.w[0] != nullptr) return;   // (wave- and grid-uniform)
  const float poison = poisoned ? __uint_as_float(0x7fc00000u) : 0.f;
  const bool sparse = all_parts ? true : sparse_any;          // dG0 parts: skip the groups without ids by their counts
  const bool sparse_g1 = all_parts ? false : sparse_any;      // dG1 slabs: skip the slices without ids by their flags
  // dG1 has few terms per output (one per slice): one thread per FOUR consecutive outputs (16-byte loads; a row of G1 is a
  // whole number of them), 1024 outputs per workgroup, the workgroups after those of dG2 / dG0.  (One output per thread
  // was 180 000 tiny workgroups for the 183 MB of dG1 at rank 256: the kernel was bound by their launch rate.)
  const int wg_g2 = (g2_floats + 31) / 32, wg_g0 = (n0 + 31) / 32;
  const int wg_a = wg_g2 + wg_g0;
#ifndef TTEMB_FINALIZE_SCALAR
  if (p1 >= 256 && (int)blockIdx.x >= wg_g2 && (int)blockIdx.x < wg_a) {
    // dG0[i0] = sum over i1 of the parts of the groups (i1, i0), tables with many i1 (papers100M: 560): 32 consecutive outputs
    // per workgroup as 8 threads x 16 bytes, 32 lane rows split the p1 terms -- a thread has p1 / 32 loads, six in flight.
    // (One output per thread and 8 lane rows -- the form below -- is p1 / 8 loads in rounds of eight: this kernel is a chain of
    // memory round trips.  papers100M r32, 819 200 ids: backward 1 063 -> 1 028 us; the products table, p1 = 140, has two
    // rounds either way and keeps the scalar form: 11.2 against 11.5 us.)
    __shared__ float4 part4[4][8];
    const int xb = threadIdx.x & 7, yr = threadIdx.x >> 3;
    const int o = ((int)blockIdx.x - wg_g2) * 32 + 4 * xb;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < n0) {
      const int i0 = o / row0, c = o - i0 * row0;   // (row0 is a multiple of 4: the four outputs share their i0)
#ifndef TTEMB_FIN_U
#define TTEMB_FIN_U 6
#endif
      constexpr int U6 = TTEMB_FIN_U;
      for (int i1 = yr; i1 < p1; i1 += 32 * U6) {
        bool on[U6];
        float4 v[U6];
#pragma unroll
        for (int u = 0; u < U6; ++u) {   // dense form: every group has a part (zeros for an empty one); sparse: empty groups have none
          const int g = (i1 + 32 * u) * p0 + i0;
          on[u] = i1 + 32 * u < p1 && (!sparse || plan.counts[g] != 0u);
        }
#pragma unroll
        for (int u = 0; u < U6; ++u) {
          const int g = (i1 + 32 * u) * p0 + i0;
          const size_t row = (all_parts & 2) ? (size_t)i0 * p1 + (i1 + 32 * u) : (size_t)g;   // (2: the parts lie by i0 -- GroupFuse::parts_by_i0)
          v[u] = on[u] ? *reinterpret_cast<const float4*>(plan.g0part + row * row0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U6; ++u) {
          s4.x += v[u].x; s4.y += v[u].y; s4.z += v[u].z; s4.w += v[u].w;
        }
      }
    }
#pragma unroll
    for (int d = 8; d < kWave; d <<= 1) {   // the eight lane rows of a wavefront
      s4.x += __shfl_xor(s4.x, d, kWave); s4.y += __shfl_xor(s4.y, d, kWave);
      s4.z += __shfl_xor(s4.z, d, kWave); s4.w += __shfl_xor(s4.w, d, kWave);
    }
    if ((threadIdx.x & 63) < 8) part4[threadIdx.x >> 6][xb] = s4;
    __syncthreads();
    if (threadIdx.x < 8 && o < n0) {
      const float4 a = part4[0][xb], b = part4[1][xb], c4 = part4[2][xb], d4 = part4[3][xb];
      finalize_emit(upd, 0, dG0, o + 0, (a.x + b.x) + (c4.x + d4.x) + poison);
      finalize_emit(upd, 0, dG0, o + 1, (a.y + b.y) + (c4.y + d4.y) + poison);
      finalize_emit(upd, 0, dG0, o + 2, (a.z + b.z) + (c4.z + d4.z) + poison);
      finalize_emit(upd, 0, dG0, o + 3, (a.w + b.w) + (c4.w + d4.w) + poison);
    }
    return;
  }
#endif
  if ((int)blockIdx.x >= wg_a) {
    const int o = (((int)blockIdx.x - wg_a) * 256 + (int)threadIdx.x) * 4;
    if (o >= g1_floats) return;
    const int i1 = o / (g1_floats / p1);
    float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
    // (eight slabs in flight per thread: the kernel is a chain of memory round trips -- 16 slices took four of them)
    for (int t = 0; t < slices; t += 8) {
      bool on[8];
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)   // slices without ids wrote nothing (sparse form)
        on[u] = t + u < slices && (!sparse_g1 || plan.epi_live[(size_t)(t + u) * p1 + i1] != 0u);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = on[u] ? *reinterpret_cast<const float4*>(plan.g1part + (size_t)(t + u) * g1_floats + o) : make_float4(0.f, 0.f, 0.f, 0.f);
      tot.x += ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
      tot.y += ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
      tot.z += ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
      tot.w += ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
    }
    finalize_emit(upd, 1, dG1, o + 0, tot.x + poison);
    finalize_emit(upd, 1, dG1, o + 1, tot.y + poison);
    finalize_emit(upd, 1, dG1, o + 2, tot.z + poison);
    finalize_emit(upd, 1, dG1, o + 3, tot.w + poison);
    return;
  }
  // (dG2 outputs come first, 32 per workgroup; dG0's workgroups start at a 32-aligned output of their own)
  const int e = (int)blockIdx.x < wg_g2 ? (blockIdx.x * 32 + x < g2_floats ? (int)(blockIdx.x * 32 + x) : g2_floats + n0)
                                        : g2_floats + ((int)blockIdx.x - wg_g2) * 32 + x;
  float s = 0.f;
  // U independent partial sums per thread keep that many loads in flight (a single running sum issues them one by one)
  constexpr int U = 8;
  if (e < g2_floats) {
    // the loads of a step go into registers first, then into the sum (accumulating load by load made every one of
    // them wait for the one before: the slabs were written on other XCDs, each dependent step is a trip to memory;
    // 8 per step measured best, 32 thrashes on shapes with hundreds of tiles)
    constexpr int U2 = 8;
    for (int t = y; t < tiles; t += 8 * U2) {
      float v[U2];
#pragma unroll
      for (int u = 0; u < U2; ++u) v[u] = t + 8 * u < tiles ? plan.g2part[(size_t)(t + 8 * u) * g2_floats + e] : 0.f;
#pragma unroll
      for (int u = 0; u < U2; u += 4) s += (v[u] + v[u + 1]) + (v[u + 2] + v[u + 3]);
    }
  } else if (e < g2_floats + n0) {
    const int o = e - g2_floats;
    const int i0 = o / row0, c = o - i0 * row0;
    for (int i1 = y; i1 < p1; i1 += 8 * U) {
      bool on[U];
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {   // dense form: every group has a part (zeros for an empty one); sparse: empty groups have none
        const int g = (i1 + 8 * u) * p0 + i0;
        on[u] = i1 + 8 * u < p1 && (!sparse || plan.counts[g] != 0u);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = (i1 + 8 * u) * p0 + i0;
        const size_t row = (all_parts & 2) ? (size_t)i0 * p1 + (i1 + 8 * u) : (size_t)g;
        v[u] = on[u] ? plan.g0part[row * row0 + c] : 0.f;
      }
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  part[y][x] = s;
  __syncthreads();
  if (y == 0) {
    float tot = poison;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += part[k][x];
    if (e < g2_floats) {  // the slabs hold rows as [kk][c2] (see fast3_bwd_chunk_kernel); dG2 rows are [c2][kk]
      const int row2 = q2 * r2;
      const int i2 = e / row2, w = e - i2 * row2;
      const int kk = w / r2, c2 = w - kk * r2;
      finalize_emit(upd, 2, dG2, i2 * row2 + c2 * q2 + kk, tot);
    } else if (e < g2_floats + n0) {
      finalize_emit(upd, 0, dG0, e - g2_floats, tot);
    }
  }
}

#include "ttemb_wide3.inc"

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
// The (q, ranks) shapes with an instantiated chain: the three BASELINE.json configurations first, then the other
// 3-core shapes the reference's run scripts train with (q = 4,4,8 / 4,5,5 / 8,4,4 at rank 16; the rank sweep of
// the products shape; a last column tile of q1 r2 that is not full is masked: q1 = 5 at rank 8).
#define TTEMB_FAST3_SHAPES(X) \
  X(4, 5, 5, 16, 16)          \
  X(4, 4, 8, 8, 8)            \
  X(8, 4, 4, 32, 32)          \
  X(4, 4, 8, 16, 16)          \
  X(8, 4, 4, 16, 16)          \
  X(4, 5, 5, 32, 32)          \
  X(4, 4, 8, 32, 32)          \
  X(5, 4, 5, 16, 16)          \
  X(5, 5, 4, 16, 16)          \
  X(5, 5, 4, 32, 32)          \
  X(5, 5, 4, 8, 8)            \
  X(4, 5, 5, 8, 8)            \
  X(8, 1, 16, 16, 16)         \
  X(10, 1, 10, 16, 16)        \
  X(16, 1, 8, 16, 16)

// the wide-rank chain (ttemb_wide3.inc): the upper half of the reference's rank sweep
#define TTEMB_WIDE3_SHAPES(X) \
  X(5, 5, 4, 64, 64)          \
  X(5, 5, 4, 128, 128)        \
  X(5, 5, 4, 256, 256)        \
  X(4, 4, 8, 64, 64)          \
  X(4, 4, 8, 128, 128)        \
  X(4, 4, 8, 256, 256)

static bool shape_is(const DevShape& s, int q0, int q1, int q2, int r1, int r2) {
  return s.q[0] == q0 && s.q[1] == q1 && s.q[2] == q2 && s.R[1] == r1 && s.R[2] == r2;
}

static bool classify(const DevShape& s) {
  if (s.T != 3) return false;
  if ((long long)s.L[0] * s.p[0] >= 0x7fffffffll) return false;  // ids must fit the uint32 sort key
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return true;
  TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  return false;
}

static bool wide(const DevShape& s) {
  if (s.T != 3) return false;
  if ((long long)s.L[0] * s.p[0] >= 0x7fffffffll) return false;
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return true;
  TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
  return false;
}

bool fast3_supported(const DevShape& s) { return classify(s) || wide(s); }
bool fast3_wide(const DevShape& s) { return wide(s); }

static int64_t num_groups(const DevShape& s) { return (int64_t)s.p[0] * s.p[1]; }

// the grouping pass: slices of the id list; ranges of 2^shift groups each
static int chain_cus();
static int sort_slices(int64_t nnz) {   // one 1024-thread workgroup per slice, at most one per CU (a CU takes one at a time)
  const int64_t s = (nnz + kSliceIds - 1) / kSliceIds;
  const int64_t most = chain_cus() < kMaxSlices ? chain_cus() : kMaxSlices;
  return (int)(s < 1 ? 1 : (s > most ? most : s));
}
static int sort_shift(int64_t G) {
  int shift = 0;
  while (((G + (int64_t(1) << shift) - 1) >> shift) > kMaxRanges) ++shift;   // as many ranges as fit: <= kMaxRanges
  return shift;
}
static int sort_ranges(int64_t G) { const int sh = sort_shift(G); return (int)((G + (int64_t(1) << sh) - 1) >> sh); }

// measured on the products shapes (tools/crossover.py, forward + dense backward): 4 096 ids 92 vs 107 us, 8 192 ids
// 101 vs 197 us for the grouped path vs the wave-per-id kernels; the grouped path's fixed cost grows with the number
// of groups (epilogue / finalize walk all of them)
bool fast3_pays(const DevShape& s, int64_t nnz) {
  if (wide(s)) {
    // wide-rank chain against the per-bag kernels (tools/crossover.py --paths per_bag fast3 on the products table, 17 500
    // groups, forward + dense backward; profiles/r02_wide_crossover.txt): rank 64 crosses at 1 300 (q = 5,5,4) / 2 700 ids
    // (q = 4,4,8), rank 128 below 128 / near 800, rank 256 never (32 ids: 0.52 against 1.37 ms -- at that rank the per-bag
    // backward's zero-fill and atomics on the 183 MB of dG1 cost more than the whole grouped chain).  The grouped side's
    // cost follows the number of non-empty groups (compacted GEMMs), its floor the size of the cores it has to write
    if (s.R[2] >= 256) return nnz >= 1;
    return nnz * (s.R[2] >= 128 ? 24 : 8) >= num_groups(s);
  }
  const int64_t by_groups = num_groups(s) / 4;
  return nnz >= (by_groups > 4096 ? by_groups : 4096);
}

// the chain kernels address every table through 32-bit byte offsets (buffer descriptors): 4 GiB each; the dG2
// reduce keeps two counters per i2 in LDS (p2 <= 4096: 36 KB); p1 is a grid.y extent
// the table itself: what no sub-batching can cure
static bool fits_shape(const DevShape& s) {
  const int64_t lim = int64_t(1) << 31;
  return num_groups(s) * (int64_t)s.q[0] * s.q[1] * s.R[2] * 4 < lim && s.p[2] <= 4096 && s.p[1] < 65536 &&
         num_groups(s) <= (int64_t)kMaxRanges * 4096;   // the grouping pass: <= 512 ranges of <= 4096 groups (64 KB of LDS)
}
// DIAGNOSTIC (ttemb_set_spin_limit): tries of the grouping pass's bounded waits; 0 = the default, negative = none at all
// (every wait expires: how a test reaches the fault path)
static std::atomic<int64_t> g_spin_limit{0};
void fast3_set_spin_limit(int64_t tries) { g_spin_limit.store(tries); }
static uint32_t spin_limit() {
  const int64_t t = g_spin_limit.load();
  // default 2^20 tries (~1.5 s of s_sleep + load per waiter): an expiry is an ERROR now (NaN results), so the bound is generous --
  // it only has to end a wait that will never be answered
  return t == 0 ? (1u << 20) : (t < 0 ? 0u : (t > 0x7fffffffll ? 0x7fffffffu : (uint32_t)t));
}
// DIAGNOSTIC (ttemb_set_piece_limits): tests cut small calls into pieces with it; 0 = the hardware's limits
static std::atomic<int64_t> g_piece_rows{0}, g_piece_ids{0};
void fast3_set_piece_limits(int64_t rows, int64_t ids) {
  g_piece_rows.store(rows > 0 ? rows : 0);
  g_piece_ids.store(ids > 0 ? ids : 0);
}
// one piece: `nnz` ids whose bags span `B` rows
static bool fits_piece(const DevShape& s, int64_t nnz, int64_t B) {
  const int64_t lim = int64_t(1) << 31;   // an offset of 2 GiB marks "no row" in the chain kernels (kOobBase)
  const int64_t tr = g_piece_rows.load(), ti = g_piece_ids.load();
  if ((tr > 0 && B > tr) || (ti > 0 && nnz > ti)) return false;
  return B * s.D * 4 < lim && B < (int64_t(1) << 24) && (wide(s) || nnz * (int64_t)s.row_len[2] * 4 < lim) &&
         nnz < (int64_t(1) << 26);        // 26-bit range counters under their epoch tags
}
// rows / ids of a piece of a call that is cut up (struct Piece): the row window the chain kernels address, and a cap on the ids
// that keeps a piece's tables (the E table of the unfused backward: ids x r2 q2 floats) under 2 GiB and the workspace modest
static int64_t piece_rows(const DevShape& s) {
  const int64_t by_bytes = ((int64_t(1) << 31) - 1) / ((int64_t)s.D * 4), by_field = (int64_t(1) << 24) - 1, t = g_piece_rows.load();
  const int64_t hw = by_bytes < by_field ? by_bytes : by_field;
  return t > 0 && t < hw ? t : hw;
}
static int64_t piece_ids(const DevShape& s) {
  const int64_t by_e = ((int64_t(1) << 31) - 1) / ((int64_t)s.row_len[2] * 4), cap = int64_t(4) << 20, t = g_piece_ids.load();
  const int64_t hw = (wide(s) || by_e > cap) ? cap : by_e;
  return t > 0 && t < hw ? t : hw;
}
static int piece_slots(const DevShape& s, int64_t nnz, int64_t B) {
  const int64_t li = piece_ids(s), lr = piece_rows(s);
  return (int)((nnz + li - 1) / li + (B + lr - 1) / lr);
}
bool fast3_fits(const DevShape& s, int64_t nnz, int64_t B) { return fits_shape(s) && fits_piece(s, nnz, B); }
// a call past one piece runs as several (needs the bag boundaries: `offsets`)
bool fast3_fits_in_pieces(const DevShape& s, int64_t nnz, int64_t B) { return fits_shape(s) && nnz > 0 && B > 0 && nnz <= 0x7fffffffll; }
static int64_t pieces_head_bytes(const DevShape& s, int64_t nnz, int64_t B) { return align256((int64_t)piece_slots(s, nnz, B) * (int64_t)sizeof(Piece)); }

#ifndef TTEMB_ROWS_B
#define TTEMB_ROWS_B 4096   // (2 048 until round 5: papers100M r32 at 819 200 / 2.4 M ids writes and re-reads half / 40 % fewer 205 KB slabs: backward -10 / -29 us)
#endif
constexpr int kRowsB = TTEMB_ROWS_B;  // E rows per workgroup of the dG2 reduce, at most
// One round of workgroups when the batch allows it: the kernel is as long as its longest workgroup, so 409 600 rows
// go as 256 tiles of 1 600 (every CU busy) rather than 200 tiles of 2 048; small batches keep >= 512 rows per tile
// (the finalize kernel reads one p2 x row slab per tile).
static int reduce_rows(int64_t nnz) {
#ifdef TTEMB_REDUCE_FIXED
  return kRowsB;
#endif
  int64_t r = ((nnz + 255) / 256 + 63) / 64 * 64;
  return (int)(r < 512 ? 512 : (r > kRowsB ? kRowsB : r));
}
static int64_t reduce_tiles(int64_t nnz) { const int r = reduce_rows(nnz); return (nnz + r - 1) / r; }
// slabs of more than 256 KB are not replicated per tile: one shared slab, float atomics (see the reduce kernel)
static bool shared_slab(const DevShape& s) { return (int64_t)s.p[2] * s.row_len[2] * 4 > (256 << 10); }
// Persistent launch of a chain kernel: `wgs_per_cu` workgroups of kChainWaves independent wavefronts per CU (fewer
// when their LDS does not fit 160 KB that often), every wavefront takes an equal share of the chunk table.
#ifndef TTEMB_BWD_WGS
#define TTEMB_BWD_WGS 2
#endif
#ifndef TTEMB_FWD_WGS
#define TTEMB_FWD_WGS 4
#endif
constexpr int kBwdWgsPerCu = TTEMB_BWD_WGS, kFwdWgsPerCu = TTEMB_FWD_WGS;
static int chain_cus() { return device_cus(); }
static int chain_grid(const void* kernel, size_t lds, int wgs_per_cu, LdsGate* gate, unsigned* grid) {
  int rc = allow_big_lds(kernel, lds, gate, "chain kernel");
  if (rc) return rc;
  int fit = (int)(kCuLds / (lds ? lds : 1));
  if (fit > wgs_per_cu) fit = wgs_per_cu;
  *grid = (unsigned)(chain_cus() * (fit < 1 ? 1 : fit));
  return TTEMB_OK;
}


// The fused form of the backward chunk kernel (dG2 reduced inside it, no E table): a row of the slab must fit a
// wavefront's lanes, the slab's rows the register quads of eight wavefronts, and eight wavefronts' staging the CU's LDS
static int64_t max_chunks(const DevShape& s, int64_t nnz);
static size_t bwd_wave_lds_floats(const DevShape& s) {   // = Cfg::PB_FLOATS + BB2_FLOATS + OB_FLOATS of the shape
  const int M2 = s.q[0] * s.q[1], R2 = s.R[2], Q2 = s.q[2], ROW2 = R2 * Q2, D = s.D;
  const int LDPB = (R2 % 32 == 0) ? R2 + 16 : R2;
  const int LDBB = (Q2 % 2 == 1 && ROW2 % 32 == 16) ? ROW2 : ROW2 + 4;
  const int DPAD = (M2 + 3) / 4 * 4 * Q2;
  const int LDOB = (((DPAD > D ? DPAD : D) + 15) / 32) * 32 + 16;
  return (size_t)((M2 * LDPB + 3) / 4 * 4) + (size_t)kChunk * LDBB + (size_t)kChunk * LDOB;
}
static bool fused_dg2(const DevShape& s) {
#ifdef TTEMB_NO_FUSE
  return false;
#endif
  const int lpr = s.row_len[2] / 4;
  if (lpr < 1 || !fuse_shape(s.R[2], s.row_len[2])) return false;   // (the shapes FUSE = true is compiled for)
  const int gpw = kWave / lpr;
  const int fq = fuse_quads(s.q[0] * s.q[1], s.row_len[2]);
  if (s.p[2] > kFuseWaves * gpw * fq) return false;
  return (size_t)kFuseWaves * (bwd_wave_lds_floats(s) + (size_t)kChunk * s.row_len[2]) * 4 +
             (size_t)(s.row_len[2] + kFuseWaves * (1 + 32 * kFuseSub) + 3 + kFuseWaves * gpw * fq) * 4 <= kCuLds;
}
static int64_t fused_tiles(const DevShape& s, int64_t nnz) {   // workgroups: at least ~2 chunks per wavefront, at most one per CU
  const int64_t t = (max_chunks(s, nnz) + 2 * kFuseSub * kFuseWaves - 1) / (2 * kFuseSub * kFuseWaves);
  const int64_t most = (int64_t)chain_cus() * (8 / kFuseWaves);
  return t < 1 ? 1 : (t > most ? most : t);
}
// Wide-rank chain: the backward that keeps its dG2 slices in LDS (wide3_bwd_slab_kernel, no E table).  A wavefront's slice is
// p2 rows of 16 q2 + 4 floats + a tag word per row; `wpb` wavefronts per workgroup (one workgroup per CU) own consecutive rank
// tiles, `shares` workgroup sets split the chunk table -- one slab of partial dG2 per share.
static int wide_slab_js(const DevShape& s) {   // q2 = 8: a tile's slice is split over two wavefronts by k2 halves
#ifdef TTEMB_WIDE_SLAB_JS1
  return 1;
#endif
  return s.q[2] % 8 == 0 ? 2 : 1;
}
static int wide_slab_wpb(const DevShape& s) {
  const int64_t per_wave = (int64_t)s.p[2] * (16 * (s.q[2] / wide_slab_js(s)) + 4 + 1) * 4;
  int wpb = (int)(kCuLds / (per_wave > 0 ? per_wave : 1));
  wpb = wpb >= 4 ? 4 : (wpb >= 2 ? 2 : wpb);
  const int units = s.R[2] / 16 * wide_slab_js(s);
  while (wpb > 1 && units % wpb != 0) wpb >>= 1;
  return wpb;
}
static int wide_slab_shares(const DevShape& s) {
  const int wpb = wide_slab_wpb(s);   // (0: a slice of this p2 does not fit a CU's LDS -- wide_slab() is false then)
  const int tgroups = (s.R[2] / 16 * wide_slab_js(s)) / (wpb > 0 ? wpb : 1);
  const int sh = chain_cus() / (tgroups > 0 ? tgroups : 1);
  return sh < 1 ? 1 : sh;
}
// taken when the E table it saves is large against the slabs it writes (16 384 ids at rank 256: 1 176 us against 1 114 with the
// table; 409 600 ids: 2 068 against 2 367): from 8 ids per (share, i2) slab row on
// DIAGNOSTIC (ttemb_set_wide_slab_min_ids): the call size from which the slab kernel is taken; 0 = the rule, 1 = always (how
// the unit tests -- a few thousand ids -- reach the kernel), a huge value = never
static std::atomic<int64_t> g_wide_slab_min{0};
void fast3_set_wide_slab_min_ids(int64_t ids) { g_wide_slab_min.store(ids > 0 ? ids : 0); }
static bool wide_slab(const DevShape& s, int64_t nnz) {
#ifdef TTEMB_WIDE_E_TABLE   // (A/B: the round-4 form -- E table + reduce kernel)
  return false;
#endif
  if (!wide(s) || wide_slab_wpb(s) < 2) return false;
  const int64_t forced = g_wide_slab_min.load();
  const int64_t least = forced > 0 ? forced : (int64_t)8 * wide_slab_shares(s) * s.p[2];
  return nnz >= least;
}
static int64_t slab_count(const DevShape& s, int64_t nnz) {
  if (wide_slab(s, nnz)) return wide_slab_shares(s);
  return fused_dg2(s) ? fused_tiles(s, nnz) : (shared_slab(s) ? 1 : reduce_tiles(nnz));
}
// the epilogue cuts the i0 range of every i1 into ~kEpiSlices slices of `gpw` groups (one wavefront each)
static int epi_groups_per_wave(const DevShape& s) {
  int gpw = (s.p[0] + kEpiSlices - 1) / kEpiSlices;
  return gpw < 1 ? 1 : (gpw > 64 ? 64 : gpw);
}
// wide-rank chain: dG1[i1] = G0^T . dP[i1] has (r1 / 64)(q1 r2 / 64) p1 tiles of K = p0 q0; K is split until ~4 tiles per SIMD exist
static int wide_k_chunk(const DevShape& s) {
  const int64_t tiles = (int64_t)(s.R[1] / 64) * (s.q[1] * s.R[2] / 64) * s.p[1];
  int64_t parts = (4096 + tiles - 1) / tiles;
  parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
  const int64_t K = (int64_t)s.p[0] * s.q[0];
  return (int)(((K + parts - 1) / parts + 31) / 32 * 32);
}
static int64_t wide_rows_stride(const DevShape& s) { return ((int64_t)s.p[0] * s.q[0] + 63) / 64 * 64 + 64; }
static int epi_slices(const DevShape& s) {   // (the wide-rank chain: one dG1 slab per K part)
  if (wide(s)) return (int)(((int64_t)s.p[0] * s.q[0] + wide_k_chunk(s) - 1) / wide_k_chunk(s));
  const int g = epi_groups_per_wave(s);
  return (s.p[0] + g - 1) / g;
}

// What forward and backward share ("plan"): grouped (i2, row) pairs, group sizes / starts, the chunk table and
// the prefix products.  It lives in a caller buffer when one is given, else in the workspace.  A backward that
// is handed the forward's plan differentiates the chain at the prefix products the forward used (what autograd's
// saved tensors mean); one that builds its own plan forms them from the cores as they are then.
static int64_t max_chunks(const DevShape& s, int64_t nnz) {
  const int64_t G = num_groups(s);
  return nnz / kChunk + (nnz < G ? nnz : G) + 1;
}

int64_t fast3_plan_bytes(const DevShape& s, int64_t nnz) {
  const int64_t G = num_groups(s);
  return 2 * align256(nnz * 4) + align256((G + 1) * 4) + align256((G + 3) * 8) + align256(max_chunks(s, nnz) * 16) +
         align256(G * (int64_t)s.q[0] * s.q[1] * s.R[2] * 4);
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
  base += align256((G + 3) * 8);   // (+ the plan's tag and its fault word)
  pl->ctab = (uint4*)base;
  base += align256(max_chunks(s, nnz) * 16);
  pl->ptab = (float*)base;
}

// workspace layout: [plan part unless external] [grouping scratch] [backward tables]
static int64_t carve_workspace(const DevShape& s, int64_t nnz, bool bwd, bool plan_inside, bool need_grouping,
                               char* base, GroupPlan* pl) {
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
    uint32_t* in[6];
    for (int i = 0; i < 6; ++i) in[i] = (uint32_t*)take(nnz * 4);
    uint32_t* sh = (uint32_t*)take((int64_t)sort_slices(nnz) * sort_ranges(G) * 4);
    uint32_t* rs = (uint32_t*)take((sort_ranges(G) + 1) * 4);
    uint32_t* gs = (uint32_t*)take(G * 4);
    uint64_t* rp = (uint64_t*)take(sort_ranges(G) * 16);
    if (pl) {
      pl->grp_in = in[0];
      pl->i2_in = in[1];
      pl->vals_in = in[2];
      pl->grp_mid = in[3];
      pl->i2_mid = in[4];
      pl->vals_mid = in[5];
      pl->shist = sh;
      pl->rstart = rs;
      pl->gstamp = gs;
      pl->rpub = rp;
    }
  }
  if (wide(s)) {   // the lists of non-empty rows the compacted GEMMs walk (rebuilt from the plan's counts by every call)
    uint32_t* wr = (uint32_t*)take((int64_t)s.p[1] * wide_rows_stride(s) * 4);
    uint32_t* wn = (uint32_t*)take((int64_t)s.p[1] * 4);
    if (pl) {
      pl->wrows = wr;
      pl->wnrows = wn;
    }
  }
  if (bwd) {
    float* e = (float*)take((fused_dg2(s) || wide_slab(s, nnz)) ? 0 : nnz * (int64_t)s.row_len[2] * 4);   // no E table in the fused forms
    float* d = (float*)take(G * (int64_t)s.q[0] * s.q[1] * s.R[2] * 4);
    float* g2 = (float*)take(slab_count(s, nnz) * (int64_t)s.p[2] * s.row_len[2] * 4);
    float* g0 = (float*)take(G * (int64_t)s.row_len[0] * 4);
    float* g1 = (float*)take((int64_t)epi_slices(s) * s.p[1] * s.row_len[1] * 4);
    uint32_t* el = (uint32_t*)take((int64_t)epi_slices(s) * s.p[1] * 4);
    if (pl) {
      pl->g1part = g1;
      pl->epi_live = el;
      pl->etab = e;
      pl->dptab = d;
      pl->g2part = g2;
      pl->g0part = g0;
    }
  }
  return off;
}

int64_t fast3_workspace_bytes(const DevShape& s, int32_t op, int64_t nnz, int64_t B) {
  if (!fits_piece(s, nnz, B)) {   // several pieces: the piece table, then one piece's tables (plan inside)
    const int64_t li = piece_ids(s);
    return pieces_head_bytes(s, nnz, B) + carve_workspace(s, nnz < li ? nnz : li, op == TTEMB_OP_BACKWARD, true, true, nullptr, nullptr) + 256;
  }
  return carve_workspace(s, nnz, op == TTEMB_OP_BACKWARD, true, true, nullptr, nullptr) + 256;
}

// fill plan->{i2s, vals, counts, gpre, ctab} from the ids
static int run_prefix(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, hipStream_t st);
static int run_spread_place_prefix(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int slices,
                                   uint32_t per_slice, int ranges, int shift, hipStream_t st);

static int group_ids(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* rowidx,
                     const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev, int64_t B, float* zero_out,
                     bool with_prefix, GroupPlan* plan, hipStream_t st) {
  const int64_t G = num_groups(s);
  const uint32_t sentinel = (uint32_t)((unsigned long long)s.L[0] * s.p[0]);
  const int slices = sort_slices(nnz), shift = sort_shift(G), ranges = sort_ranges(G);
  const uint32_t per_slice = (uint32_t)((nnz + slices - 1) / slices);
  const size_t span = (size_t)1 << shift;
  if (ranges > kMaxRanges || span * 16 > 64 * 1024) return fail(TTEMB_E_UNSUPPORTED, "too many (i0, i1) groups for the grouping pass");
  if (nnz >= (int64_t(1) << 26)) return fail(TTEMB_E_UNSUPPORTED, "too many ids for the grouping pass (26-bit range counters)");
  hipLaunchKernelGGL(fast3_decode_kernel, dim3((unsigned)slices), dim3(kSortThreads), 0, st, indices, rowidx, offsets,
                     (uint32_t)nnz, per_slice, nnz_dev, B, s.D, zero_out, sentinel, (uint32_t)s.p[0], (uint32_t)s.p[1],
                     (uint32_t)s.p[2], (uint32_t)shift, (uint32_t)ranges, *plan);
  int rc = check_hip(hipGetLastError(), "fast3_decode_kernel");
  if (rc) return rc;
  if (with_prefix && !wide(s)) return run_spread_place_prefix(s, cores, *plan, nnz, slices, per_slice, ranges, shift, st);
  hipLaunchKernelGGL(fast3_spread_kernel, dim3((unsigned)slices), dim3(kSortThreads), 0, st, (uint32_t)nnz, per_slice,
                     (uint32_t)shift, (uint32_t)ranges, *plan);
  rc = check_hip(hipGetLastError(), "fast3_spread_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(fast3_place_kernel, dim3((unsigned)ranges), dim3(kRangeThreads), span * 16, st, (uint32_t)nnz,
                     (uint32_t)max_chunks(s, nnz), (uint32_t)G, (uint32_t)shift, *plan);
  rc = check_hip(hipGetLastError(), "fast3_place_kernel");
  if (rc == TTEMB_OK && with_prefix && wide(s)) rc = run_prefix(s, cores, *plan, st);   // wide ranks: the prefix products are a GEMM of their own
  return rc;
}

// resolve where the plan lives, carve the workspace, group the ids unless a ready plan was passed
// plan_state: 0 = build the whole plan (grouping + prefix products), 1 = grouping only (the id-only half of a
// two-phase forward), 2 = the plan is grouped, add the prefix products, 3 = the plan is complete
static int prepare(const DevShape& s, const CorePtrs& cores, bool bwd, const int64_t* indices, const int64_t* rowidx,
                   const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev, int64_t B, float* zero_out, void* ws,
                   int64_t ws_bytes, void* plan_buf, int64_t plan_bytes, int plan_state, GroupPlan* plan, hipStream_t st,
                   void* header, const Piece* piece = nullptr, bool prefix_in_chain = false) {
  memset(plan, 0, sizeof(*plan));
  plan->piece = piece;
  plan->fault_host = fault_word(st);
  plan->spin_limit = spin_limit();
  // the words that outlive a call -- the grouping pass's epoch and its pre-tagged range counters -- sit in the header of
  // the caller's workspace (kFast3HeaderBytes, at the same address for every op on that workspace): tables carved per call
  // would be overwritten by the next call's other tables (the backward's gradients land where the forward grouped)
  if (plan_state <= 1 && header == nullptr) return fail(TTEMB_E_WORKSPACE, "fast path needs the workspace header");
  plan->epochs = reinterpret_cast<uint64_t*>(header);
  plan->rcount = plan->epochs ? plan->epochs + 2 : nullptr;
  plan->ticket = header ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(header) + kHeaderPoisonOffset + 8) : nullptr;
#ifdef TTEMB_PLACE_TICKET_ALWAYS
  plan->use_ticket = 1u;
#else
  plan->use_ticket = plan_state == 1 ? 1u : 0u;   // (see place_range)
#endif
  static_assert(kHeaderPoisonOffset + 16 <= kFast3HeaderBytes, "the header holds the epoch words, every bank of range counters, the poison word and the ticket");
  const bool external = plan_buf != nullptr && plan_bytes >= fast3_plan_bytes(s, nnz);
#ifdef TTEMB_KEEP_P_ALWAYS   // (A/B: the P table written by every forward)
  plan->keep_p = 1u;
#else
  plan->keep_p = external ? 1u : 0u;
#endif
  if (plan_state != 0 && !external) return fail(TTEMB_E_BADARG, "this call needs a plan buffer of ttemb_plan_bytes() bytes");
  const bool reuse = plan_state >= 2;
  const int64_t need = carve_workspace(s, nnz, bwd, !external, !reuse, reinterpret_cast<char*>(ws), plan);
  if (need > 0 && ws == nullptr) return fail(TTEMB_E_WORKSPACE, "fast path needs a workspace");
  if (need > ws_bytes)
    return fail(TTEMB_E_WORKSPACE, "fast path needs %lld workspace bytes, got %lld", (long long)need, (long long)ws_bytes);
  if (external) carve_plan_part(s, nnz, reinterpret_cast<char*>(plan_buf), plan);
  if (plan_state == 3) return TTEMB_OK;
  // (the lookup half of a two-phase forward whose chain kernel forms the prefix products launches nothing here: the bracket of
  //  the id-only half -- the grouping pass -- stays the one ttemb_profile_read(3) returns)
  if (plan_state == 2 && prefix_in_chain) return TTEMB_OK;
  profile_begin(3, st);
  int rc = TTEMB_OK;
  // building the whole plan: the prefix products ride in the last grouping launch
  // (prefix_in_chain: the forward chain kernel forms the prefix products itself -- fast3_forward_pfuse_kernel)
  if (plan_state <= 1) rc = group_ids(s, cores, indices, rowidx, offsets, nnz, nnz_dev, B, zero_out, plan_state == 0 && !prefix_in_chain, plan, st);
  if (rc == TTEMB_OK && plan_state == 2 && !prefix_in_chain) rc = run_prefix(s, cores, *plan, st);
  profile_end(3, st);
  return rc;
}

// P of every non-empty group from the cores as they are now
template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_prefix_t(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, hipStream_t st) {
  hipLaunchKernelGGL((fast3_prefix_kernel<Q0, Q1, Q2, R1, R2>),
                     dim3((unsigned)((s.p[0] + kPrefixGroups - 1) / kPrefixGroups), (unsigned)s.p[1]), dim3(64), 0, st,
                     cores.c[0], cores.c[1], (uint32_t)s.p[0], 0u, plan);   // (on the counters of a complete grouping)
  return check_hip(hipGetLastError(), "fast3_prefix_kernel");
}

// ---- wide-rank chain: launches ----
template <bool A_KC, bool B_KC, int COMPACT>
static int run_wide_gemm(WideGemm g, const DevShape& s, const GroupPlan& plan, hipStream_t st, const char* what, uint32_t k_chunk = 0,
                         uint32_t c_slice = 0) {
  const uint32_t units = g.tiles_m * g.tiles_n;
  g.k_chunk = k_chunk ? k_chunk : (g.K + 31u) / 32u * 32u;
  g.c_slice = c_slice;
  g.rows = plan.wrows;
  g.n_rows = plan.wnrows;
  g.rows_stride = (uint32_t)wide_rows_stride(s);
  g.full = (uint32_t)(s.p[0] * s.q[0]);
  g.batches = (uint32_t)s.p[1];
  g.splits = (g.K + g.k_chunk - 1) / g.k_chunk;
  const dim3 grid((((units + 3) / 4) * g.batches * g.splits + 7u) / 8u * 8u);   // 1-D, a multiple of the 8 XCDs
  hipLaunchKernelGGL((wide3_gemm_kernel<A_KC, B_KC, COMPACT>), grid, dim3(256), 0, st, g);
  return check_hip(hipGetLastError(), what);
}

static int run_wide_rows(const DevShape& s, const GroupPlan& plan, hipStream_t st) {
  hipLaunchKernelGGL(wide3_rows_kernel, dim3((unsigned)s.p[1]), dim3(256), 0, st, plan, (uint32_t)s.p[0], (uint32_t)s.q[0],
                     (uint32_t)wide_rows_stride(s), plan.wrows, plan.wnrows);
  return check_hip(hipGetLastError(), "wide3_rows_kernel");
}

// P[i1] (p0 q0 x q1 r2) = G0 (p0 q0 x r1) . G1[i1] (r1 x q1 r2), the rows of the groups that hold an id
static int run_prefix_wide(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, hipStream_t st) {
  int rc = run_wide_rows(s, plan, st);
  if (rc) return rc;
  const uint32_t M = (uint32_t)(s.p[0] * s.q[0]), N = (uint32_t)(s.q[1] * s.R[2]), K = (uint32_t)s.R[1];
  WideGemm g;
  g.A = cores.c[0]; g.B = cores.c[1]; g.C = plan.ptab;
  g.K = K;
  g.lda = K; g.ldb = N; g.ldc = N;
  g.a_batch = 0; g.b_batch = K * N; g.c_batch = M * N;
  g.a_bytes = M * K * 4; g.b_bytes = K * N * 4; g.c_bytes = M * N * 4;
  g.tiles_m = (M + 63) / 64; g.tiles_n = N / 64;
  return run_wide_gemm<true, false, 1>(g, s, plan, st, "wide3_gemm_kernel (prefix)");
}

static int run_prefix(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, hipStream_t st) {
  if (wide(s)) return run_prefix_wide(s, cores, plan, st);
  if (classify(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_prefix_t<a, b, c, d, e>(s, cores, plan, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
}

// spread step + first half of the prefix units, then place step + the rest
template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_spread_place_prefix_t(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int slices,
                                     uint32_t per_slice, int ranges, int shift, hipStream_t st) {
  const unsigned units = (unsigned)((s.p[0] + kPrefixGroups - 1) / kPrefixGroups) * (unsigned)s.p[1];
  // A CU takes ONE 1024-thread workgroup at a time: a spread launch of more workgroups than CUs runs its surplus as a second
  // round (split 50 %: 200 + 70 workgroups on 256 CUs, 17.3 us against 11.8 at 25 %).  A quarter of the units rides with the
  // spread step -- what fits the CUs its slices leave free -- the rest with the place step (409 600 ids, spread + place:
  // 25.6 / 25.2 / 30.1 / 30.0 / 29.6 us at 0 / 25 / 50 / 75 / 100 %; as launches of their own 10.7 + 11.2 + 8.8).
  const unsigned per_a = kSortThreads / kWave, per_b = kRangeThreads / kWave;
  const unsigned room = chain_cus() > slices ? (unsigned)(chain_cus() - slices) * per_a : 0u;
  unsigned first = PrefixRides<Q0, Q1, Q2, R1, R2>::value ? units / 4 : 0u;
  first = first > room ? room : first;
  hipLaunchKernelGGL((fast3_spread_prefix_kernel<Q0, Q1, Q2, R1, R2>), dim3((unsigned)slices + (first + per_a - 1) / per_a),
                     dim3(kSortThreads), 0, st, (uint32_t)slices, (uint32_t)nnz, per_slice, (uint32_t)shift, (uint32_t)ranges, first,
                     cores.c[0], cores.c[1], (uint32_t)s.p[0], (uint32_t)s.p[1], plan);
  int rc = check_hip(hipGetLastError(), "fast3_spread_prefix_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((fast3_place_prefix_kernel<Q0, Q1, Q2, R1, R2>), dim3((unsigned)ranges + (units - first + per_b - 1) / per_b),
                     dim3(kRangeThreads), ((size_t)16 << shift), st, (uint32_t)ranges, (uint32_t)nnz, (uint32_t)max_chunks(s, nnz),
                     (uint32_t)num_groups(s), (uint32_t)shift, first, cores.c[0], cores.c[1], (uint32_t)s.p[0], (uint32_t)s.p[1], plan);
  return check_hip(hipGetLastError(), "fast3_place_prefix_kernel");
}

static int run_spread_place_prefix(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int slices,
                                   uint32_t per_slice, int ranges, int shift, hipStream_t st) {
  if (classify(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_spread_place_prefix_t<a, b, c, d, e>(s, cores, plan, nnz, slices, per_slice, ranges, shift, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward_direct(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                              float* output, hipStream_t st);


template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                       float* output, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
#ifdef TTEMB_FWD_DIRECT   // experiment: the LDS-free forward of the wide ranks on a narrow-rank shape (slower there, see ttemb_wide3.inc)
  if constexpr (DirectCfg<Q0, Q1, Q2, R1, R2>::ok) return run_forward_direct<Q0, Q1, Q2, R1, R2>(s, cores, plan, nnz, B, output, st);
#endif
  const size_t lds = (size_t)kChainWaves * C::WAVE_FLOATS * sizeof(float);
  static LdsGate lds_ok;
  unsigned grid = 0;
  int rc = chain_grid(reinterpret_cast<const void*>(fast3_forward_kernel<Q0, Q1, Q2, R1, R2>), lds, kFwdWgsPerCu, &lds_ok, &grid);
  if (rc) return rc;
  profile_begin(0, st);
  hipLaunchKernelGGL((fast3_forward_kernel<Q0, Q1, Q2, R1, R2>), dim3(grid), dim3(kChainWaves * 64), lds, st, cores.c[2],
                     plan, (uint32_t)num_groups(s), (uint32_t)s.p[2], (uint32_t)nnz, output, (uint32_t)(B * s.D * 4));
  profile_end(0, st);
  return check_hip(hipGetLastError(), "fast3_forward_kernel");
}

// Frontiers with few ids per group form the prefix products inside the chain kernel (fast3_forward_pfuse_kernel).  The rule
// looks at the call's size only (the occupancy of the groups is known on the device alone): fewer than `limit` ids per group
// on average.  Measured crossovers (profiles/r04_pfuse_forward.txt, grouping + chain kernel, uniform ids): q = 4,4,8 rank 16
// at 9-10 ids per group; q = 4,5,5 at ranks 16 / 32 and q = 5,5,4 rank 8 at 16-20 (gains of 2-4 % above 8); q0 = 8 (two
// groups per tile: the prefix launch costs most there) -11 ... -17 % still at 15 ids per group, on the papers100M table
// -17 % at 8.8.  Hence 8, and 16 for q0 = 8.  (TTEMB_PFUSE_IDS: 0 switches the route off, another value replaces both.)
#ifndef TTEMB_PFUSE_IDS
#define TTEMB_PFUSE_IDS -1
#endif
constexpr int64_t kPFuseIdsPerGroup = TTEMB_PFUSE_IDS;
static int64_t pfuse_limit(const DevShape& s) { return kPFuseIdsPerGroup >= 0 ? kPFuseIdsPerGroup : (s.q[0] == 8 ? 16 : 8); }
template <int Q0, int Q1, int Q2, int R1, int R2>
static bool pfuse_shape() { return PFuseCfg<Q0, Q1, Q2, R1, R2>::ok; }
static bool pfuse_pays(const DevShape& s, int64_t nnz) {
  if (pfuse_limit(s) <= 0 || !classify(s)) return false;
  bool ok = false;
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) ok = pfuse_shape<a, b, c, d, e>();
  TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  return ok && nnz < pfuse_limit(s) * num_groups(s);
}

bool fast3_prefix_in_chain(const DevShape& s, int64_t nnz, int64_t B) { return fits_piece(s, nnz, B) && pfuse_pays(s, nnz); }
// The lookup half of a two-phase forward (ttemb_forward_lookup: the data-parallel step groups the ids, finishes the
// all-reduce + update, then looks up) cannot let its prefix products ride in the grouping launches -- they read the cores the
// update has just written -- so they are a launch of their own there (8.6 us + its ramp at 409 600 ids on the products table).
// The chain kernel that forms them itself is EQUAL to prefix launch + chain kernel at the products frontier's 23 ids per group
// (profiles/r04_pfuse_forward.txt) when the launch rides for free; against a launch of its own it wins well past that: taken
// up to 64 ids per group on average.
static bool pfuse_pays_after_grouping(const DevShape& s, int64_t nnz) {
#ifdef TTEMB_NO_PFUSE_PHASE2
  return false;
#endif
  if (pfuse_limit(s) <= 0 || !classify(s)) return false;
  bool ok = false;
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) ok = pfuse_shape<a, b, c, d, e>();
  TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  return ok && nnz < 64 * num_groups(s);
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward_pfuse(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                             float* output, hipStream_t st) {
  using PC = PFuseCfg<Q0, Q1, Q2, R1, R2>;
  if constexpr (PC::ok) {
    const size_t lds = (size_t)kChainWaves * PC::WAVE_FLOATS * sizeof(float);
    static LdsGate lds_ok;
    unsigned grid = 0;
    int rc = chain_grid(reinterpret_cast<const void*>(fast3_forward_pfuse_kernel<Q0, Q1, Q2, R1, R2>), lds, kFwdWgsPerCu, &lds_ok, &grid);
    if (rc) return rc;
    const uint64_t magic = (uint64_t(1) << 40) / (uint64_t)s.p[0] + 1ull;   // g / p0 = (g * magic) >> 40 for g p0 < 2^40
    profile_begin(0, st);
    hipLaunchKernelGGL((fast3_forward_pfuse_kernel<Q0, Q1, Q2, R1, R2>), dim3(grid), dim3(kChainWaves * 64), lds, st, cores.c[0],
                       cores.c[1], cores.c[2], plan, (uint32_t)num_groups(s), (uint32_t)s.p[0], magic, (uint32_t)s.p[2], (uint32_t)nnz,
                       output, (uint32_t)(B * s.D * 4));
    profile_end(0, st);
    return check_hip(hipGetLastError(), "fast3_forward_pfuse_kernel");
  } else {
    return fail(TTEMB_E_HIP, "internal: pfuse_pays() and PFuseCfg out of step");
  }
}

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_forward_direct(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                            float* output, hipStream_t st) {
  const uint32_t waves = (uint32_t)max_chunks(s, nnz);   // the chunk count is known on the device only: surplus wavefronts leave at once
#ifndef TTEMB_WIDE_FWD_SIMPLE
  if constexpr (DirectCfg<Q0, Q1, Q2, R1, R2>::NS >= 2) {   // wide ranks: the persistent chain
#ifndef TTEMB_WIDE_FWD_WGS
#define TTEMB_WIDE_FWD_WGS 4
#endif
    profile_begin(0, st);
    hipLaunchKernelGGL((direct_forward_chain_kernel<Q0, Q1, Q2, R1, R2>), dim3((unsigned)(chain_cus() * TTEMB_WIDE_FWD_WGS)), dim3(256), 0, st,
                       cores.c[2], plan, (uint32_t)num_groups(s), (uint32_t)s.p[2], output, (uint32_t)(B * s.D * 4));
    profile_end(0, st);
    return check_hip(hipGetLastError(), "direct_forward_chain_kernel");
  }
#endif
  profile_begin(0, st);
  hipLaunchKernelGGL((direct_forward_kernel<Q0, Q1, Q2, R1, R2>), dim3((waves + 3) / 4), dim3(256), 0, st, cores.c[2], plan,
                     (uint32_t)num_groups(s), (uint32_t)s.p[2], output, (uint32_t)(B * s.D * 4));
  profile_end(0, st);
  return check_hip(hipGetLastError(), "direct_forward_kernel");
}

// backward of the wide-rank chain: per-group products, the E reduce (a column slice of 256 floats per launch row),
// dG1 / dG0 as GEMMs over the dP table, the shared finalize kernel
template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_backward_wide(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                             const float* d_output, const CorePtrsMut& d_cores, const FusedUpdate& upd, hipStream_t st) {
  using C = WideCfg<Q0, Q1, Q2, R1, R2>;
  const int64_t G = num_groups(s);
  profile_begin(1, st);
  int rc = run_wide_rows(s, plan, st);   // (the workspace of this call; the forward's lists are not part of the plan)
  if (rc) return rc;
  static_assert(C::ROW2 % 256 == 0, "the E reduce takes 256 columns per launch row");
  const int tiles = (int)reduce_tiles(nnz);
  const size_t reduce_lds = (size_t)(2 * s.p[2] + 1) * 4 + kRowsB * 2;
  if (wide_slab(s, nnz)) {   // chunk products and the dG2 reduction in one launch: the tile slices of dG2 stay in LDS (no E table)
    constexpr int kJS = Q2 % 8 == 0 ? 2 : 1;
    const int js = wide_slab_js(s), wpb = wide_slab_wpb(s), shares = wide_slab_shares(s);
    const size_t lds = (size_t)wpb * s.p[2] * (16 * (Q2 / js) + 4 + 1) * 4;
    const dim3 grid((unsigned)(shares * (C::RT2 * js / wpb)));
    static LdsGate lds_ok[2];
    profile_begin(2, st);
    if (js == kJS && kJS == 2) {
      if constexpr (kJS == 2) {
        rc = allow_big_lds(reinterpret_cast<const void*>(wide3_bwd_slab_kernel<Q0, Q1, Q2, R1, R2, kJS>), lds, &lds_ok[1], "wide3_bwd_slab_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL((wide3_bwd_slab_kernel<Q0, Q1, Q2, R1, R2, kJS>), grid, dim3(256), lds, st, cores.c[2], (uint32_t)G, (uint32_t)s.p[2],
                           d_output, (uint32_t)(B * s.D * 4), plan, (uint32_t)wpb, (uint32_t)shares);
      }
    } else {
      rc = allow_big_lds(reinterpret_cast<const void*>(wide3_bwd_slab_kernel<Q0, Q1, Q2, R1, R2, 1>), lds, &lds_ok[0], "wide3_bwd_slab_kernel");
      if (rc) return rc;
      hipLaunchKernelGGL((wide3_bwd_slab_kernel<Q0, Q1, Q2, R1, R2, 1>), grid, dim3(256), lds, st, cores.c[2], (uint32_t)G, (uint32_t)s.p[2],
                         d_output, (uint32_t)(B * s.D * 4), plan, (uint32_t)wpb, (uint32_t)shares);
    }
    profile_end(2, st);
    rc = check_hip(hipGetLastError(), "wide3_bwd_slab_kernel");
    if (rc) return rc;
  } else {
  profile_begin(2, st);
  hipLaunchKernelGGL((wide3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2>), dim3((unsigned)((G * C::TS + 3) / 4)), dim3(256), 0, st, cores.c[2],
                     (uint32_t)G, (uint32_t)s.p[2], d_output, (uint32_t)(B * s.D * 4), plan);
  profile_end(2, st);
  rc = check_hip(hipGetLastError(), "wide3_bwd_chunk_kernel");
  if (rc) return rc;
#if defined(TTEMB_ABL) && (TTEMB_ABL & 2048)   // (ablation 2048: no E stores in the wide chunk kernel, no reduce launch: what the E round trip costs; timing only)
  if (nnz < 0) {
#else
  if (shared_slab(s)) {
#endif
    rc = launch_zero(plan.g2part, (size_t)s.p[2] * C::ROW2 * 4, st, "zero the shared dG2 slab");
    if (rc) return rc;
    hipLaunchKernelGGL((fast3_dg2_reduce_kernel<256, kRowsB, NWB, true>), dim3((unsigned)tiles, C::ROW2 / 256), dim3(NWB * 64), reduce_lds,
                       st, plan, (int)G, (uint32_t)s.p[2], (uint32_t)reduce_rows(nnz), (uint32_t)C::ROW2);
  } else {
#if defined(TTEMB_ABL) && (TTEMB_ABL & 2048)
    if (nnz < 0)
#endif
    hipLaunchKernelGGL((fast3_dg2_reduce_kernel<256, kRowsB, NWB, false>), dim3((unsigned)tiles, C::ROW2 / 256), dim3(NWB * 64), reduce_lds,
                       st, plan, (int)G, (uint32_t)s.p[2], (uint32_t)reduce_rows(nnz), (uint32_t)C::ROW2);
  }
  rc = check_hip(hipGetLastError(), "fast3_dg2_reduce_kernel (wide)");
  if (rc) return rc;
  }
  const uint32_t M = (uint32_t)(s.p[0] * Q0), N1 = (uint32_t)C::N1;
  {  // dG1[i1] (r1 x q1 r2) = G0^T (r1 x p0 q0) . dP[i1] (p0 q0 x q1 r2)
    WideGemm g;
    g.A = cores.c[0]; g.B = plan.dptab; g.C = plan.g1part;
    g.K = M;
    g.lda = R1; g.ldb = N1; g.ldc = N1;
    g.a_batch = 0; g.b_batch = M * N1; g.c_batch = (uint32_t)C::ROW1;
    g.a_bytes = M * R1 * 4; g.b_bytes = M * N1 * 4; g.c_bytes = (uint32_t)C::ROW1 * 4;
    g.tiles_m = R1 / 64; g.tiles_n = N1 / 64;
    rc = run_wide_gemm<false, false, 2>(g, s, plan, st, "wide3_gemm_kernel (dG1)", (uint32_t)wide_k_chunk(s),
                                        (uint32_t)s.p[1] * (uint32_t)C::ROW1);
    if (rc) return rc;
  }
  {  // dG0 parts of i1 (p0 q0 x r1) = dP[i1] (p0 q0 x q1 r2) . G1[i1]^T (q1 r2 x r1); the finalize kernel sums over i1
    WideGemm g;
    g.A = plan.dptab; g.B = cores.c[1]; g.C = plan.g0part;
    g.K = N1;
    g.lda = N1; g.ldb = N1; g.ldc = R1;
    g.a_batch = M * N1; g.b_batch = (uint32_t)C::ROW1; g.c_batch = M * R1;
    g.a_bytes = M * N1 * 4; g.b_bytes = (uint32_t)C::ROW1 * 4; g.c_bytes = M * R1 * 4;
    g.tiles_m = (M + 63) / 64; g.tiles_n = R1 / 64;
#ifndef TTEMB_WIDE_DG0_DIRECT
    if (R1 % kLdsGemmBlock == 0 && N1 % kLdsGemmK == 0) {   // ranks 128 / 256: through LDS (see wide3_gemm_lds_kernel)
      static LdsGate lds_ok;
      rc = allow_big_lds(reinterpret_cast<const void*>(wide3_gemm_lds_kernel), kLdsGemmBytes, &lds_ok, "wide3_gemm_lds_kernel");
      if (rc) return rc;
      g.tiles_m = (M + kLdsGemmBlock - 1) / kLdsGemmBlock;   // (128 x 128 blocks)
      g.tiles_n = R1 / kLdsGemmBlock;
      g.rows = plan.wrows;
      g.n_rows = plan.wnrows;
      g.rows_stride = (uint32_t)wide_rows_stride(s);
      g.full = M;
      g.batches = (uint32_t)s.p[1];
      g.splits = 1;
      g.k_chunk = N1;
      g.c_slice = 0;
      const dim3 grid((g.tiles_m * g.tiles_n * g.batches + 7u) / 8u * 8u);   // 1-D, a multiple of the 8 XCDs
      hipLaunchKernelGGL(wide3_gemm_lds_kernel, grid, dim3(256), kLdsGemmBytes, st, g);
      rc = check_hip(hipGetLastError(), "wide3_gemm_lds_kernel (dG0)");
    } else
#endif
    rc = run_wide_gemm<true, true, 1>(g, s, plan, st, "wide3_gemm_kernel (dG0)");
    if (rc) return rc;
  }
  {
    const int g2_floats = s.p[2] * C::ROW2, g1_floats = s.p[1] * C::ROW1;
    const int wgs = (g2_floats + 31) / 32 + (s.p[0] * C::ROW0 + 31) / 32 + (g1_floats + 1023) / 1024;
    hipLaunchKernelGGL(fast3_finalize_kernel, dim3((unsigned)wgs), dim3(256), 0, st, plan, (int)slab_count(s, nnz), epi_slices(s), s.p[0], s.p[1],
                       g2_floats, (int)C::ROW0, g1_floats, Q2, R2, d_cores.c[0], d_cores.c[1], d_cores.c[2], upd, 1);
  }
  profile_end(1, st);
  return check_hip(hipGetLastError(), "fast3_finalize_kernel (wide)");
}

int launch_forward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev,
                         int64_t B, float* output, bool zero_rows, void* ws, int64_t ws_bytes, void* plan_buf,
                         int64_t plan_bytes, int phase, hipStream_t st, void* header) {
  if (nnz <= 0) return TTEMB_OK;
  if (!fits_piece(s, nnz, B)) {
    // The call is cut into pieces (struct Piece).  The id-only half of a two-phase forward does nothing then, the lookup half
    // is the whole forward; the caller's plan buffer is not used (a plan describes one piece; the backward regroups).
    if (phase == 1) return TTEMB_OK;
    if (offsets == nullptr) return fail(TTEMB_E_UNSUPPORTED, "a call of this size needs the bag boundaries (offsets)");
    const int slots = piece_slots(s, nnz, B);
    const int64_t head = pieces_head_bytes(s, nnz, B), li = piece_ids(s), np = nnz < li ? nnz : li;
    if (ws == nullptr || ws_bytes < head) return fail(TTEMB_E_WORKSPACE, "forward needs room for the piece table");
    Piece* tab = reinterpret_cast<Piece*>(ws);
    hipLaunchKernelGGL(plan_pieces_kernel, dim3(1), dim3(64), 0, st, offsets, B, nnz, nnz_dev, (long long)li, (long long)piece_rows(s),
                       s.D, slots, tab);
    int rc = check_hip(hipGetLastError(), "plan_pieces_kernel");
    for (int k = 0; k < slots && rc == TTEMB_OK; ++k) {
      GroupPlan plan;
      rc = prepare(s, cores, false, indices, rowidx, offsets, np, nullptr, B, zero_rows ? output : nullptr,
                   reinterpret_cast<char*>(ws) + head, ws_bytes - head, nullptr, 0, 0, &plan, st, header, tab + k);
      if (rc) break;
      rc = fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
      if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_forward_direct<a, b, c, d, e>(s, cores, plan, np, piece_rows(s), output, st);
        TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
      } else {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_forward<a, b, c, d, e>(s, cores, plan, np, piece_rows(s), output, st);
        TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
      }
    }
    return rc;
  }
  GroupPlan plan;
  // a whole forward (phase 0) on a frontier with few ids per group: the chain kernel forms the prefix products itself
  const bool pfuse = (phase == 0 && pfuse_pays(s, nnz)) || (phase == 2 && pfuse_pays_after_grouping(s, nnz));
  int rc = prepare(s, cores, false, indices, rowidx, offsets, nnz, nnz_dev, B, zero_rows ? output : nullptr, ws, ws_bytes,
                   plan_buf, plan_bytes, phase, &plan, st, header, nullptr, pfuse);   // phase 0 / 1 / 2 = whole forward / ids only / lookup on a grouped plan
  if (rc || phase == 1) return rc;
  if (pfuse) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_forward_pfuse<a, b, c, d, e>(s, cores, plan, nnz, B, output, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_forward_direct<a, b, c, d, e>(s, cores, plan, nnz, B, output, st);
    TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  if (classify(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_forward<a, b, c, d, e>(s, cores, plan, nnz, B, output, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
}

// The backward chunk kernel that also forms the per-group products (GF): shapes with at least two groups per MFMA tile and
// q1 r2 a multiple of 16 (the K permutation of its dG0 product moves 16-byte pieces).  Taken by frontiers with few ids per
// group (the rule of the forward that forms its own prefix products) whose dG2 reduction is not fused into the chunk kernel.
#ifndef TTEMB_GFUSE_IDS
#define TTEMB_GFUSE_IDS -1
#endif
constexpr int64_t kGFuseIdsPerGroup = TTEMB_GFUSE_IDS;   // (0 switches the route off, another value replaces the rule)
template <int Q0, int Q1, int Q2, int R1, int R2>
struct GFuseCfg {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  static constexpr bool ok = 16 / Q0 >= 2 && C::N1 % 16 == 0 && Q1 > 1;
};
// ids per group up to which the route is taken.  At rank 32 it wins at every density that was measured (papers100M 1.5 ... 8.6
// ids per group and a METIS-like frontier of 130 per touched group; the products table at 5.6 / 7.7 / 23 uniform and 80 per
// group METIS-like: -5 ... -22 %, profiles/r05_gf_rule.txt): no limit there.  Rank <= 16 shapes reach this kernel only when
// their p2 is past the fused dG2 form; they keep the forward's rule (measured at ~3-4 ids per group only).
static int64_t gfuse_limit(const DevShape& s) {
  if (kGFuseIdsPerGroup >= 0) return kGFuseIdsPerGroup;
  return s.R[2] >= 32 ? (int64_t(1) << 40) : (s.q[0] == 8 ? 16 : 8);
}

static bool gfuse_pays(const DevShape& s, int64_t nnz) {   // (the rule run_backward applies, for ttemb_kernel_family)
  if (gfuse_limit(s) <= 0 || !classify(s) || fused_dg2(s)) return false;
  bool ok = false;
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) ok = GFuseCfg<a, b, c, d, e>::ok;
  TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  const int64_t G = num_groups(s);
  return ok && (gfuse_limit(s) >= (int64_t(1) << 40) || nnz < gfuse_limit(s) * G) && (uint64_t)G * (uint64_t)s.p[0] < (uint64_t(1) << 40);
}
bool fast3_group_products_in_chain(const DevShape& s, int64_t nnz, int64_t B) { return fits_piece(s, nnz, B) && gfuse_pays(s, nnz); }

template <int Q0, int Q1, int Q2, int R1, int R2>
static int run_backward(const DevShape& s, const CorePtrs& cores, const GroupPlan& plan, int64_t nnz, int64_t B,
                        const float* d_output, const CorePtrsMut& d_cores, const FusedUpdate& upd, hipStream_t st) {
  using C = Cfg<Q0, Q1, Q2, R1, R2>;
  const int64_t G = num_groups(s);
  profile_begin(1, st);
  int rc;
  constexpr size_t wave_lds = (size_t)(C::PB_FLOATS + C::BB2_FLOATS + C::OB_FLOATS) * sizeof(float);
  constexpr bool kCanFuse = fuse_shape(R2, C::ROW2);
  const bool fused = kCanFuse && fused_dg2(s);
  constexpr bool kCanGFuse = GFuseCfg<Q0, Q1, Q2, R1, R2>::ok;
  const bool gfuse = kCanGFuse && !fused && gfuse_pays(s, nnz);
  if constexpr (!kCanFuse) {
    if (fused_dg2(s)) return fail(TTEMB_E_HIP, "internal: fused_dg2() and fuse_shape() out of step");
  }
  if (fused) {   // chunk products and the dG2 reduction in one launch
    if (wave_lds != bwd_wave_lds_floats(s) * sizeof(float)) return fail(TTEMB_E_HIP, "internal: LDS size formula out of step");
    const size_t lds = kFuseWaves * (wave_lds + kChunk * C::ROW2 * sizeof(float)) +
                       (size_t)(C::ROW2 + kFuseWaves * (1 + 32 * kFuseSub) + 3 + kFuseWaves * (kWave / (C::ROW2 / 4)) * fuse_quads(C::M2, C::ROW2)) * sizeof(uint32_t);
    if constexpr (kCanFuse) {
      static LdsGate lds_ok;
      rc = allow_big_lds(reinterpret_cast<const void*>(fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, true>), lds, &lds_ok, "fused backward chunk kernel");
      if (rc) return rc;
      profile_begin(2, st);
      hipLaunchKernelGGL((fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, true>), dim3((unsigned)fused_tiles(s, nnz)), dim3(kFuseWaves * 64),
                         lds, st, cores.c[2], (uint32_t)G, (uint32_t)s.p[2], (uint32_t)nnz, d_output, (uint32_t)(B * s.D * 4), plan, GroupFuse{});
      profile_end(2, st);
      rc = check_hip(hipGetLastError(), "fast3_bwd_chunk_kernel (fused)");
      if (rc) return rc;
    }
  } else if (gfuse) {
    if constexpr (kCanGFuse) {   // chunk products AND the per-group products in one launch: no dP table, no epilogue launch
      const size_t lds = kChainWaves * (wave_lds + 16 * (C::N1 + 4) * sizeof(float));
      static LdsGate lds_ok;
      unsigned grid = 0;
      rc = chain_grid(reinterpret_cast<const void*>(fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, false, true>), lds, kBwdWgsPerCu, &lds_ok, &grid);
      if (rc) return rc;
      rc = launch_zero(plan.g1part, (size_t)s.p[1] * C::ROW1 * 4, st, "zero the dG1 slab");   // (the kernel adds with float atomics)
      if (rc) return rc;
      GroupFuse gf;
      gf.G0 = cores.c[0];
      gf.G1 = cores.c[1];
      gf.dg1 = plan.g1part;
      gf.p0 = (uint32_t)s.p[0];
      gf.p0_magic = (uint64_t(1) << 40) / (uint64_t)s.p[0] + 1ull;
      gf.p1 = (uint32_t)s.p[1];
      gf.parts_by_i0 = kPartsByI0 ? 1u : 0u;
      profile_begin(2, st);
      hipLaunchKernelGGL((fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, false, true>), dim3(grid), dim3(kChainWaves * 64), lds, st, cores.c[2],
                         (uint32_t)G, (uint32_t)s.p[2], (uint32_t)nnz, d_output, (uint32_t)(B * s.D * 4), plan, gf);
      profile_end(2, st);
      rc = check_hip(hipGetLastError(), "fast3_bwd_chunk_kernel (group products fused)");
      if (rc) return rc;
    }
  } else {
    const size_t lds = kChainWaves * wave_lds;
    static LdsGate lds_ok;
    unsigned grid = 0;
    rc = chain_grid(reinterpret_cast<const void*>(fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, false>), lds, kBwdWgsPerCu, &lds_ok, &grid);
    if (rc) return rc;
    profile_begin(2, st);
    hipLaunchKernelGGL((fast3_bwd_chunk_kernel<Q0, Q1, Q2, R1, R2, false>), dim3(grid), dim3(kChainWaves * 64), lds, st, cores.c[2],
                       (uint32_t)G, (uint32_t)s.p[2], (uint32_t)nnz, d_output, (uint32_t)(B * s.D * 4), plan, GroupFuse{});
    profile_end(2, st);
    rc = check_hip(hipGetLastError(), "fast3_bwd_chunk_kernel");
    if (rc) return rc;
  }
  const int tiles = (int)reduce_tiles(nnz);
  const size_t reduce_lds = (size_t)(2 * s.p[2] + 1) * 4 + kRowsB * 2;
  if (fused) {
    // nothing: the slabs are written
  } else if (shared_slab(s)) {
    rc = launch_zero(plan.g2part, (size_t)s.p[2] * C::ROW2 * 4, st, "zero the shared dG2 slab");
    if (rc) return rc;
    hipLaunchKernelGGL((fast3_dg2_reduce_kernel<C::ROW2, kRowsB, NWB, true>), dim3((unsigned)tiles), dim3(NWB * 64), reduce_lds, st,
                       plan, (int)G, (uint32_t)s.p[2], (uint32_t)reduce_rows(nnz), (uint32_t)C::ROW2);

  } else {
    hipLaunchKernelGGL((fast3_dg2_reduce_kernel<C::ROW2, kRowsB, NWB, false>), dim3((unsigned)tiles), dim3(NWB * 64), reduce_lds, st,
                       plan, (int)G, (uint32_t)s.p[2], (uint32_t)reduce_rows(nnz), (uint32_t)C::ROW2);
  }
  rc = check_hip(hipGetLastError(), "fast3_dg2_reduce_kernel");
  if (rc) return rc;
  const int gpw = epi_groups_per_wave(s), slices = gfuse ? 1 : epi_slices(s);
  if (!gfuse) {
    constexpr int EW = EpiCfg<Q0, Q1, Q2, R1, R2>::WAVES;
    const unsigned epi_blocks = (unsigned)((slices + EW - 1) / EW) * (unsigned)s.p[1];
    profile_begin(8, st);
#if defined(TTEMB_ABL) && (TTEMB_ABL & 512)
    if (fused)
#endif
    hipLaunchKernelGGL((fast3_group_epilogue_kernel<Q0, Q1, Q2, R1, R2>), dim3(epi_blocks),
                       dim3(EW * 64), 0, st, cores.c[0], cores.c[1], (uint32_t)s.p[0], (uint32_t)s.p[1], (uint32_t)gpw, (uint32_t)slices, plan);
    profile_end(8, st);
    rc = check_hip(hipGetLastError(), "fast3_group_epilogue_kernel");
    if (rc) return rc;
  }
  {
    // (group products fused: ONE dG1 slab, whole; dG0 parts of the non-empty groups only -- the form the wide chain leaves)
    const int g2_floats = s.p[2] * C::ROW2, g1_floats = s.p[1] * C::ROW1;
    const int wgs = (g2_floats + 31) / 32 + (s.p[0] * C::ROW0 + 31) / 32 + (g1_floats + 1023) / 1024;   // dG2 | dG0 | dG1
    profile_begin(9, st);
    hipLaunchKernelGGL(fast3_finalize_kernel, dim3((unsigned)wgs), dim3(256), 0, st, plan, (int)slab_count(s, nnz), slices,
                       s.p[0], s.p[1], g2_floats, (int)C::ROW0, g1_floats, Q2, R2, d_cores.c[0], d_cores.c[1], d_cores.c[2], upd, gfuse ? (kPartsByI0 ? 3 : 1) : 0);
    profile_end(9, st);
  }
  profile_end(1, st);
  return check_hip(hipGetLastError(), "fast3_finalize_kernel");
}

int launch_backward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                          const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev,
                          int64_t B, const float* d_output, const CorePtrsMut& d_cores, void* ws, int64_t ws_bytes,
                          const void* plan_buf, int64_t plan_bytes, hipStream_t st, const FusedUpdate* update, void* header) {
  FusedUpdate upd;
  memset(&upd, 0, sizeof(upd));
  if (update != nullptr) upd = *update;
  upd.poison_out = header != nullptr ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(header) + kHeaderPoisonOffset) : nullptr;
  upd.sticky = 0;
  static_assert(kHeaderPoisonOffset == 16 + (int64_t)kCountBanks * kMaxRanges * 8, "the poison word sits behind the range counters");
  // every core gradient is written whole by the finalize kernel (an empty call in a fused mode is a no-op)
  for (int t = 0; t < s.T; ++t) {
    if (nnz > 0 || update != nullptr) continue;
    int rc = launch_zero(d_cores.c[t], (size_t)s.p[t] * s.row_len[t] * 4, st, "zero d_core");
    if (rc) return rc;
  }
  if (nnz <= 0) return TTEMB_OK;
  if (!fits_piece(s, nnz, B)) {
    // piece by piece (struct Piece): every piece regroups its ids (a plan describes one piece) and ADDS its gradient to what
    // the pieces before it left -- the reference accumulates its batch_count chunks into d_tt_cores the same way and steps
    // once (tt_embeddings_cuda.cu:633-651).  The optimiser step is the caller's, on the summed gradient.
    if (update != nullptr) return fail(TTEMB_E_BADARG, "internal: a call in pieces writes gradients, the step follows");
    if (offsets == nullptr) return fail(TTEMB_E_UNSUPPORTED, "a call of this size needs the bag boundaries (offsets)");
    const int slots = piece_slots(s, nnz, B);
    const int64_t head = pieces_head_bytes(s, nnz, B), li = piece_ids(s), np = nnz < li ? nnz : li;
    if (ws == nullptr || ws_bytes < head) return fail(TTEMB_E_WORKSPACE, "backward needs room for the piece table");
    Piece* tab = reinterpret_cast<Piece*>(ws);
    hipLaunchKernelGGL(plan_pieces_kernel, dim3(1), dim3(64), 0, st, offsets, B, nnz, nnz_dev, (long long)li, (long long)piece_rows(s),
                       s.D, slots, tab);
    int rc = check_hip(hipGetLastError(), "plan_pieces_kernel");
    for (int k = 0; k < slots && rc == TTEMB_OK; ++k) {
      GroupPlan plan;
      rc = prepare(s, cores, true, indices, rowidx, offsets, np, nullptr, B, nullptr, reinterpret_cast<char*>(ws) + head, ws_bytes - head,
                   nullptr, 0, 0, &plan, st, header, tab + k);
      if (rc) break;
      upd.eps = k > 0 ? 1.f : 0.f;   // (dense mode: finalize adds to the gradient instead of writing it)
      upd.sticky = k > 0 ? 1 : 0;    // (a poisoned piece marks the whole call)
      rc = fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
      if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_backward_wide<a, b, c, d, e>(s, cores, plan, np, piece_rows(s), d_output, d_cores, upd, st);
        TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
      } else {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_backward<a, b, c, d, e>(s, cores, plan, np, piece_rows(s), d_output, d_cores, upd, st);
        TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
      }
    }
    return rc;
  }
  GroupPlan plan;
  int rc = prepare(s, cores, true, indices, rowidx, offsets, nnz, nnz_dev, B, nullptr, ws, ws_bytes,
                   const_cast<void*>(plan_buf), plan_bytes,
                   plan_buf != nullptr && plan_bytes >= fast3_plan_bytes(s, nnz) ? 3 : 0, &plan, st, header);
  if (rc) return rc;
  if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_backward_wide<a, b, c, d, e>(s, cores, plan, nnz, B, d_output, d_cores, upd, st);
    TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  if (classify(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) return run_backward<a, b, c, d, e>(s, cores, plan, nnz, B, d_output, d_cores, upd, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
}

// ---------------------------------------------------------------------------------
// A WINDOW of a longer id list: the bags [bag0, bag0 + B) of `offsets` and the ids that belong to them -- one table of a
// table-batched call (the reference passes `tableidx` to its kernels and never learns on the host where a table's ids begin,
// tt_embeddings_cuda.cu:1349-1365).  The window is a Piece whose bounds are read from `offsets` ON THE DEVICE; the launches
// are sized by `nnz`, the length of the whole list, and a window that turns out empty costs its launches and nothing else.
// `output` / `d_output` are the [bags_total][D] tensors of the whole call.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void window_piece_kernel(const int64_t* __restrict__ offsets, long long bag0, long long B, int D,
                                                          Piece* __restrict__ out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Piece pc;
  pc.pos0 = (long long)offsets[bag0];
  pc.count = (long long)offsets[bag0 + B] - pc.pos0;
  if (pc.count < 0) pc.count = 0;
  pc.rowbase = bag0;
  pc.zero0 = bag0;
  pc.zero1 = bag0 + B;
  pc.window_bytes = B * (long long)D * 4;
  *out = pc;
}
constexpr int64_t kWindowHeadBytes = 256;   // the Piece, in front of the call's tables
bool fast3_window_fits(const DevShape& s, int64_t nnz, int64_t bags_total, int64_t B) {
  return fits_shape(s) && (classify(s) || wide(s)) && nnz > 0 && B > 0 && fits_piece(s, nnz, B) && bags_total < 0x7fffffffll;
}
int64_t fast3_window_workspace_bytes(const DevShape& s, bool bwd, int64_t nnz) {
  return kWindowHeadBytes + carve_workspace(s, nnz, bwd, true, true, nullptr, nullptr) + 256;
}

int launch_forward_window_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets,
                                int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, float* output, void* ws, int64_t ws_bytes,
                                hipStream_t st, void* header) {
  if (!fast3_window_fits(s, nnz, bags_total, B)) return fail(TTEMB_E_UNSUPPORTED, "the grouped kernels do not cover this window (shape or size)");
  if (ws == nullptr || ws_bytes < kWindowHeadBytes) return fail(TTEMB_E_WORKSPACE, "forward needs room for the window");
  Piece* tab = reinterpret_cast<Piece*>(ws);
  hipLaunchKernelGGL(window_piece_kernel, dim3(1), dim3(64), 0, st, offsets, (long long)bag0, (long long)B, s.D, tab);
  int rc = check_hip(hipGetLastError(), "window_piece_kernel");
  if (rc) return rc;
  GroupPlan plan;
  rc = prepare(s, cores, false, indices, nullptr, offsets, nnz, nullptr, bags_total, output, reinterpret_cast<char*>(ws) + kWindowHeadBytes,
               ws_bytes - kWindowHeadBytes, nullptr, 0, 0, &plan, st, header, tab);
  if (rc) return rc;
  rc = fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_forward_direct<a, b, c, d, e>(s, cores, plan, nnz, B, output, st);
    TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
  } else {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_forward<a, b, c, d, e>(s, cores, plan, nnz, B, output, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return rc;
}

// `update` != null: the optimiser step of THIS table rides in the finalize kernel (a window is a whole table's share of the
// call: nothing of another call has to be summed first); null: the gradient is written to d_cores
int launch_backward_window_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets,
                                 int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, const float* d_output,
                                 const CorePtrsMut& d_cores, void* ws, int64_t ws_bytes, hipStream_t st, const FusedUpdate* update,
                                 void* header) {
  if (!fast3_window_fits(s, nnz, bags_total, B)) return fail(TTEMB_E_UNSUPPORTED, "the grouped kernels do not cover this window (shape or size)");
  if (ws == nullptr || ws_bytes < kWindowHeadBytes) return fail(TTEMB_E_WORKSPACE, "backward needs room for the window");
  FusedUpdate upd;
  memset(&upd, 0, sizeof(upd));
  if (update != nullptr) upd = *update;
  upd.poison_out = header != nullptr ? reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(header) + kHeaderPoisonOffset) : nullptr;
  upd.sticky = 0;
  if (update == nullptr) upd.eps = 0.f;   // (dense mode: the gradient is written, not added to an earlier piece's)
  Piece* tab = reinterpret_cast<Piece*>(ws);
  hipLaunchKernelGGL(window_piece_kernel, dim3(1), dim3(64), 0, st, offsets, (long long)bag0, (long long)B, s.D, tab);
  int rc = check_hip(hipGetLastError(), "window_piece_kernel");
  if (rc) return rc;
  GroupPlan plan;
  rc = prepare(s, cores, true, indices, nullptr, offsets, nnz, nullptr, bags_total, nullptr, reinterpret_cast<char*>(ws) + kWindowHeadBytes,
               ws_bytes - kWindowHeadBytes, nullptr, 0, 0, &plan, st, header, tab);
  if (rc) return rc;
  rc = fail(TTEMB_E_UNSUPPORTED, "fast3 path does not cover this shape");
  if (wide(s)) {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_backward_wide<a, b, c, d, e>(s, cores, plan, nnz, B, d_output, d_cores, upd, st);
    TTEMB_WIDE3_SHAPES(TTEMB_X)
#undef TTEMB_X
  } else {
#define TTEMB_X(a, b, c, d, e) if (shape_is(s, a, b, c, d, e)) rc = run_backward<a, b, c, d, e>(s, cores, plan, nnz, B, d_output, d_cores, upd, st);
    TTEMB_FAST3_SHAPES(TTEMB_X)
#undef TTEMB_X
  }
  return rc;
}

#include "ttemb_small3.inc"

}  // namespace ttemb
