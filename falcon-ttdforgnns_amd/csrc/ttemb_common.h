// Shared declarations for the libttemb_hip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/ttemb.h"

namespace ttemb {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMaxProbes = 3;      // reference MAX_PROBES, FBTT/tt_embeddings_cuda.cu:31
constexpr int64_t kEmptyKey = -1;  // reference UNUSED_KEY, FBTT/hashtbl_cuda_utils.cuh:100

// Device-side view of one table's factorisation (passed to kernels by value).
struct DevShape {
  int T;
  int p[TTEMB_MAX_CORES];
  int q[TTEMB_MAX_CORES];
  int R[TTEMB_MAX_CORES + 1];
  long long L[TTEMB_MAX_CORES];   // L[t] = prod(p[t+1:])
  int D;                          // prod(q)
  int row_len[TTEMB_MAX_CORES];   // floats per core row: R[t]*q[t]*R[t+1]
  int part_len[TTEMB_MAX_CORES];  // floats of the partial product after core t: q0..qt * R[t+1]
  int part_max;                   // max part_len over t < T-1 (and q0*R1)
};

struct CorePtrs {
  const float* c[TTEMB_MAX_CORES];
};
struct CorePtrsMut {
  float* c[TTEMB_MAX_CORES];
};

// host helpers (ttemb_api.hip)
int fail(int code, const char* fmt, ...);
int check_hip(hipError_t e, const char* what);
int make_dev_shape(const ttemb_shape_t* s, DevShape* out);
int current_path();
// measurement hook: bracket the main chain kernel with events when profiling is on
void profile_begin(int which, hipStream_t st);
void profile_end(int which, hipStream_t st);

// gfx950 has 160 KB of LDS per CU; a launch above the default 64 KB of dynamic LDS has to be allowed per kernel AND per
// device (hipFuncSetAttribute acts on the current device's function object; a process may drive several GPUs).  One gate
// per kernel: bit d = "allowed on device d"; devices past 63 ask the runtime every time.
constexpr size_t kLdsDefault = 64 * 1024, kCuLds = 160 * 1024;
struct LdsGate {
  std::atomic<uint64_t> devices{0};
};
int allow_big_lds(const void* kernel, size_t lds_bytes, LdsGate* gate, const char* what);
int device_cus();   // CUs of the current device (cached per device; 256 when it cannot be asked)

// generic kernels (ttemb_generic.hip)
int64_t generic_fwd_lds_bytes(const DevShape& s);
int64_t generic_bwd_lds_bytes(const DevShape& s);
int launch_forward_generic(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                           const int64_t* rowidx, const int64_t* offsets, int64_t B, int64_t nnz,
                           const int32_t* nnz_dev, float* output, hipStream_t st);
int launch_backward_generic(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                            const int64_t* rowidx, const int64_t* offsets, int64_t B, int64_t nnz,
                            const int32_t* nnz_dev, const float* d_output, const CorePtrsMut& d_cores, hipStream_t st);
// zero-fills the T gradient cores in one launch
int launch_zero_cores(const DevShape& s, const CorePtrsMut& d_cores, hipStream_t st);

// fast 3-core path (ttemb_fast3.hip)
bool fast3_supported(const DevShape& s);
bool fast3_pays(const DevShape& s, int64_t nnz);
bool fast3_prefix_in_chain(const DevShape& s, int64_t nnz, int64_t B);   // a whole forward of this size forms P inside its chain kernel
bool fast3_group_products_in_chain(const DevShape& s, int64_t nnz, int64_t B);   // a backward of this size forms the per-group products inside its chunk kernel
bool fast3_fits(const DevShape& s, int64_t nnz, int64_t B);   // the call fits one 32-bit row window / one grouping pass
bool fast3_fits_in_pieces(const DevShape& s, int64_t nnz, int64_t B);   // ... or runs as several pieces (needs `offsets`)
void fast3_set_piece_limits(int64_t rows, int64_t ids);                  // diagnostic: smaller pieces than the hardware's
void fast3_set_spin_limit(int64_t tries);                                // diagnostic: tries of the grouping pass's bounded waits
void fast3_set_wide_slab_min_ids(int64_t ids);                           // diagnostic: call size from which the wide backward reduces dG2 in LDS
// The pinned host word a device-side wait that ran out reports to (ttemb_api.hip): its device address for the kernels
// (null until ttemb_init() / ttemb_status() has created it -- the lookups allocate nothing), and the host-side check every
// lookup entry point starts with -- TTEMB_E_HIP once per reported fault.
uint32_t* fault_word(hipStream_t st);
int fault_word_init();
int pending_device_fault();
int64_t fast3_workspace_bytes(const DevShape& s, int32_t op, int64_t nnz, int64_t B);
int64_t fast3_plan_bytes(const DevShape& s, int64_t nnz);
// phase: 0 = whole forward, 1 = id-only half (grouping into `plan`), 2 = lookup on a plan grouped by phase 1.
// rowidx may be null when offsets is given (rows are derived while grouping); zero_rows: clear the output rows
// of bags that do not hold exactly one id (needs offsets)
int launch_forward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                         const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                         const int32_t* nnz_dev, int64_t B, float* output, bool zero_rows, void* ws,
                         int64_t ws_bytes, void* plan, int64_t plan_bytes, int phase, hipStream_t st, void* header);
// `header`: kFast3HeaderBytes at the start of the caller's workspace, the same address for every op on that workspace: the
// words of the grouping pass that outlive a call (its epoch, the pre-tagged range counters).  Any content is valid.
constexpr int64_t kFast3HeaderBytes = 40960;
// optimiser step folded into the last backward kernel (w == nullptr: none, gradients are written instead)
struct FusedUpdate {
  float* w[TTEMB_MAX_CORES];
  float* st[TTEMB_MAX_CORES];   // Adagrad state, or null for SGD
  float lr, eps;
  // A word of the workspace header the finalize kernel sets to 1 when the plan it ran on is POISONED and the host hears of
  // it (fault word present), else to 0 (sticky = 1: a later piece of the same call only ever sets it): the optimiser step
  // that follows a gradient-writing route (padded ranks, merged pairs, calls in pieces) reads it and leaves the parameters
  // alone -- what the fused finalize kernel does by itself.  Written by every grouped backward, so any content is valid.
  uint32_t* poison_out;
  int32_t sticky;
};
// where that word sits in the header: behind the epoch words and the banks of range counters (16 + 8 * 512 * 8 bytes)
constexpr int64_t kHeaderPoisonOffset = 16 + 8 * 512 * 8;
static_assert(kHeaderPoisonOffset + 8 <= kFast3HeaderBytes, "the poison word lies inside the header");
static_assert(kHeaderPoisonOffset == TTEMB_HEADER_POISON_OFFSET, "include/ttemb.h names the same offset");
int launch_backward_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                          const int64_t* rowidx, const int64_t* offsets, int64_t nnz, const int32_t* nnz_dev,
                          int64_t B, const float* d_output, const CorePtrsMut& d_cores, void* ws,
                          int64_t ws_bytes, const void* plan, int64_t plan_bytes, hipStream_t st,
                          const FusedUpdate* update, void* header);

// a window of a longer id list (one table of a table-batched call): bags [bag0, bag0 + B) of `offsets`, bounds read on the device
bool fast3_window_fits(const DevShape& s, int64_t nnz, int64_t bags_total, int64_t B);
int64_t fast3_window_workspace_bytes(const DevShape& s, bool bwd, int64_t nnz);
int launch_forward_window_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets,
                                int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, float* output, void* ws, int64_t ws_bytes,
                                hipStream_t st, void* header);
int launch_backward_window_fast3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets,
                                 int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B, const float* d_output,
                                 const CorePtrsMut& d_cores, void* ws, int64_t ws_bytes, hipStream_t st, const FusedUpdate* update,
                                 void* header);

// small batches of a 3-core table (ttemb_small3.inc): one wavefront per bag, MFMA per id, no grouping; `offsets` required.
// The backward ADDS into d_cores (zeroed by the caller) with float atomics.
bool small3_supported(const DevShape& s);   // (includes the run-time-shape kernels of ttemb_rt3.inc)
bool rt3_supported(const DevShape& s);
bool small3_templated_shape(const DevShape& s);   // a per-bag shape with an instantiated template
bool fast3_wide(const DevShape& s);               // a shape of the wide-rank grouped chain
bool small3_only(const DevShape& s);   // a rank-sweep shape only these kernels are instantiated for
int launch_forward_small3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets, int64_t nnz,
                          const int32_t* nnz_dev, int64_t B, float* output, hipStream_t st);
int launch_backward_small3(const DevShape& s, const CorePtrs& cores, const int64_t* indices, const int64_t* offsets, int64_t nnz,
                           const int32_t* nnz_dev, int64_t B, const float* d_output, const CorePtrsMut& d_cores, hipStream_t st);

// zero `bytes` (a multiple of 4) at `p` with a kernel on `st`.  Used instead of hipMemsetAsync everywhere: inside a
// captured HIP graph (ROCm 7.2) a memset node was seen to race the kernel node that follows it.
int launch_zero(void* p, size_t bytes, hipStream_t st, const char* what);

// ---------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------
// does the bag of position n hold exactly one id?  With `offsets` the answer comes from the bag
// lengths of the original id list; otherwise from the neighbouring rows of the live range.
__device__ __forceinline__ bool bag_is_single(const int64_t* __restrict__ rowidx,
                                              const int64_t* __restrict__ offsets, int64_t n, int64_t cnt,
                                              int64_t row) {
  if (offsets != nullptr) return offsets[row + 1] - offsets[row] == 1;
  return (n == 0 || rowidx[n - 1] != row) && (n + 1 >= cnt || rowidx[n + 1] != row);
}

// bag of position n: the last b with offsets[b] <= n  (tt_embeddings_cuda.cu:1349-1365 expands the same map)
__device__ __forceinline__ int64_t bag_of_position(const int64_t* __restrict__ offsets, int64_t B, int64_t n) {
  if (n < B) {   // the usual case: every bag holds one id
    const int64_t o0 = offsets[n], o1 = offsets[n + 1];
    if (o0 <= n && n < o1) return n;
  }
  int64_t lo = 0, hi = B;  // invariant: offsets[lo] <= n < offsets[hi]
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= n) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int64_t live_count(int64_t nnz, const int32_t* nnz_dev) {
  if (nnz_dev == nullptr) return nnz;
  int64_t c = *nnz_dev;
  return c < nnz ? (c < 0 ? 0 : c) : nnz;
}

// id -> per-core row numbers (reference: FBTT/tt_embeddings_cuda.cu:796-802).
__device__ __forceinline__ void split_index(const DevShape& s, int64_t idx, int (&i)[TTEMB_MAX_CORES]) {
  // an id outside the table is clamped as a whole (to the first / last row), the rule of every kernel family, so a bad
  // id gives the same row whichever kernels a batch size selects
  const int64_t rows = (int64_t)s.L[0] * s.p[0];
  idx = idx < 0 ? 0 : (idx >= rows ? rows - 1 : idx);
#pragma unroll
  for (int t = 0; t < TTEMB_MAX_CORES; ++t) {
    if (t < s.T) {
      int64_t d = idx / s.L[t];
      idx -= d * s.L[t];
      // ids beyond prod(p) would index past core 0; clamp so a bad id cannot fault
      i[t] = (int)(d < s.p[t] ? (d < 0 ? 0 : d) : s.p[t] - 1);
    } else {
      i[t] = 0;
    }
  }
}

}  // namespace ttemb
