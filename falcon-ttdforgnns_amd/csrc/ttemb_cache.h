// Launchers of ttemb_cache.hip (internal).
#pragma once
#include "ttemb_common.h"

namespace ttemb {

static inline int64_t align256(int64_t x) { return (x + 255) & ~int64_t(255); }

int launch_cache_update(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* freq,
                        int64_t H, hipStream_t st, bool one_sweep = false);
int64_t populate_workspace_bytes(int64_t H);
int launch_cache_populate_rank(int64_t* hashtbl, int64_t* freq, int32_t* state, int64_t H, int64_t C,
                               void* ws, int64_t ws_bytes, int64_t** sorted_keys_out, hipStream_t st);
int64_t preprocess_workspace_bytes(int64_t nnz);
int launch_rowidx(const int64_t* offsets, int64_t B, int64_t nnz, int64_t* rowidx, hipStream_t st);
int launch_set_count(int32_t* dst, int32_t v, hipStream_t st);
// freq != nullptr: the LFU update of the same ids (find-first form) rides in the probe pass (hashtbl is then written)
int launch_partition(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B,
                     int64_t* hashtbl, int64_t* freq, const int32_t* state, int64_t H, int64_t* indices_out,
                     int64_t* rowidx_out, int32_t* loc_out, int32_t* nnz_tt_dev, int32_t* dup_stamp,
                     void* ws, int64_t ws_bytes, hipStream_t st);
int launch_cache_forward(const int32_t* loc, const int64_t* rowidx, const int64_t* offsets, int64_t start,
                         const int32_t* start_dev, int64_t nnz, const float* weight, int64_t D,
                         float* out, hipStream_t st);
int launch_cache_scatter_add(const int32_t* loc, const int64_t* rowidx, int64_t start,
                             const int32_t* start_dev, int64_t nnz, const float* grad, int64_t D,
                             float scale, float* target, const int32_t* unique_dev, hipStream_t st);
int launch_cache_rowwise_adagrad(const int32_t* loc, const int64_t* rowidx, int64_t start,
                                 const int32_t* start_dev, int64_t nnz, const float* grad, int64_t D,
                                 float lr, float eps, float* state_sum, float* weight, hipStream_t st);

}  // namespace ttemb
