// LFU hash-table row cache + index preprocessing (HBM-bound integer / gather work).
//
// Reference counterparts: FBTT/hashtbl_cuda_utils.cuh (hash, probe),
// FBTT/tt_embeddings_cuda.cu:1083-1847 (cache kernels, preprocess).  Geometry is
// re-derived for 64-lane wavefronts: one wavefront owns one id and its D-float row
// (float4 per lane), instead of the reference's 32-lane "warp per id" blocks.
#include "ttemb_common.h"
#include "ttemb_cache.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace ttemb {

// ---------------------------------------------------------------------------------
// hash (bit-exact with FBTT/hashtbl_cuda_utils.cuh:48-76 so that saved
// (hashtbl, cache_state, cache_weight) triples stay consistent across backends)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__device__ __forceinline__ uint32_t hash_slot(int64_t key, uint32_t H) {
  const uint64_t u = (uint64_t)key;
  uint32_t h = 0;
  uint32_t k = (uint32_t)u;
  k *= 0xcc9e2d51u;
  k = rotl32(k, 15);
  k *= 0x1b873593u;
  h ^= k;
  h = rotl32(h, 13);
  h = h * 5u + 0xe6546b64u;
  k = (uint32_t)(u >> 32);
  k *= 0xcc9e2d51u;
  k = rotl32(k, 15);
  k *= 0x1b873593u;
  h ^= k;
  h = rotl32(h, 13);
  h = h * 5u + 0xe6546b64u;
  h ^= 2u;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return (uint32_t)(((uint64_t)h * (uint64_t)H) >> 32);
}

// slot of `key`, or -1.  Like the reference it does not stop at empty slots
// (hashtbl_cuda_utils.cuh:135-154), so evictions never hide a displaced key.
__device__ __forceinline__ int32_t table_find(int64_t key, const int64_t* __restrict__ keys, uint32_t H) {
  if (key == kEmptyKey) return -1;
  uint32_t s = hash_slot(key, H);
#pragma unroll
  for (int probe = 0; probe < kMaxProbes; ++probe) {
    if (keys[s] == key) return (int32_t)s;
    s = (s + 1 == H) ? 0 : s + 1;
  }
  return -1;
}

// LFU update of one key: count it in the slot that holds it, else insert it into the first of its probe slots that is
// empty.  Returns the slot that took the count, or -1 (three occupied probes: the id is simply not tracked, as in the
// reference).
// Pass 1, reads only: is the key already in one of its probe slots?  The reference probes and
// inserts in one sweep (hashtbl_cuda_utils.cuh:102-133), which after cache_populate's evictions
// re-inserts a cached id into a hole in front of its own slot; the id then resolves to the new slot
// (cache_state -1) and silently leaves the cache, for a thread-order-dependent set of ids.  Finding
// first keeps every tracked key in one slot; before any eviction the two are the same table.
// (A slot never changes once it holds a real key until populate evicts, so what pass 1 saw occupied
// stays occupied: pass 2 only needs a CAS on the slots it saw empty -- one atomic per id.)
__device__ __forceinline__ int32_t lfu_count(int64_t key, int64_t* __restrict__ keys, int64_t* __restrict__ freq, uint32_t H) {
  const uint32_t s0 = hash_slot(key, H);
  unsigned long long seen[kMaxProbes];
  uint32_t s = s0;
#pragma unroll
  for (int probe = 0; probe < kMaxProbes; ++probe) {
    seen[probe] = (unsigned long long)__builtin_nontemporal_load(&keys[s]);
    if (seen[probe] == (unsigned long long)key) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&freq[s]), 1ull);
      return (int32_t)s;
    }
    s = (s + 1 == H) ? 0 : s + 1;
  }
  s = s0;
#pragma unroll
  for (int probe = 0; probe < kMaxProbes; ++probe) {
    if (seen[probe] == (unsigned long long)kEmptyKey) {
      const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[s]),
                                               (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (old == (unsigned long long)kEmptyKey || old == (unsigned long long)key) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&freq[s]), 1ull);
        return (int32_t)s;
      }
    }
    s = (s + 1 == H) ? 0 : s + 1;
  }
  return -1;
}

__global__ void cache_update_kernel(const int64_t* __restrict__ indices, int64_t nnz,
                                    int64_t* __restrict__ keys, int64_t* __restrict__ freq,
                                    uint32_t H) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnz) return;
  (void)lfu_count(indices[n], keys, freq, H);
}

// The reference's own insert, for callers that need its table bit for bit: probe and insert in ONE sweep
// (hashtbl_cuda_utils.cuh:102-133) -- the first probe slot that is empty or already holds the key takes the count.
// After cache_populate's evictions this re-inserts a cached id into a hole in front of its slot (see above).
__global__ void cache_update_one_sweep_kernel(const int64_t* __restrict__ indices, int64_t nnz, int64_t* __restrict__ keys,
                                              int64_t* __restrict__ freq, uint32_t H) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnz) return;
  const int64_t key = indices[n];
  uint32_t s = hash_slot(key, H);
#pragma unroll
  for (int probe = 0; probe < kMaxProbes; ++probe) {
    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[s]), (unsigned long long)kEmptyKey,
                                             (unsigned long long)key);
    if (old == (unsigned long long)kEmptyKey || old == (unsigned long long)key) {
      atomicAdd(reinterpret_cast<unsigned long long*>(&freq[s]), 1ull);
      return;
    }
    s = (s + 1 == H) ? 0 : s + 1;
  }
}

int launch_cache_update(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* freq,
                        int64_t H, hipStream_t st, bool one_sweep) {
  if (nnz <= 0) return TTEMB_OK;
  const int threads = 256;
  const int64_t blocks = (nnz + threads - 1) / threads;
  if (one_sweep)
    hipLaunchKernelGGL(cache_update_one_sweep_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, indices, nnz, hashtbl, freq,
                       (uint32_t)H);
  else
    hipLaunchKernelGGL(cache_update_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, indices,
                       nnz, hashtbl, freq, (uint32_t)H);
  return check_hip(hipGetLastError(), "cache_update_kernel");
}

// ---------------------------------------------------------------------------------
// cache_populate
// ---------------------------------------------------------------------------------
__global__ void mark_popular_kernel(int64_t H, int64_t C, int64_t* __restrict__ sorted_keys,
                                    int64_t* __restrict__ keys, int64_t* __restrict__ freq,
                                    int32_t* __restrict__ state) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= H) return;
  const int64_t key = sorted_keys[n];
  if (key != kEmptyKey) {
    const int32_t slot = table_find(key, keys, (uint32_t)H);
    if (slot < 0) return;  // cannot happen for a key read out of the table itself
    if (n < C) {
      state[slot] = (int32_t)n;
    } else {
      keys[slot] = kEmptyKey;
      freq[slot] = 0;
    }
  } else if (n < C) {
    sorted_keys[n] = 0;  // unused rank: its cache row holds the row of id 0 (reference :1144-1147)
  }
}

int64_t populate_workspace_bytes(int64_t H) {
  size_t tmp = 0;
  int64_t* nul = nullptr;
  hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, tmp, nul, nul, nul, nul, (size_t)H, 0, 64,
                                                (hipStream_t)0, false);
  if (e != hipSuccess) return -1;
  return align256(H * 8) * 2 + align256((int64_t)tmp) + 256;
}

int launch_cache_populate_rank(int64_t* hashtbl, int64_t* freq, int32_t* state, int64_t H, int64_t C,
                               void* ws, int64_t ws_bytes, int64_t** sorted_keys_out, hipStream_t st) {
  char* base = reinterpret_cast<char*>(ws);
  int64_t* sorted_freq = reinterpret_cast<int64_t*>(base);
  int64_t* sorted_keys = reinterpret_cast<int64_t*>(base + align256(H * 8));
  char* tmp = base + 2 * align256(H * 8);
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs_desc(nullptr, tmp_bytes, freq, sorted_freq, hashtbl,
                                                sorted_keys, (size_t)H, 0, 64, st, false);
  if (e != hipSuccess) return check_hip(e, "radix_sort_pairs_desc(size)");
  if (2 * align256(H * 8) + (int64_t)tmp_bytes > ws_bytes)
    return fail(TTEMB_E_WORKSPACE, "cache_populate needs %lld workspace bytes, got %lld",
                (long long)(2 * align256(H * 8) + tmp_bytes), (long long)ws_bytes);
  // stable: equal frequencies keep slot order, like cub::DeviceRadixSort (reference :1292-1318)
  e = rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, freq, sorted_freq, hashtbl, sorted_keys,
                                     (size_t)H, 0, 64, st, false);
  if (e != hipSuccess) return check_hip(e, "radix_sort_pairs_desc");
  const int threads = 256;
  const int64_t blocks = (H + threads - 1) / threads;
  hipLaunchKernelGGL(mark_popular_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, H, C,
                     sorted_keys, hashtbl, freq, state);
  *sorted_keys_out = sorted_keys;
  return check_hip(hipGetLastError(), "mark_popular_kernel");
}

// ---------------------------------------------------------------------------------
// preprocess: rowidx expansion, cache lookup, stable partition
// ---------------------------------------------------------------------------------
__global__ void rowidx_kernel(const int64_t* __restrict__ offsets, int64_t B, int64_t nnz,
                              int64_t* __restrict__ rowidx) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int64_t lo = offsets[b];
  int64_t hi = offsets[b + 1];
  lo = lo < 0 ? 0 : lo;
  hi = hi > nnz ? nnz : hi;
  for (int64_t l = lo; l < hi; ++l) rowidx[l] = b;
}

// Stable partition in two launches: the lookup counts the TT (uncached) ids of every 256-id block; the scatter
// kernel adds up the counts of the blocks before its own (a few hundred coalesced loads), ranks its ids with a
// ballot, finds their bags and writes the three partitioned arrays.  (rowidx expansion + lookup + a rocPRIM scan
// -- two launches -- + scatter were five launches and 36 us for 409 600 ids.)
constexpr int kPartThreads = 256;

// UPDATE: the LFU update of the same ids (ttemb_cache_update, the find-first form) rides in the probe pass -- the class
// calls update_cache_state and preprocess_indices_sync back to back on the same ids (tt_embeddings_ops.py:836-870), and
// both visit the same <= 3 slots.  Same table and same locations as the two passes in sequence: the find-first update
// never moves or evicts a key, so a key resolves to the same slot before, during and after the update of the batch
// (an id inserted by this very pass sits in a slot whose state is whatever cache_populate left there, which is what
// the lookup after a separate update would read, too).
// Duplicate detection among the cached ids (the cache backward may then update rows without float atomics): every cached id
// stores its POSITION into the stamp of its cache row; the scatter kernel of the same call reads the stamp back and a
// position that lost its row to another one has met a duplicate.  No returning atomic, no epoch, no initial state.
template <bool UPDATE>
__global__ __launch_bounds__(kPartThreads) void cache_lookup_kernel(const int64_t* __restrict__ indices, int64_t nnz,
                                                                    int64_t* __restrict__ keys, int64_t* __restrict__ freq,
                                                                    const int32_t* __restrict__ state, uint32_t H,
                                                                    int32_t* __restrict__ loc,
                                                                    int32_t* __restrict__ blockcnt,
                                                                    int32_t* __restrict__ dup_stamp, int32_t* __restrict__ nnz_tt) {
  const int64_t n = (int64_t)blockIdx.x * kPartThreads + threadIdx.x;
  int32_t where = -1;
  if (n < nnz) {
    const int64_t key = indices[n];
    int32_t slot;
    if constexpr (UPDATE) {
      slot = lfu_count(key, keys, freq, H);
      if (key == kEmptyKey) slot = -1;   // (table_find's rule: the empty key is never "found")
    } else {
      slot = table_find(key, keys, H);
    }
    if (slot >= 0) where = state[slot];
    loc[n] = where;   // < 0: not cached, the id goes through the TT chain
    if (dup_stamp != nullptr && where >= 0) dup_stamp[where] = (int32_t)n;
  }
  if (n == 0 && dup_stamp != nullptr) nnz_tt[1] = 0;   // "no cache row met twice" until the scatter kernel finds one
  const int c = __syncthreads_count(n < nnz && where < 0);
  if (threadIdx.x == 0) blockcnt[blockIdx.x] = c;
}

// selected (TT) items keep input order at the front; rejected (cached) items fill the
// tail from the end backwards -- the order cub::DevicePartition::Flagged produces and
// the reference's cache kernels therefore see (tt_embeddings_cuda.cu:1448-1490).
__global__ __launch_bounds__(kPartThreads) void partition_scatter_kernel(int64_t nnz, int64_t B,
                                                                         const int64_t* __restrict__ offsets,
                                                                         const int32_t* __restrict__ blockcnt,
                                                                         const int64_t* __restrict__ indices,
                                                                         const int32_t* __restrict__ loc,
                                                                         int64_t* __restrict__ indices_out,
                                                                         int64_t* __restrict__ rowidx_out,
                                                                         int32_t* __restrict__ loc_out,
                                                                         int32_t* __restrict__ nnz_tt,
                                                                         const int32_t* __restrict__ dup_stamp) {
  __shared__ int wave_cnt[kPartThreads / kWave];
  __shared__ int64_t before_block;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * kPartThreads + threadIdx.x;
  int64_t id = 0, row = 0;
  int32_t where = 0;
  if (n < nnz) {   // independent of the counts: issued first
    id = indices[n];
    where = loc[n];
    row = bag_of_position(offsets, B, n);
  }
  // a cached id whose row's stamp is not its own position shares the row with another id of this call
  const bool dup = dup_stamp != nullptr && n < nnz && where >= 0 && dup_stamp[where] != (int32_t)n;
  const bool last = blockIdx.x == gridDim.x - 1;
  if (wave == 0) {
    int64_t v = 0;
    const int64_t nb = (int64_t)blockIdx.x;
    for (int64_t b0 = 0; b0 < nb; b0 += 8 * kWave) {   // eight loads per lane in flight, then the sums
      int32_t c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t b = b0 + k * kWave + lane;
        c[k] = b < nb ? blockcnt[b] : 0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v += c[k];
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
      const uint32_t lo_w = __shfl_down((uint32_t)v, d, kWave), hi_w = __shfl_down((uint32_t)((uint64_t)v >> 32), d, kWave);
      v += (int64_t)((uint64_t)lo_w | ((uint64_t)hi_w << 32));
    }
    if (lane == 0) before_block = v;
  }
  const bool f = n < nnz && where < 0;
  const unsigned long long ball = __ballot(f);
  if (lane == 0) wave_cnt[wave] = __popcll(ball);
  __syncthreads();
  int64_t before = before_block + __popcll(ball & ((1ull << lane) - 1ull));
  int total = 0;
  for (int w = 0; w < kPartThreads / kWave; ++w) {
    if (w < wave) before += wave_cnt[w];
    total += wave_cnt[w];
  }
  if (last && threadIdx.x == 0) nnz_tt[0] = (int32_t)(before_block + total);
  if (dup) nnz_tt[1] = 1;   // (cleared by the lookup kernel; every writer stores the same word)
  if (n >= nnz) return;
  const int64_t dst = f ? before : (nnz - 1 - (n - before));
  indices_out[dst] = id;
  rowidx_out[dst] = row;
  loc_out[dst] = where;
}

__global__ void set_count_kernel(int32_t* dst, int32_t v) { *dst = v; }

static int64_t part_blocks(int64_t nnz) { return (nnz + kPartThreads - 1) / kPartThreads; }

int64_t preprocess_workspace_bytes(int64_t nnz) {
  if (nnz < 0) nnz = 0;
  return align256(nnz * 4) + align256(part_blocks(nnz > 0 ? nnz : 1) * 4) + 256;   // cache locations, per-block counts
}

int launch_rowidx(const int64_t* offsets, int64_t B, int64_t nnz, int64_t* rowidx, hipStream_t st) {
  if (B <= 0 || nnz <= 0) return TTEMB_OK;
  const int threads = 256;
  const int64_t blocks = (B + threads - 1) / threads;
  hipLaunchKernelGGL(rowidx_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, offsets, B, nnz, rowidx);
  return check_hip(hipGetLastError(), "rowidx_kernel");
}

int launch_set_count(int32_t* dst, int32_t v, hipStream_t st) {
  hipLaunchKernelGGL(set_count_kernel, dim3(1), dim3(1), 0, st, dst, v);
  return check_hip(hipGetLastError(), "set_count_kernel");
}

int launch_partition(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B,
                     int64_t* hashtbl, int64_t* freq, const int32_t* state, int64_t H, int64_t* indices_out,
                     int64_t* rowidx_out, int32_t* loc_out, int32_t* nnz_tt_dev, int32_t* dup_stamp,
                     void* ws, int64_t ws_bytes, hipStream_t st) {
  const int64_t need = preprocess_workspace_bytes(nnz);
  if (need > ws_bytes)
    return fail(TTEMB_E_WORKSPACE, "preprocess needs %lld workspace bytes, got %lld", (long long)need,
                (long long)ws_bytes);
  char* base = reinterpret_cast<char*>(ws);
  int32_t* loc = reinterpret_cast<int32_t*>(base);
  base += align256(nnz * 4);
  int32_t* blockcnt = reinterpret_cast<int32_t*>(base);
  const int64_t blocks = part_blocks(nnz);
  profile_begin(4, st);
  if (freq != nullptr)   // the LFU update of the same ids rides in the probe pass
    hipLaunchKernelGGL(cache_lookup_kernel<true>, dim3((unsigned)blocks), dim3(kPartThreads), 0, st, indices, nnz,
                       hashtbl, freq, state, (uint32_t)H, loc, blockcnt, dup_stamp, nnz_tt_dev);
  else
    hipLaunchKernelGGL(cache_lookup_kernel<false>, dim3((unsigned)blocks), dim3(kPartThreads), 0, st, indices, nnz,
                       hashtbl, freq, state, (uint32_t)H, loc, blockcnt, dup_stamp, nnz_tt_dev);
  profile_end(4, st);
  int rc = check_hip(hipGetLastError(), "cache_lookup_kernel");
  if (rc) return rc;
  profile_begin(5, st);
  hipLaunchKernelGGL(partition_scatter_kernel, dim3((unsigned)blocks), dim3(kPartThreads), 0, st, nnz, B, offsets,
                     blockcnt, indices, loc, indices_out, rowidx_out, loc_out, nnz_tt_dev, dup_stamp);
  profile_end(5, st);
  return check_hip(hipGetLastError(), "partition_scatter_kernel");
}

// ---------------------------------------------------------------------------------
// cached-row forward / backward: one wavefront per id, float4 per lane.
// A bag with one cached id is updated with plain loads/stores, otherwise atomics.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int64_t live_start(int64_t start, const int32_t* start_dev, int64_t nnz) {
  if (start_dev != nullptr) start = *start_dev;
  if (start < 0) start = 0;
  return start > nnz ? nnz : start;
}

// Forward: lanes are tied to ids four by four (lane = 4 * id + piece, 16 ids per wavefront step): every lane loads
// the 16-byte pieces j, j + 4, ... of "its" cached row and stores them into its output row, so the index loads, the
// bag test and up to eight row pieces per lane are all in flight together (one wavefront per id with D/4 of 64
// lanes busy ran at 2 TB/s: a chain of four dependent loads per id).
// single-id bags: plain store when the caller vouches (offsets) that no TT id shares the row, read-modify-write when
// only the cached part is known to be alone in it; else float atomics.
#ifndef TTEMB_CACHE_IDS
#define TTEMB_CACHE_IDS 32
#endif
constexpr int kIdsPerWave = TTEMB_CACHE_IDS;   // steps of 16 ids

__global__ __launch_bounds__(256) void cache_forward_kernel(const int32_t* __restrict__ loc,
                                                            const int64_t* __restrict__ rowidx,
                                                            const int64_t* __restrict__ offsets,
                                                            int64_t start, const int32_t* start_dev,
                                                            int64_t nnz,
                                                            const float* __restrict__ weight, int D,
                                                            float* __restrict__ out) {
  const int64_t s0 = live_start(start, start_dev, nnz);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b_l = lane >> 2, j_l = lane & 3;
  const int64_t first = s0 + ((int64_t)blockIdx.x * 4 + wave) * kIdsPerWave;
  if (first >= nnz) return;
  const int D4 = D >> 2;
  int64_t row[kIdsPerWave / 16];
  int32_t l[kIdsPerWave / 16];
  bool on[kIdsPerWave / 16];
#pragma unroll
  for (int c = 0; c < kIdsPerWave / 16; ++c) {
    const int64_t n = first + 16 * c + b_l;
    on[c] = n < nnz;
    row[c] = on[c] ? rowidx[n] : 0;
    l[c] = on[c] ? loc[n] : 0;
  }
#pragma unroll
  for (int c = 0; c < kIdsPerWave / 16; ++c) {
    const int64_t n = first + 16 * c + b_l;
    bool alone = false, store_only = false;
    if (on[c]) {
      if (offsets != nullptr) {
        alone = offsets[row[c] + 1] - offsets[row[c]] == 1;
        store_only = alone;
      } else {
        alone = (n == s0 || rowidx[n - 1] != row[c]) && (n + 1 >= nnz || rowidx[n + 1] != row[c]);
      }
    }
    const float4* w = reinterpret_cast<const float4*>(weight + (int64_t)l[c] * D);
    float* o = out + row[c] * D;
    for (int base = 0; base < D4; base += 32) {   // eight pieces per lane at a time
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pc = base + 4 * k + j_l;
        v[k] = (on[c] && pc < D4) ? w[pc] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pc = base + 4 * k + j_l;
        if (!on[c] || pc >= D4) continue;
        if (store_only) {
          reinterpret_cast<float4*>(o)[pc] = v[k];
        } else if (alone) {
          float4 cur = reinterpret_cast<float4*>(o)[pc];
          cur.x += v[k].x; cur.y += v[k].y; cur.z += v[k].z; cur.w += v[k].w;
          reinterpret_cast<float4*>(o)[pc] = cur;
        } else {
          atomicAdd(&o[4 * pc + 0], v[k].x);
          atomicAdd(&o[4 * pc + 1], v[k].y);
          atomicAdd(&o[4 * pc + 2], v[k].z);
          atomicAdd(&o[4 * pc + 3], v[k].w);
        }
      }
    }
  }
}

// The same copy as a STREAM (bag boundaries given, D <= 128): a fixed number of wavefronts per SIMD, each taking steps of 16
// ids with a stride; the (location, bag) pair of the next step is requested before a step's rows are stored.
// One wavefront per 32 ids and all of them resident at once (the kernel above) runs the chip in lock step -- every
// wavefront reads its indices, then every wavefront gathers, then every wavefront stores: 133 MB in 42 us although either
// stream alone runs at 4-5 TB/s.  Here wavefronts drift apart and the gathers of one overlap the stores of another.
#ifndef TTEMB_CACHE_WPS
#define TTEMB_CACHE_WPS 4
#endif
constexpr int kStreamWavesPerSimd = TTEMB_CACHE_WPS;

template <int NP>
struct RowSet {
  float4 v[NP];
  int64_t row, o0, o1;
  bool on;
};

template <int NP>   // NP = 16-byte pieces of a row per lane (four lanes per row): D <= 16 NP
__global__ __launch_bounds__(256) void cache_forward_stream_kernel(const int32_t* __restrict__ loc, const int64_t* __restrict__ rowidx,
                                                                   const int64_t* __restrict__ offsets, int64_t start,
                                                                   const int32_t* start_dev, int64_t nnz,
                                                                   const float* __restrict__ weight, int D, float* __restrict__ out) {
  const int64_t s0 = live_start(start, start_dev, nnz);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b_l = lane >> 2, j_l = lane & 3;
  const int D4 = D >> 2;
  const int64_t steps = (nnz - s0 + 15) >> 4, stride = (int64_t)gridDim.x * 4;
  int64_t step = (int64_t)blockIdx.x * 4 + wave;
  if (step >= steps) return;
  struct Idx {
    int32_t l;
    int64_t row;
    bool on;
  };
  auto idx_of = [&](int64_t st, Idx& i) {   // past the end: the last id's pair (a valid address), nothing of it is stored
    const int64_t n = s0 + 16 * st + b_l, last = nnz - 1;
    i.on = n <= last;   // (st >= steps implies n > last)
    const int64_t nc = n < last ? n : last;
    i.l = loc[nc];
    i.row = rowidx[nc];
  };
  auto request = [&](RowSet<NP>& r, const Idx& i) {
    r.row = i.row;
    r.on = i.on;
    r.o0 = offsets[i.row];
    r.o1 = offsets[i.row + 1];
    // one address per row: every piece but a lane's last exists (NP = ceil(D4 / 4)) and rides in the immediate offset
    const float4* w = reinterpret_cast<const float4*>(weight + (int64_t)i.l * D) + j_l;
#pragma unroll
    for (int k = 0; k + 1 < NP; ++k) r.v[k] = w[4 * k];
    r.v[NP - 1] = w[4 * (NP - 1) + j_l < D4 ? 4 * (NP - 1) : D4 - 1 - j_l];
  };
  auto emit = [&](const RowSet<NP>& r) {
    const bool single = r.on && r.o1 - r.o0 == 1;   // the caller's offsets vouch that no TT id shares the row: plain stores
    float* o = out + r.row * D;
    if (single) {   // (every piece but a lane's last exists: NP = ceil(D4 / 4))
#pragma unroll
      for (int k = 0; k + 1 < NP; ++k) reinterpret_cast<float4*>(o)[4 * k + j_l] = r.v[k];
      if (4 * (NP - 1) + j_l < D4) reinterpret_cast<float4*>(o)[4 * (NP - 1) + j_l] = r.v[NP - 1];
    }
    if (__ballot(r.on && !single) != 0ull) {   // bags of several ids accumulate
      if (r.on && !single) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int pc = 4 * k + j_l;
          if (pc < D4) {
            atomicAdd(&o[4 * pc + 0], r.v[k].x);
            atomicAdd(&o[4 * pc + 1], r.v[k].y);
            atomicAdd(&o[4 * pc + 2], r.v[k].z);
            atomicAdd(&o[4 * pc + 3], r.v[k].w);
          }
        }
      }
    }
  };
  // per step: request the rows | request the pair of the next step | store (waits for the rows, not for the pair).  One
  // row set: with two the compiler's register sharing made every request wait for the one before anyway, and the
  // registers are better spent on more wavefronts; the stores of a step are still in flight when the next gather starts.
  RowSet<NP> r;
  Idx ia, ib;
  idx_of(step, ia);
  for (;;) {
    request(r, ia);
    __builtin_amdgcn_sched_barrier(0);
    idx_of(step + stride, ib);
    __builtin_amdgcn_sched_barrier(0);
    emit(r);
    __builtin_amdgcn_sched_barrier(0);
    step += stride;
    if (step >= steps) break;
    request(r, ib);
    __builtin_amdgcn_sched_barrier(0);
    idx_of(step + stride, ia);
    __builtin_amdgcn_sched_barrier(0);
    emit(r);
    __builtin_amdgcn_sched_barrier(0);
    step += stride;
    if (step >= steps) break;
  }
}

// scale == -lr : cache_backward_sgd ; scale == 1 : cache_backward_dense (target pre-zeroed).
// Duplicate ids may hit one cache row, so the adds are float atomics -- issued as whole rows of
// consecutive floats (64 lanes = 256 contiguous bytes per instruction, the full-rate shape).  A wavefront takes
// kScatterIds ids: their (row, location) pairs are loaded by one lane each, then every gradient piece of the step is
// requested before the first atomic is issued.
constexpr int kScatterIds = 8;

template <int NP>   // unique form: 16-byte pieces of a row per lane and pass (four lanes per row)
__global__ __launch_bounds__(256) void cache_scatter_add_kernel(const int32_t* __restrict__ loc,
                                                                const int64_t* __restrict__ rowidx,
                                                                int64_t start, const int32_t* start_dev,
                                                                int64_t nnz,
                                                                const float* __restrict__ grad, int D,
                                                                float scale, float* __restrict__ target,
                                                                const int32_t* __restrict__ unique_dev, int stream_blocks) {
  const int64_t s0 = live_start(start, start_dev, nnz);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (unique_dev != nullptr && *unique_dev == 0) {
    // no cache row occurs twice in this call (ttemb_preprocess checked): every row has one writer, so the update is a
    // plain read-modify-write, four lanes per row.  A stream like cache_forward_stream_kernel's: `stream_blocks`
    // workgroups take steps of 16 ids with a stride; the rest of the grid (sized for the atomic form below) leaves.
    if ((int)blockIdx.x >= stream_blocks) return;
    const int b_l = lane >> 2, j_l = lane & 3;
    const int D4 = D >> 2;
    const int64_t steps = (nnz - s0 + 15) >> 4, stride = (int64_t)stream_blocks * 4;
    int64_t step = (int64_t)blockIdx.x * 4 + wave;
    if (step >= steps) return;
    // (NP pieces per lane and pass: D <= 16 NP in one pass, wider rows in several)
    struct Pair {
      float4 g[NP], t[NP];
      float4* tp;
      bool on;
    };
    struct Idx {
      int32_t l;
      int64_t row;
      bool on;
    };
    auto idx_of = [&](int64_t st, Idx& i) {
      const int64_t n = s0 + 16 * st + b_l, last = nnz - 1;
      i.on = n <= last;   // (st >= steps implies n > last)
      const int64_t nc = n < last ? n : last;
      i.l = loc[nc];
      i.row = rowidx[nc];
    };
    for (int base = 0; base < D4; base += 4 * NP) {
      auto request = [&](Pair& r, const Idx& i) {
        // one address per row and table: pieces that exist ride in the immediate offset, the others re-read the row's last
        const float4* g = reinterpret_cast<const float4*>(grad + i.row * D) + base + j_l;
        r.tp = reinterpret_cast<float4*>(target + (int64_t)i.l * D);
        const float4* t = r.tp + base + j_l;
        r.on = i.on;
        const int left = D4 - 1 - base - j_l;   // the last piece of the row, counted from this lane's first
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          if (base + 4 * k + 3 < D4) {   // (every lane has it: uniform)
            r.g[k] = g[4 * k];
            r.t[k] = t[4 * k];
          } else {
            r.g[k] = g[4 * k <= left ? 4 * k : left];
            r.t[k] = t[4 * k <= left ? 4 * k : left];
          }
        }
      };
      auto emit = [&](Pair& r) {
        if (r.on) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            const int pc = base + 4 * k + j_l;
            if (pc < D4) {
              float4 x = r.t[k];
              x.x += r.g[k].x * scale; x.y += r.g[k].y * scale; x.z += r.g[k].z * scale; x.w += r.g[k].w * scale;
              r.tp[pc] = x;
            }
          }
        }
      };
      Pair r;
      Idx ia, ib;
      int64_t st = step;
      idx_of(st, ia);
      for (;;) {
        request(r, ia);
        __builtin_amdgcn_sched_barrier(0);
        idx_of(st + stride, ib);
        __builtin_amdgcn_sched_barrier(0);
        emit(r);
        __builtin_amdgcn_sched_barrier(0);
        st += stride;
        if (st >= steps) break;
        request(r, ib);
        __builtin_amdgcn_sched_barrier(0);
        idx_of(st + stride, ia);
        __builtin_amdgcn_sched_barrier(0);
        emit(r);
        __builtin_amdgcn_sched_barrier(0);
        st += stride;
        if (st >= steps) break;
      }
    }
    return;
  }
  const int64_t first = s0 + ((int64_t)blockIdx.x * 4 + wave) * kScatterIds;
  if (first >= nnz) return;
  const int64_t mine = first + lane;
  const bool have = lane < kScatterIds && mine < nnz;
  const int64_t my_row = have ? rowidx[mine] : 0;
  const int32_t my_loc = have ? loc[mine] : 0;
  const int n_here = (int)(nnz - first < kScatterIds ? nnz - first : kScatterIds);
  for (int base = 0; base < D; base += 2 * kWave) {   // two 64-float pieces per id at a time
    float v[kScatterIds][2];
#pragma unroll
    for (int k = 0; k < kScatterIds; ++k) {
      const int64_t r = __shfl(my_row, k, kWave);
      const float* g = grad + r * D;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = base + h * kWave + lane;
        v[k][h] = (k < n_here && e < D) ? g[e] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < kScatterIds; ++k) {
      const int32_t lk = __shfl(my_loc, k, kWave);
      float* t = target + (int64_t)lk * D;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = base + h * kWave + lane;
        if (k < n_here && e < D) atomicAdd(&t[e], v[k][h] * scale);
      }
    }
  }
}

__global__ __launch_bounds__(256) void cache_rowwise_adagrad_kernel(
    const int32_t* __restrict__ loc, const int64_t* __restrict__ rowidx, int64_t start,
    const int32_t* start_dev, int64_t nnz, const float* __restrict__ grad, int D, float lr, float eps,
    float* __restrict__ state_sum, float* __restrict__ weight) {
  const int64_t s0 = live_start(start, start_dev, nnz);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = s0 + (int64_t)blockIdx.x * 4 + wave;
  if (n >= nnz) return;
  const float4* g = reinterpret_cast<const float4*>(grad + rowidx[n] * D);
  float sq = 0.f;
  for (int c = lane; c * 4 < D; c += kWave) {
    const float4 v = g[c];
    sq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) sq += __shfl_xor(sq, m, kWave);  // all 64 lanes, not 32
  const float g2 = sq / (float)D;
  const int32_t l = loc[n];
  float mult = 0.f;
  if (lane == 0) {
    const float old = atomicAdd(&state_sum[l], g2);
    mult = lr * (1.0f / (sqrtf(old + g2) + eps));
  }
  mult = __shfl(mult, 0, kWave);
  // A cache row can occur several times in a batch (the reference's read-modify-write loses updates then,
  // tt_embeddings_cuda.cu:1746-1806): the row is updated with float atomics, 64 consecutive floats per instruction.
  float* w = weight + (int64_t)l * D;
  const float* gs = grad + rowidx[n] * D;
  for (int c = lane; c < D; c += kWave) atomicAdd(w + c, -gs[c] * mult);
}

static inline unsigned wave_blocks(int64_t nnz) { return (unsigned)((nnz + 3) / 4); }
static inline unsigned multi_blocks(int64_t nnz) { return (unsigned)((nnz + 4 * kIdsPerWave - 1) / (4 * kIdsPerWave)); }
static inline unsigned scatter_blocks(int64_t nnz) { return (unsigned)((nnz + 4 * kScatterIds - 1) / (4 * kScatterIds)); }

static unsigned stream_blocks_for(int64_t span) {   // four wavefronts per workgroup, kStreamWavesPerSimd per SIMD, no more than steps
  const int64_t steps = (span + 15) / 16;
  int64_t blocks = (int64_t)device_cus() * kStreamWavesPerSimd;
  if (blocks * 4 > steps) blocks = (steps + 3) / 4;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

int launch_cache_forward(const int32_t* loc, const int64_t* rowidx, const int64_t* offsets, int64_t start,
                         const int32_t* start_dev, int64_t nnz, const float* weight, int64_t D,
                         float* out, hipStream_t st) {
  const int64_t span = start_dev ? nnz : nnz - start;
  if (span <= 0) return TTEMB_OK;
  if (offsets != nullptr && D <= 128) {   // the pipelined copy (the cached range may turn out shorter than `span`: fewer steps)
    const dim3 grid(stream_blocks_for(span));
    const int np = (int)((D / 4 + 3) / 4);   // pieces per lane
    profile_begin(6, st);
#define TTEMB_FWD_STREAM(NP)                                                                                                   \
  case NP:                                                                                                                     \
    hipLaunchKernelGGL(cache_forward_stream_kernel<NP>, grid, dim3(256), 0, st, loc, rowidx, offsets, start, start_dev, nnz, \
                       weight, (int)D, out);                                                                                   \
    break;
    switch (np) {
      TTEMB_FWD_STREAM(1)
      TTEMB_FWD_STREAM(2)
      TTEMB_FWD_STREAM(3)
      TTEMB_FWD_STREAM(4)
      TTEMB_FWD_STREAM(5)
      TTEMB_FWD_STREAM(6)
      TTEMB_FWD_STREAM(7)
      default:
      TTEMB_FWD_STREAM(8)
    }
#undef TTEMB_FWD_STREAM
    profile_end(6, st);
    return check_hip(hipGetLastError(), "cache_forward_stream_kernel");
  }
  hipLaunchKernelGGL(cache_forward_kernel, dim3(multi_blocks(span)), dim3(256), 0, st, loc, rowidx, offsets,
                     start, start_dev, nnz, weight, (int)D, out);
  return check_hip(hipGetLastError(), "cache_forward_kernel");
}

int launch_cache_scatter_add(const int32_t* loc, const int64_t* rowidx, int64_t start,
                             const int32_t* start_dev, int64_t nnz, const float* grad, int64_t D,
                             float scale, float* target, const int32_t* unique_dev, hipStream_t st) {
  const int64_t span = start_dev ? nnz : nnz - start;
  if (span <= 0) return TTEMB_OK;
  unsigned blocks = scatter_blocks(span), sb = stream_blocks_for(span);
  if (sb > blocks) blocks = sb;
  const int np = D >= 128 ? 8 : (int)((D / 4 + 3) / 4);
  profile_begin(7, st);
#define TTEMB_SCATTER(NP)                                                                                              \
  case NP:                                                                                                             \
    hipLaunchKernelGGL(cache_scatter_add_kernel<NP>, dim3(blocks), dim3(256), 0, st, loc, rowidx, start, start_dev, nnz, \
                       grad, (int)D, scale, target, unique_dev, (int)sb);                                              \
    break;
  switch (np) {
    TTEMB_SCATTER(1)
    TTEMB_SCATTER(2)
    TTEMB_SCATTER(3)
    TTEMB_SCATTER(4)
    TTEMB_SCATTER(5)
    TTEMB_SCATTER(6)
    TTEMB_SCATTER(7)
    default:
    TTEMB_SCATTER(8)
  }
#undef TTEMB_SCATTER
  profile_end(7, st);
  return check_hip(hipGetLastError(), "cache_scatter_add_kernel");
}

int launch_cache_rowwise_adagrad(const int32_t* loc, const int64_t* rowidx, int64_t start,
                                 const int32_t* start_dev, int64_t nnz, const float* grad, int64_t D,
                                 float lr, float eps, float* state_sum, float* weight, hipStream_t st) {
  const int64_t span = start_dev ? nnz : nnz - start;
  if (span <= 0) return TTEMB_OK;
  hipLaunchKernelGGL(cache_rowwise_adagrad_kernel, dim3(wave_blocks(span)), dim3(256), 0, st, loc,
                     rowidx, start, start_dev, nnz, grad, (int)D, lr, eps, state_sum, weight);
  return check_hip(hipGetLastError(), "cache_rowwise_adagrad_kernel");
}

}  // namespace ttemb
