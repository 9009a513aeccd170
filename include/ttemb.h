/*
 * ttemb.h -- C ABI of libttemb_hip.so: the MI355X (gfx950) Tensor-Train embedding
 * hot path.
 *
 * Every entry point replaces one function of the reference's pybind module
 * `tt_embeddings` (FBTT/tt_embeddings.cpp:131-161); the replaced interface is cited
 * per function.  The boundary is plain C: raw device pointers, explicit sizes, an
 * opaque `hipStream_t` passed as `void*`, a caller-provided workspace.  No allocation,
 * no host synchronisation unless a function says so, and no state in the library between
 * calls except process-wide DIAGNOSTIC switches that never change a result: the kernel-family
 * override (ttemb_set_path), the piece limits (ttemb_set_piece_limits), the spin limit (ttemb_set_spin_limit) and the
 * event profiler (ttemb_profile_enable) -- and one pinned host word through which a device-side wait that ran out is
 * reported (ttemb_init / ttemb_status).  A caller that never touches them has none.
 *
 * Tracing: with TTEMB_ROCTX=1 in the environment every lookup / cache entry point is bracketed by a roctx range
 * (roctxRangePush / Pop from librocprofiler-sdk-roctx.so or libroctx64.so, looked up at run time), so a
 * `rocprofv3 --marker-trace --kernel-trace` timeline shows the calls above their kernels.
 *
 * The workspace: every op leaves the first 40 KB of its workspace alone except the grouped
 * lookup, which keeps a call counter and a few pre-tagged counters there (in DEVICE memory, so
 * that a replayed HIP graph counts on).  Whatever those bytes hold is valid -- a fresh or
 * recycled buffer costs the first call a slower counting step, never a wrong result -- so
 * nothing has to be initialised; handing the same workspace to successive calls (what the
 * Python class does) keeps the fast step.  One workspace must not serve two streams at once.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - cores[t] is float32 [p_t][R_t*q_t*R_{t+1}] row-major, a core row being a
 *     row-major [R_t][q_t][R_{t+1}] block (FBTT/tt_embeddings_ops.py:528-545);
 *   - ids are int64, decomposed with L = [p1*p2.., .., 1] by repeated div/mod
 *     (FBTT/tt_embeddings_cuda.cu:796-802);
 *   - `rowidx[n]` is the bag (output row) of position n, non-decreasing inside the
 *     TT part and inside the cached part (what preprocess produces).  Where a function
 *     takes both `rowidx` and `offsets`, rowidx may be NULL when offsets (int64[B+1]) is
 *     given and the ids are exactly the concatenated bags: the rows are then derived
 *     from offsets inside the call (no separate ttemb_preprocess launch needed);
 *   - return value: 0 = ok, <0 = error (TTEMB_E_*), text via ttemb_last_error().
 */
#ifndef TTEMB_H_
#define TTEMB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTEMB_ABI_VERSION 4   /* 4: + the *_window calls (one table of a table-batched call); 3: + ttemb_init / ttemb_status / ttemb_set_spin_limit, TTEMB_FAMILY_PREFIX_IN_CHAIN, plan of G + 3 group words */
#define TTEMB_MAX_CORES 4

enum {
  TTEMB_OK = 0,
  TTEMB_E_BADARG = -1,     /* shape / size / null-pointer check failed            */
  TTEMB_E_WORKSPACE = -2,  /* workspace too small (see ttemb_workspace_bytes)      */
  TTEMB_E_UNSUPPORTED = -3,/* shape outside what the kernels support               */
  TTEMB_E_HIP = -4         /* a HIP runtime call failed (message has the detail)   */
};

/* TT factorisation of one table.  R has T+1 entries with R[0] = R[T] = 1. */
typedef struct ttemb_shape {
  int32_t T;                    /* number of cores, 2..4                          */
  int32_t p[TTEMB_MAX_CORES];   /* row factors,    prod(p) >= num_embeddings       */
  int32_t q[TTEMB_MAX_CORES];   /* column factors, prod(q) == D, D % 4 == 0        */
  int32_t R[TTEMB_MAX_CORES + 1];
} ttemb_shape_t;

/* Which computation a workspace is being sized for. */
enum {
  TTEMB_OP_FORWARD = 0,
  TTEMB_OP_BACKWARD = 1,        /* dense / sgd / adagrad all use the same size    */
  TTEMB_OP_PREPROCESS = 2,
  TTEMB_OP_CACHE_POPULATE = 3   /* nnz = hashtbl_size, B = cache rows              */
};

/* Kernel-selection knob for forward/backward (tests drive both; 0 is the default). */
enum {
  TTEMB_PATH_AUTO = 0,          /* grouped MFMA path when the shape has it and the batch is past its measured crossover
                                   (max(4096, p0*p1/4) ids; ranks >= 64: p0*p1/8 at 64, p0*p1/24 at 128, any at 256), else the
                                   per-bag MFMA kernels (ids + offsets), else generic */
  TTEMB_PATH_GENERIC = 1,       /* shape-generic wave-per-id kernels (T = 2..4)    */
  TTEMB_PATH_FAST3 = 2,         /* sorted / grouped MFMA path, T == 3 only         */
  TTEMB_PATH_PER_BAG = 3        /* one wavefront per bag (MFMA per id) whenever the shape has it and the ids come
                                   with their offsets, at every batch size; else generic.  For crossover measurements */
};

int ttemb_abi_version(void);
const char* ttemb_last_error(void);   /* thread-local, valid until the next call  */

/* Bytes of scratch the op needs for `nnz` ids and `B` bags (0 is a valid answer). */
int64_t ttemb_workspace_bytes(const ttemb_shape_t* shape, int32_t op, int64_t nnz, int64_t B);

/* Bytes of an id-grouping plan for `nnz` ids (0 when the selected kernel family needs none).
 * ttemb_forward leaves its grouping of the ids in a caller buffer of this size; passing the same
 * buffer to the backward of the SAME (indices, rowidx, nnz, nnz_dev) skips the regrouping.  The plan also
 * carries the prefix products G0[i0].G1[i1] the forward used: a backward handed the plan differentiates the
 * chain at those (the saved-tensor meaning of autograd); one that is not forms them from the cores it is given. */
int64_t ttemb_plan_bytes(const ttemb_shape_t* shape, int64_t nnz);

/* Select the kernel family used by later calls, process-wide (TTEMB_PATH_*). */
int ttemb_set_path(int32_t path);

/* A call whose [B][D] tensor does not fit one 32-bit window of byte offsets (2^24 rows or 2 GiB: what the grouped kernels
 * address through buffer descriptors) -- SAGE.inference looks up every node of the graph in one call, gnn_model.py:220-253 --
 * runs on the grouped path as a sequence of PIECES of the id list (the reference chunks any call by batch_count,
 * tt_embeddings_cuda.cu:1011-1027).  The boundaries follow `offsets` and are computed on the device: no host
 * synchronisation.  Needs `offsets`; ttemb_plan_bytes() is 0 for such a call (the backward regroups piece by piece), and
 * the fused backward entry points write the summed gradient into the workspace and step once.
 * DIAGNOSTIC, process-wide: smaller limits than the hardware's (rows per piece, ids per piece; 0 = default), so that
 * tests can cut a small call into many pieces.  Never changes a result. */
int ttemb_set_piece_limits(int64_t rows, int64_t ids);

/* DIAGNOSTIC, process-wide: the number of ids from which the backward of the wide-rank chain (ranks 64 / 128 / 256) reduces
 * dG2 inside its chunk kernel (LDS slabs, no E table) instead of writing one E row per id and reducing them in a second
 * kernel.  0 = the library's rule (8 ids per slab row: the slabs are written whatever the batch), 1 = always, a huge value =
 * never.  Changes workspace sizes (ask ttemb_workspace_bytes again); never changes a result beyond fp32 summation order. */
int ttemb_set_wide_slab_min_ids(int64_t ids);

/* Device-side faults.  The grouping pass of the grouped lookup has two BOUNDED waits between workgroups (the take-over of a
 * range counter that carries another call's tag, and the place step's look back at the chunk totals of the ranges before
 * it).  They cannot run out on a GPU that runs the launch's workgroups together; under CU masking, profiler
 * serialisation or several processes time-sliced on one GPU they can.  A wait that runs out never becomes plausible
 * numbers (the reference's kernels cannot time out; it checks its launches with AT_CUDA_CHECK,
 * FBTT/tt_embeddings_cuda.cu:1666,1742,1845):
 *   - the call's plan is POISONED on the device: the forward writes NaN into its whole output window, a backward on that
 *     plan writes NaN gradients; no kernel walks the chunk table.  A FUSED backward (ttemb_backward_sgd / _adagrad) on a
 *     poisoned plan leaves the parameters and the optimizer state UNTOUCHED when the host word below exists (the step can
 *     be repeated); without the word -- a graph captured before ttemb_init() -- it writes NaN parameters, the only signal
 *     such a graph has;
 *   - the reason is stored to a pinned host word, and the next ttemb_forward* / ttemb_backward* call of the process that
 *     sees it returns TTEMB_E_HIP with a message (once per fault, consumed by exactly one caller through an atomic
 *     exchange; no synchronisation is added for this).  The error refers to an EARLIER call; the call that returns it has
 *     not been started.
 * ttemb_init() creates that word (64 bytes of pinned, device-mapped host memory -- the ONE allocation this library ever
 * makes, which is why it is a call of its own and no lookup does it): call it once per process, from the host, OUTSIDE any
 * stream capture and before capturing a graph that contains grouped lookups.  Lookups made before it report an expired wait
 * through their NaN results only.  Idempotent; TTEMB_E_HIP when the memory cannot be had.
 * ttemb_status() is the check as a call of its own, for callers that synchronise: 0, or TTEMB_E_HIP (and the fault is
 * consumed); it creates the word when ttemb_init() has not.  ttemb_set_spin_limit is a DIAGNOSTIC, process-wide: tries of
 * those waits (0 = the default of 2^20, about 1.5 s; negative = none at all, every wait expires -- how tests reach the
 * fault path). */
int ttemb_init(void);
/* Byte offset, inside the header every lookup workspace begins with, of the uint32 the LAST grouped backward on that
 * workspace left: 1 = its plan was poisoned and the host word exists (gradients NaN, a fused update skipped), 0 = healthy.
 * Written by every backward of the grouped families (ttemb_kernel_family & 7 in {GROUPED, GROUPED_WIDE}); not written by the
 * others, which have no bounded waits. */
#define TTEMB_HEADER_POISON_OFFSET 32784
int ttemb_status(void);
int ttemb_set_spin_limit(int64_t tries);

/* Which kernels a lookup of `nnz` ids in `B` bags on this table would run, under the current ttemb_set_path: a
 * DIAGNOSTIC for tests, benchmarks and tuners (tuning_SAGE.py searches ranks in [2, 256]) -- it launches nothing.
 * `ids_with_offsets` != 0: the ids come with their bag boundaries and no row index (what TTEmbeddingBag.forward passes).
 *   TTEMB_FAMILY_SCALAR   wave-per-id kernels, plain FMA (any T, any shape)
 *   TTEMB_FAMILY_PER_BAG  one wavefront per bag, fp32 MFMA per id, an instantiated (q, ranks) shape
 *   TTEMB_FAMILY_PER_BAG_RT  the same with run-time (q, ranks): any 3-core shape with q0, q2 <= 16 and q0 q1 <= 64
 *   TTEMB_FAMILY_GROUPED  the grouped chain (ranks <= 32, an instantiated shape)
 *   TTEMB_FAMILY_GROUPED_WIDE  the grouped chain of ranks 64 / 128 / 256
 * | TTEMB_FAMILY_MERGED when a 2- or 4-core table rides on a 3-core view (a virtual core built per call),
 * | TTEMB_FAMILY_PADDED when a 3-core table whose ranks are off the instantiated list rides on the grouped kernels of the
 *   next listed rank through zero-padded copies of its cores (same rows, same gradients: the added rank positions hold
 *   zeros),
 * | TTEMB_FAMILY_PREFIX_IN_CHAIN when a whole ttemb_forward of this size forms the prefix products G0[i0].G1[i1] inside
 *   its chain kernel (fewer than 8 ids per (i0, i1) group on average, 16 for q0 = 8: 819 200 ids on the papers100M table) instead of in
 *   a launch of its own; the plan it leaves and every result are the same,
 * | TTEMB_FAMILY_GROUP_PRODUCTS_IN_CHAIN when a backward of this size forms the two per-group products (the dG0 parts and
 *   dG1: FBTT/tt_embeddings_cuda.cu:531-609 runs them as the t = 0 GEMM pair over a partial-product table) inside its chunk
 *   kernel, while a group's dP is still in registers: no dP table, no epilogue launch.  Same rule on the ids per group,
 *   for shapes whose dG2 reduction is not already fused into the chunk kernel (rank 32; large p2). */
enum {
  TTEMB_FAMILY_SCALAR = 0,
  TTEMB_FAMILY_PER_BAG = 1,
  TTEMB_FAMILY_PER_BAG_RT = 2,
  TTEMB_FAMILY_GROUPED = 3,
  TTEMB_FAMILY_GROUPED_WIDE = 4,
  TTEMB_FAMILY_MERGED = 16,
  TTEMB_FAMILY_PADDED = 32,
  TTEMB_FAMILY_PREFIX_IN_CHAIN = 64,
  TTEMB_FAMILY_GROUP_PRODUCTS_IN_CHAIN = 128
};
int ttemb_kernel_family(const ttemb_shape_t* shape, int64_t nnz, int64_t B, int32_t ids_with_offsets);

/* Measurement hook (bench.py's roofline leg).  While enabled (process-wide), the
 * main chain kernel of every ttemb_forward / ttemb_backward_* call is bracketed by
 * hipEvents on the call's stream.  ttemb_profile_read waits for the most recent bracket of
 * `which` (0 = forward chain kernel, 1 = all backward chain kernels, 2 = the backward chunk
 * kernel alone, 3 = the id-grouping pass including the prefix-product kernel, 4 = the cache probe pass of
 * ttemb_preprocess / ttemb_preprocess_update, 5 = its partition scatter, 6 = the cached-row gather of ttemb_cache_forward,
 * 7 = the cached-row update of ttemb_cache_backward_*, 8 = the group epilogue kernel of the backward, 9 = its finalize
 * kernel) and returns its duration in milliseconds; a slot no launch has bracketed since the last ttemb_profile_enable(1)
 * is TTEMB_E_BADARG (enabling forgets every earlier bracket).  Off by default; costs two event records per bracket when on. */
int ttemb_profile_enable(int32_t on);
int ttemb_profile_read(int32_t which, float* ms_host);

/* ---------------------------------------------------------------------------------
 * tt_forward  (replaces tt_embeddings.tt_forward -- FBTT/tt_embeddings.cpp:132,
 * tt_embeddings_forward_cuda FBTT/tt_embeddings_cuda.cu:967-1081).
 * output[B][D] is fully written: bag sums of the TT rows of indices[0:nnz], zeros
 * for bags with no id in that range.  `nnz_dev`, when non-null, points at a device
 * int32 holding the live count (<= nnz); the launch is sized by nnz and the kernels
 * read the count themselves, so no host sync is needed after ttemb_preprocess.
 * `offsets` (int64[B+1]) describes the bags of the ORIGINAL id list: with it only bags whose
 * length is not 1 are zero-filled before the lookups (a bag of one id has exactly one writer,
 * this call or ttemb_cache_forward(offsets)); with offsets == NULL the whole output is
 * zero-filled first and single-id detection falls back to neighbouring rowidx values.
 * `plan` (nullable, ttemb_plan_bytes() bytes) receives the id grouping (and the prefix products) for the backward.
 * plan == NULL is the INFERENCE form: the grouping lives and dies in the workspace, and a forward that forms its prefix
 * products inside the chain kernel (TTEMB_FAMILY_PREFIX_IN_CHAIN) stores none of them -- same rows, less HBM traffic.
 * There is no batch_count chunking: intermediates never leave the chip.
 * ------------------------------------------------------------------------------- */
int ttemb_forward(const ttemb_shape_t* shape, const float* const* cores,
                  const int64_t* indices, const int64_t* rowidx, const int64_t* offsets,
                  int64_t nnz, const int32_t* nnz_dev, int64_t B, float* output,
                  void* workspace, int64_t workspace_bytes, void* plan, int64_t plan_bytes,
                  void* stream);

/* Two-phase forward, for a caller that wants to overlap the id-only work with something that still writes the
 * cores (ttemb_dist.TTDataParallel: the all-reduce + update of the previous step):
 *   ttemb_forward_group   everything that depends only on the ids -- the grouping pass into `plan`, bag rows,
 *                         zeroing of the output rows of bags that do not hold exactly one id.  `cores` is not read.
 *   ttemb_forward_lookup  the rest: prefix products and the chain kernel, on the plan the first call filled.
 * Same arguments for both; `plan` (ttemb_plan_bytes() bytes) is required whenever ttemb_plan_bytes() != 0.
 * ttemb_forward == group followed by lookup.  When the selected kernel family keeps no plan, group does nothing
 * and lookup is the whole forward. */
int ttemb_forward_group(const ttemb_shape_t* shape, const float* const* cores,
                  const int64_t* indices, const int64_t* rowidx, const int64_t* offsets,
                  int64_t nnz, const int32_t* nnz_dev, int64_t B, float* output,
                  void* workspace, int64_t workspace_bytes, void* plan, int64_t plan_bytes,
                  void* stream);
int ttemb_forward_lookup(const ttemb_shape_t* shape, const float* const* cores,
                  const int64_t* indices, const int64_t* rowidx, const int64_t* offsets,
                  int64_t nnz, const int32_t* nnz_dev, int64_t B, float* output,
                  void* workspace, int64_t workspace_bytes, void* plan, int64_t plan_bytes,
                  void* stream);

/* ---------------------------------------------------------------------------------
 * tt_dense_backward  (tt_embeddings.tt_dense_backward -- tt_embeddings.cpp:133-136,
 * tt_embeddings_cuda.cu:656-686, 421-654).  d_cores[t] (same shape as cores[t]) is
 * overwritten with the gradient of sum(output * d_output) w.r.t. cores[t].
 * ------------------------------------------------------------------------------- */
int ttemb_backward_dense(const ttemb_shape_t* shape, const float* const* cores,
                         const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                         const int32_t* nnz_dev, int64_t B, const float* d_output,
                         float* const* d_cores,
                         void* workspace, int64_t workspace_bytes, const void* plan, int64_t plan_bytes,
                         void* stream);

/* tt_sgd_backward (tt_embeddings.cpp:137-138, tt_embeddings_cuda.cu:688-719):
 * cores[t] -= lr * d_core_t, every row (the reference's grid defect at :633-651 is
 * not reproduced).  The gradient lives in the workspace only. */
int ttemb_backward_sgd(const ttemb_shape_t* shape, float* const* cores,
                       const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                       const int32_t* nnz_dev, int64_t B, const float* d_output, float lr,
                       void* workspace, int64_t workspace_bytes, const void* plan, int64_t plan_bytes,
                       void* stream);

/* tt_adagrad_backward (tt_embeddings.cpp:139-142, tt_embeddings_cuda.cu:721-754,
 * 399-419): state += g*g ; core -= lr * g / (sqrt(state) + eps). */
int ttemb_backward_adagrad(const ttemb_shape_t* shape, float* const* cores,
                           float* const* opt_state,
                           const int64_t* indices, const int64_t* rowidx, const int64_t* offsets, int64_t nnz,
                           const int32_t* nnz_dev, int64_t B, const float* d_output,
                           float lr, float eps,
                           void* workspace, int64_t workspace_bytes, const void* plan, int64_t plan_bytes,
                           void* stream);

/* ---------------------------------------------------------------------------------
 * A WINDOW of a longer id list: one table of a table-batched call.
 * Replaces, for `num_tables` > 1: the `tableidx` the reference's kernels derive per id (compute_rowidx_kernel,
 * tt_embeddings_cuda.cu:1349-1365) and index their per-table core pointers with (tt_embeddings_cuda.cu:443-609,
 * called by TableBatchedTTEmbeddingBag.forward, tt_embeddings_ops.py:862-916) -- the host never learns where a table's ids
 * begin.  Here a table is a call of its own on a window: the bags [bag0, bag0 + B) of `offsets` (int64[bags_total + 1], the
 * whole call's) and the ids that belong to them; where the window begins and ends in `indices` is read from `offsets` ON THE
 * DEVICE -- no host synchronisation.  `nnz` = length of the WHOLE id list (the launches are sized by it; a window that turns
 * out empty costs its launches and nothing else); `output` / `d_output` = the [bags_total][D] tensor of the whole call (the
 * window's rows are bag0 .. bag0 + B - 1; no other row is touched).  `cores` / `d_cores` / `opt_state` are the window's own
 * table.  Served by the grouped kernels, whatever the size: TTEMB_E_UNSUPPORTED (from the size query already) for a shape they
 * do not cover (2- / 4-core tables, ranks off the list) or a window past their limits -- the caller then splits the id list on
 * the host, one plain call per table.  Workspace: ttemb_window_workspace_bytes(shape, TTEMB_OP_FORWARD | TTEMB_OP_BACKWARD, ...);
 * one workspace serves every window and every other op.  A window keeps no plan: its backward groups the window's ids again.
 * ------------------------------------------------------------------------------- */
int64_t ttemb_window_workspace_bytes(const ttemb_shape_t* shape, int32_t op, int64_t nnz, int64_t bags_total, int64_t B);
int ttemb_forward_window(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                         const int64_t* offsets, int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B,
                         float* output, void* workspace, int64_t workspace_bytes, void* stream);
int ttemb_backward_dense_window(const ttemb_shape_t* shape, const float* const* cores, const int64_t* indices,
                                const int64_t* offsets, int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B,
                                const float* d_output, float* const* d_cores, void* workspace,
                                int64_t workspace_bytes, void* stream);
int ttemb_backward_sgd_window(const ttemb_shape_t* shape, float* const* cores, const int64_t* indices,
                              const int64_t* offsets, int64_t nnz, int64_t bags_total, int64_t bag0, int64_t B,
                              const float* d_output, float lr, void* workspace, int64_t workspace_bytes, void* stream);
int ttemb_backward_adagrad_window(const ttemb_shape_t* shape, float* const* cores, float* const* opt_state,
                                  const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t bags_total,
                                  int64_t bag0, int64_t B, const float* d_output, float lr, float eps,
                                  void* workspace, int64_t workspace_bytes, void* stream);

/* Flat optimiser epilogues over n floats (used after the data-parallel all-reduce of
 * the flattened core gradients; same arithmetic as tt_embeddings_cuda.cu:381-419). */
int ttemb_sgd_step(float* weights, const float* grads, int64_t n, float lr, void* stream);
/* The same step, skipped as a whole when the 32-bit device word *skip is non-zero (tested bit-wise: a float count k.0f and
 * the uint32 poison word of a workspace header both work; a single process passes the header word itself).  The data-parallel step (no counterpart in the
 * reference: its DDP path is a stub, sage_dgl_partition.py:198-255) all-reduces, next to the gradients, the number of ranks
 * whose gradient came from a POISONED plan (the word at TTEMB_HEADER_POISON_OFFSET of the workspace the backward ran on,
 * see "Device-side faults"): when any did, the summed gradient is NaN and every rank skips the update together -- replicas
 * stay identical and untouched, the faulted rank's next call returns TTEMB_E_HIP. */
int ttemb_sgd_step_guarded(float* weights, const float* grads, int64_t n, float lr, const float* skip, void* stream);
int ttemb_adagrad_step(float* weights, float* state, const float* grads, int64_t n,
                       float lr, float eps, void* stream);

/* ---------------------------------------------------------------------------------
 * LFU hash-table cache.
 * update_cache_state (tt_embeddings.cpp:144, tt_embeddings_cuda.cu:1083-1119):
 * insert every id with count+1 (Murmur3-style hash, <=3 linear probes, 64-bit CAS;
 * hashtbl_cuda_utils.cuh:48-133).  hashtbl/cache_freq are int64[H].
 * ------------------------------------------------------------------------------- */
int ttemb_cache_update(const int64_t* indices, int64_t nnz, int64_t* hashtbl,
                       int64_t* cache_freq, int64_t H, void* stream);

/* The same update with the reference's ONE-SWEEP insert (hashtbl_cuda_utils.cuh:102-133: the first probe slot that is
 * empty or holds the key takes the count), for callers that need the reference's table bit for bit.  ttemb_cache_update
 * looks the key up in all of its probe slots first: identical tables until cache_populate has evicted keys; afterwards
 * the one-sweep form can re-insert a cached id into a hole in front of its slot, after which the id resolves to the new
 * slot and drops out of the cache (which ids depends on thread order). */
int ttemb_cache_update_one_sweep(const int64_t* indices, int64_t nnz, int64_t* hashtbl, int64_t* cache_freq,
                                 int64_t H, void* stream);

/* cache_populate (tt_embeddings.cpp:145, tt_embeddings_cuda.cu:1270-1347): stable
 * descending sort of the H slots by frequency, keep ranks < C (cache_state[slot] =
 * rank), evict the rest, fill cache_weight[C][D] with the TT rows of the kept ids
 * (empty ranks stand for id 0, as in the reference). */
int ttemb_cache_populate(const ttemb_shape_t* shape, const float* const* cores,
                         int64_t* hashtbl, int64_t* cache_freq, int32_t* cache_state,
                         int64_t H, float* cache_weight, int64_t C,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* preprocess_indices_sync (tt_embeddings.cpp:146-149, tt_embeddings_cuda.cu:1388-1507),
 * single table.  Always writes rowidx_out[nnz] from offsets[B+1].  With warmup != 0
 * (or H == 0) indices are passed through (indices_out may alias indices or be null) and
 * *nnz_tt_dev = nnz.  Otherwise ids are looked up in the cache and (indices, rowidx,
 * cache_loc) are partitioned: TT ids first in input order, cached ids from the end
 * backwards (the CUB DevicePartition::Flagged order); *nnz_tt_dev = #TT ids.
 * Does NOT synchronise: the count stays on the device (pass it to the other entry
 * points as nnz_dev, or copy it back yourself).
 * `dup_stamp` (nullable; int32[C] of scratch, any content) turns on duplicate detection among the cached ids:
 * nnz_tt_dev must then hold TWO int32 and nnz_tt_dev[1] becomes 1 when some cache row is met more than once in
 * this call, else 0 -- the word ttemb_cache_backward_sgd / _dense take as `dup_dev`.  (Every cached id stamps its
 * row with its position, the partition step reads the stamps back.)  `epoch` is ignored (ABI 1 needed it). */
int ttemb_preprocess(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B,
                     int32_t warmup, const int64_t* hashtbl, const int32_t* cache_state,
                     int64_t H, int64_t* indices_out, int64_t* rowidx_out,
                     int32_t* cache_loc_out, int32_t* nnz_tt_dev, int32_t* dup_stamp, int32_t epoch,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* update_cache_state + preprocess_indices_sync in ONE probe pass -- what TTEmbeddingBag.forward does with a live cache
 * (tt_embeddings_ops.py:836-870 calls the two back to back on the same ids, and both visit the same <= 3 slots per id).
 * Same hashtbl / cache_freq / outputs as ttemb_cache_update followed by ttemb_preprocess(warmup = 0): the find-first
 * update never moves or evicts a key, so an id resolves to the same slot before and after its batch was counted.
 * (A caller that needs the reference's one-sweep insert keeps the two calls apart.) */
int ttemb_preprocess_update(const int64_t* indices, const int64_t* offsets, int64_t nnz, int64_t B,
                            int64_t* hashtbl, int64_t* cache_freq, const int32_t* cache_state, int64_t H,
                            int64_t* indices_out, int64_t* rowidx_out, int32_t* cache_loc_out,
                            int32_t* nnz_tt_dev, int32_t* dup_stamp, void* workspace, int64_t workspace_bytes,
                            void* stream);

/* cache_forward (tt_embeddings.cpp:151, tt_embeddings_cuda.cu:1509-1583):
 * output[rowidx[n]] += cache_weight[cache_loc[n]] for n in [start, nnz).  `start` is
 * the host value, or *start_dev when start_dev is non-null.  With `offsets` (the same
 * int64[B+1] given to ttemb_forward) a bag of exactly one id is written with a plain
 * store -- ttemb_forward(offsets) left such rows untouched, so nothing is read or zeroed. */
int ttemb_cache_forward(const int32_t* cache_loc, const int64_t* rowidx, const int64_t* offsets, int64_t start,
                        const int32_t* start_dev, int64_t nnz, const float* cache_weight,
                        int64_t D, float* output, void* stream);

/* cache_backward_sgd (tt_embeddings.cpp:152, tt_embeddings_cuda.cu:1585-1668):
 * cache_weight[loc] -= lr * d_output[row].  The adds are float atomics (an id may occur twice, as in the
 * reference) unless `dup_dev` (nullable) points at a device int32 that is 0: the word ttemb_preprocess wrote
 * when no cache row occurs twice in the call -- every row then has one writer and is updated by a plain
 * read-modify-write. */
int ttemb_cache_backward_sgd(const int32_t* cache_loc, const int64_t* rowidx, int64_t start,
                             const int32_t* start_dev, int64_t nnz, const float* d_output,
                             int64_t D, float lr, float* cache_weight, const int32_t* dup_dev, void* stream);

/* cache_backward_dense (tt_embeddings.cpp:153-156, tt_embeddings_cuda.cu:1670-1744):
 * d_cache_weight[C][D] is zero-filled, then d_cache_weight[loc] += d_output[row]. */
int ttemb_cache_backward_dense(const int32_t* cache_loc, const int64_t* rowidx, int64_t start,
                               const int32_t* start_dev, int64_t nnz, const float* d_output,
                               int64_t D, int64_t C, float* d_cache_weight, const int32_t* dup_dev, void* stream);

/* cache_backward_rowwise_adagrad_approx (tt_embeddings.cpp:157-160,
 * tt_embeddings_cuda.cu:1746-1846): per id, g2 = mean(d_output[row]^2);
 * old = atomic_add(state[loc], g2); cache_weight[loc] -= lr/(sqrt(old+g2)+eps) * d_output[row]. */
int ttemb_cache_backward_rowwise_adagrad(const int32_t* cache_loc, const int64_t* rowidx,
                                         int64_t start, const int32_t* start_dev, int64_t nnz,
                                         const float* d_output, int64_t D, float lr, float eps,
                                         float* cache_state_sum, float* cache_weight,
                                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TTEMB_H_ */
