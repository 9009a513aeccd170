// Shape-generic TT chain kernels (T = 2..4, any q / rank): one wavefront per id.
//
// These are the correctness backstop and the path for shapes the MFMA fast path does
// not cover.  Partials stay in LDS -- nothing like the reference's tr_0/tr_1 HBM
// round trip (FBTT/tt_embeddings_cuda.cu:1011-1017) or its per-id pointer arrays
// (:1018-1026) exists here.
#include "ttemb_common.h"

namespace ttemb {

static inline int round4(int x) { return (x + 3) & ~3; }


int64_t generic_fwd_lds_bytes(const DevShape& s) {
  return 2ll * round4(s.part_max) * sizeof(float);
}

int64_t generic_bwd_lds_bytes(const DevShape& s) {
  int64_t v = 0;
  for (int t = 0; t + 1 < s.T; ++t) v += round4(s.part_len[t]);
  int dvmax = s.D > s.part_max ? s.D : s.part_max;
  return (v + 2ll * round4(dvmax)) * sizeof(float);
}

// ---------------------------------------------------------------------------------
// forward: out[row(n)] (+)= G0[i0] . G1[i1] ... G_{T-1}[i_{T-1}]
// (reference: init_batch_gemm_forward_*T_kernel + 2x cublasGemmBatchedEx +
//  reduce_output_kernel, FBTT/tt_embeddings_cuda.cu:757-965, 1045-1077)
// A bag with exactly one id in the live range is written with plain stores; a bag with
// several ids accumulates with float atomics into a row the caller zeroed beforehand.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fwd_generic_kernel(DevShape s, CorePtrs cores,
                                                         const int64_t* __restrict__ indices,
                                                         const int64_t* __restrict__ rowidx,
                                                         const int64_t* __restrict__ offsets, int64_t B,
                                                         int64_t nnz,
                                                         const int32_t* __restrict__ nnz_dev,
                                                         float* __restrict__ output) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pm = (s.part_max + 3) & ~3;
  float* buf0 = smem;
  float* buf1 = smem + pm;
  const int lane = threadIdx.x;
  const int64_t cnt = live_count(nnz, nnz_dev);
  for (int64_t n = blockIdx.x; n < cnt; n += gridDim.x) {
    int it[TTEMB_MAX_CORES];
    split_index(s, indices[n], it);
    int64_t row = n;
    bool single = true;
    if (rowidx != nullptr) {
      row = rowidx[n];
      single = bag_is_single(rowidx, offsets, n, cnt, row);
    } else if (offsets != nullptr) {   // rows straight from the bag boundaries: no rowidx launch, no rowidx array
      row = bag_of_position(offsets, B, n);
      single = offsets[row + 1] - offsets[row] == 1;
    }
    const float* g0 = cores.c[0] + (int64_t)it[0] * s.row_len[0];
    for (int e = lane; e < s.row_len[0]; e += kWave) buf0[e] = g0[e];
    __syncthreads();
    float* prev = buf0;
    float* next = buf1;
    int M = s.q[0];
    for (int t = 1; t < s.T; ++t) {
      const int K = s.R[t];
      const int Nc = s.q[t] * s.R[t + 1];
      const float* g = cores.c[t] + (int64_t)it[t] * s.row_len[t];
      const bool last = (t == s.T - 1);
      for (int e = lane; e < M * Nc; e += kWave) {
        const int m = e / Nc;
        const int c = e - m * Nc;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(prev[m * K + k], g[k * Nc + c], acc);
        if (!last) {
          next[e] = acc;
        } else if (single) {
          output[row * s.D + e] = acc;
        } else {
          atomicAdd(&output[row * s.D + e], acc);
        }
      }
      __syncthreads();
      float* tmp = prev;
      prev = next;
      next = tmp;
      M *= s.q[t];
    }
  }
}

int launch_forward_generic(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                           const int64_t* rowidx, const int64_t* offsets, int64_t B, int64_t nnz,
                           const int32_t* nnz_dev, float* output, hipStream_t st) {
  if (nnz <= 0) return TTEMB_OK;
  const int64_t lds = generic_fwd_lds_bytes(s);
  static LdsGate gate;   // (q = 5,5,4 at ranks 256,256 -- a shape run_script.sh:250-268 trains -- needs 80 KB in the backward)
  int rc = allow_big_lds(reinterpret_cast<const void*>(fwd_generic_kernel), (size_t)lds, &gate, "generic forward (partial products)");
  if (rc) return rc;
  const int64_t grid = nnz < 262144 ? nnz : 262144;
  profile_begin(0, st);
  hipLaunchKernelGGL(fwd_generic_kernel, dim3((unsigned)grid), dim3(kWave), (size_t)lds, st, s,
                     cores, indices, rowidx, offsets, B, nnz, nnz_dev, output);
  profile_end(0, st);
  return check_hip(hipGetLastError(), "fwd_generic_kernel");
}

// ---------------------------------------------------------------------------------
// backward: d_core_t[i_t] += v_{t-1}^T dV_t ; dV_{t-1} = dV_t G_t[i_t]^T
// (reference: init_batch_gemm_backward_*T_kernel + 5 batched GEMMs +
//  update_d_tt_cores_kernel, FBTT/tt_embeddings_cuda.cu:81-379, 505-611).
// d_cores must be zero on entry.  Scatter is by float atomics, as in the reference.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void bwd_generic_kernel(DevShape s, CorePtrs cores,
                                                         const int64_t* __restrict__ indices,
                                                         const int64_t* __restrict__ rowidx,
                                                         const int64_t* __restrict__ offsets, int64_t B,
                                                         int64_t nnz,
                                                         const int32_t* __restrict__ nnz_dev,
                                                         const float* __restrict__ d_output,
                                                         CorePtrsMut d_cores) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* v[TTEMB_MAX_CORES];
  int off = 0;
  for (int t = 0; t + 1 < s.T; ++t) {
    v[t] = smem + off;
    off += (s.part_len[t] + 3) & ~3;
  }
  const int dvmax = ((s.D > s.part_max ? s.D : s.part_max) + 3) & ~3;
  float* dv0 = smem + off;
  float* dv1 = dv0 + dvmax;
  const int lane = threadIdx.x;
  const int64_t cnt = live_count(nnz, nnz_dev);
  for (int64_t n = blockIdx.x; n < cnt; n += gridDim.x) {
    int it[TTEMB_MAX_CORES];
    split_index(s, indices[n], it);
    const int64_t row = rowidx != nullptr ? rowidx[n] : (offsets != nullptr ? bag_of_position(offsets, B, n) : n);
    // recompute the forward partials v[0..T-2]
    const float* g0 = cores.c[0] + (int64_t)it[0] * s.row_len[0];
    for (int e = lane; e < s.row_len[0]; e += kWave) v[0][e] = g0[e];
    const float* dout = d_output + row * s.D;
    for (int e = lane; e < s.D; e += kWave) dv0[e] = dout[e];
    __syncthreads();
    int M = s.q[0];
    for (int t = 1; t + 1 < s.T; ++t) {
      const int K = s.R[t];
      const int Nc = s.q[t] * s.R[t + 1];
      const float* g = cores.c[t] + (int64_t)it[t] * s.row_len[t];
      for (int e = lane; e < M * Nc; e += kWave) {
        const int m = e / Nc;
        const int c = e - m * Nc;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(v[t - 1][m * K + k], g[k * Nc + c], acc);
        v[t][e] = acc;
      }
      __syncthreads();
      M *= s.q[t];
    }
    // M == q0*..*q_{T-2} here
    float* dv = dv0;
    float* dvn = dv1;
    for (int t = s.T - 1; t >= 1; --t) {
      const int K = s.R[t];
      const int Nc = s.q[t] * s.R[t + 1];
      const float* g = cores.c[t] + (int64_t)it[t] * s.row_len[t];
      float* dg = d_cores.c[t] + (int64_t)it[t] * s.row_len[t];
      const float* a = v[t - 1];
      for (int e = lane; e < K * Nc; e += kWave) {
        const int k = e / Nc;
        const int c = e - k * Nc;
        float acc = 0.f;
        for (int m = 0; m < M; ++m) acc = fmaf(a[m * K + k], dv[m * Nc + c], acc);
        atomicAdd(&dg[e], acc);
      }
      for (int e = lane; e < M * K; e += kWave) {
        const int m = e / K;
        const int k = e - m * K;
        float acc = 0.f;
        for (int c = 0; c < Nc; ++c) acc = fmaf(dv[m * Nc + c], g[k * Nc + c], acc);
        dvn[e] = acc;
      }
      __syncthreads();
      float* tmp = dv;
      dv = dvn;
      dvn = tmp;
      if (t > 1) M /= s.q[t - 1];
    }
    float* dg0 = d_cores.c[0] + (int64_t)it[0] * s.row_len[0];
    for (int e = lane; e < s.row_len[0]; e += kWave) atomicAdd(&dg0[e], dv[e]);
    __syncthreads();
  }
}

int launch_backward_generic(const DevShape& s, const CorePtrs& cores, const int64_t* indices,
                            const int64_t* rowidx, const int64_t* offsets, int64_t B, int64_t nnz,
                            const int32_t* nnz_dev, const float* d_output, const CorePtrsMut& d_cores, hipStream_t st) {
  if (nnz <= 0) return TTEMB_OK;
  const int64_t lds = generic_bwd_lds_bytes(s);
  static LdsGate gate;
  int rc = allow_big_lds(reinterpret_cast<const void*>(bwd_generic_kernel), (size_t)lds, &gate, "generic backward (partial products)");
  if (rc) return rc;
  const int64_t grid = nnz < 262144 ? nnz : 262144;
  profile_begin(1, st);
  profile_begin(2, st);
  hipLaunchKernelGGL(bwd_generic_kernel, dim3((unsigned)grid), dim3(kWave), (size_t)lds, st, s,
                     cores, indices, rowidx, offsets, B, nnz, nnz_dev, d_output, d_cores);
  profile_end(2, st);
  profile_end(1, st);
  return check_hip(hipGetLastError(), "bwd_generic_kernel");
}

}  // namespace ttemb
