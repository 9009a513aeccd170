// Micro-benchmark: does v_mfma_f32_16x16x4_f32 (f32 in, the MFMA every contraction of the TT chain uses) share the
// SIMD with other work?  Four questions, each answered by wall time of one 512-thread workgroup per CU (two waves
// per SIMD; waves 0-3 and 4-7 are SIMD partners):
//   1. one wave per SIMD, MFMA only                                    -> cycles per MFMA (expected 32)
//   2. one wave per SIMD, MFMA + k independent v_fma_f32 per MFMA      -> does VALU hide under the MFMA of the SAME wave?
//   3. two waves per SIMD: one MFMA-only, the partner VALU-only        -> do they overlap across waves?
//   4. two waves per SIMD: one MFMA-only, the partner ds_read_b32-only -> do LDS reads overlap?
// hipcc -O3 --offload-arch=gfx950 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 20000;

// role: 0 = idle, 1 = MFMA only, 2 = VALU only, 3 = MFMA + K VALU per MFMA, 4 = ds_read_b32 only, 5 = MFMA + K ds_read
template <int K>
__device__ __forceinline__ void body(int role, float* out, float* lds) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + lane * 1e-3f, b = 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.1f * i + lane;
  if (role == 1) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
    }
  } else if (role == 3) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
      }
    }
  } else if (role == 4) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        float x;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x) : "v"(lane * 4), "n"(0));
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        v[i & 7] += x;
      }
    }
  } else if (role == 5) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float x;
          asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lane * 4 + 256 * k));
          v[k & 7] = x;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  else if (role == 6) {   // the 16-block 4x4x1 form (512 flop per instruction)
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[threadIdx.x] = s + lds[lane];
}

template <int K>
__global__ __launch_bounds__(512) void k(int role_lo, int role_hi, float* out) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  body<K>(wave < 4 ? role_lo : role_hi, out, lds);
}

template <int K>
static float run(int role_lo, int role_hi, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, role_lo, role_hi, out);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, role_lo, role_hi, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 4);
  const double mf = 4.0 * ITERS;   // MFMAs per wave
  auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / mf; };
  float t;
  t = run<0>(1, 0, out); printf("1 wave/SIMD  MFMA only               : %7.3f ms = %6.1f cyc/MFMA @2.4GHz\n", t, cyc(t));
  t = run<0>(1, 1, out); printf("2 waves/SIMD MFMA | MFMA             : %7.3f ms = %6.1f cyc per MFMA pair\n", t, cyc(t));
  t = run<0>(2, 0, out); printf("1 wave/SIMD  VALU only (32 fma/iter) : %7.3f ms = %6.1f cyc per 8 v_fma\n", t, cyc(t));
  t = run<0>(1, 2, out); printf("2 waves/SIMD MFMA | VALU (8 fma per MFMA slot): %7.3f ms (sum if serial, max if overlapped)\n", t);
  t = run<2>(3, 0, out); printf("1 wave/SIMD  MFMA + 2 v_fma each     : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<4>(3, 0, out); printf("1 wave/SIMD  MFMA + 4 v_fma each     : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<6>(3, 0, out); printf("1 wave/SIMD  MFMA + 6 v_fma each     : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<8>(3, 0, out); printf("1 wave/SIMD  MFMA + 8 v_fma each     : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<0>(4, 0, out); printf("1 wave/SIMD  ds_read_b32 only (32/iter): %7.3f ms = %6.1f cyc per 8 reads\n", t, cyc(t));
  t = run<0>(1, 4, out); printf("2 waves/SIMD MFMA | ds_read_b32      : %7.3f ms\n", t);
  t = run<2>(5, 0, out); printf("1 wave/SIMD  MFMA + 2 ds_read each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<4>(5, 0, out); printf("1 wave/SIMD  MFMA + 4 ds_read each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<0>(6, 0, out); printf("1 wave/SIMD  v_mfma_f32_4x4x1_16b only: %7.3f ms = %6.1f cyc per instruction (512 flop)\n", t, cyc(t));
  t = run<0>(6, 6, out); printf("2 waves/SIMD 4x4x1 | 4x4x1           : %7.3f ms = %6.1f cyc per pair\n", t, cyc(t));
  t = run<2>(3, 3, out); printf("2 waves/SIMD both MFMA + 2 v_fma     : %7.3f ms = %6.1f cyc per MFMA pair\n", t, cyc(t));
  t = run<4>(3, 3, out); printf("2 waves/SIMD both MFMA + 4 v_fma     : %7.3f ms = %6.1f cyc per MFMA pair\n", t, cyc(t));
  return 0;
}
