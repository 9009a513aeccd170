// Micro-benchmark: the bf16 MFMA forms next to other work on the same SIMD (the f32-input form blocks the SIMD's vector
// issue: mfma_valu.hip).  One 512-thread workgroup per CU, waves 0-3 and 4-7 are SIMD partners.
//   1. cycles per v_mfma_f32_16x16x16_bf16 and per v_mfma_f32_16x16x32_bf16, one wave per SIMD and two
//   2. partner waves: bf16 MFMA only | v_fma_f32 only        -> sum (serial) or max (overlapped)?
//   3. one wave: bf16 MFMA + k v_fma_f32 each                -> does VALU of the SAME wave hide under its MFMA?
//   4. partner waves: bf16 MFMA only | ds_read_b128 only
// hipcc -O3 --offload-arch=gfx950 mfma_bf16.hip -o mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITERS = 20000;

// role: 0 idle, 1 f32 16x16x4, 2 VALU only, 7 bf16 16x16x16, 8 bf16 16x16x32, 9 = x16 + K v_fma each, 10 = x32 + K v_fma each,
// 11 ds_read_b128 only, 12 = x32 + K ds_read_b128 each
template <int K>
__device__ __forceinline__ void body(int role, float* out, float* lds) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = 1.0f + lane * 1e-3f, b = 0.5f;
  s16x4 a4 = {(short)(0x3f80 + lane), 0x3f80, 0x3f00, 0x3e80}, b4 = {0x3f80, 0x3f00, 0x3f80, 0x3f00};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(1.0f + 0.01f * (lane + i)); b8[i] = (__bf16)(0.5f + 0.01f * i); }
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
  } else if (role == 7) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
    }
  } else if (role == 8) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
    }
  } else if (role == 9) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
      }
    }
  } else if (role == 10) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k & 7]) : "v"(a), "v"(b));
      }
    }
  } else if (role == 11) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f32x4 x;
        asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(lane * 16));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        v[i & 7] += x[0];
      }
    }
  } else if (role == 12) {
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          f32x4 x;
          asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(lane * 16 + 1024 * k));
          v[k & 7] = x[1];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
  t = run<0>(1, 0, out);  printf("1 wave/SIMD  f32 16x16x4 only          : %7.3f ms = %6.1f cyc/MFMA @2.4GHz\n", t, cyc(t));
  t = run<0>(7, 0, out);  printf("1 wave/SIMD  bf16 16x16x16 only        : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<0>(7, 7, out);  printf("2 waves/SIMD bf16 16x16x16 | same      : %7.3f ms = %6.1f cyc per pair\n", t, cyc(t));
  t = run<0>(8, 0, out);  printf("1 wave/SIMD  bf16 16x16x32 only        : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<0>(8, 8, out);  printf("2 waves/SIMD bf16 16x16x32 | same      : %7.3f ms = %6.1f cyc per pair\n", t, cyc(t));
  t = run<0>(2, 0, out);  printf("1 wave/SIMD  VALU only (32 fma/iter)   : %7.3f ms = %6.1f cyc per 8 v_fma\n", t, cyc(t));
  t = run<0>(1, 2, out);  printf("2 waves/SIMD f32 MFMA | VALU           : %7.3f ms (sum if serial, max if overlapped)\n", t);
  t = run<0>(7, 2, out);  printf("2 waves/SIMD bf16 x16 | VALU           : %7.3f ms (sum if serial, max if overlapped)\n", t);
  t = run<0>(8, 2, out);  printf("2 waves/SIMD bf16 x32 | VALU           : %7.3f ms (sum if serial, max if overlapped)\n", t);
  t = run<1>(9, 0, out);  printf("1 wave/SIMD  bf16 x16 + 1 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<2>(9, 0, out);  printf("1 wave/SIMD  bf16 x16 + 2 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<4>(9, 0, out);  printf("1 wave/SIMD  bf16 x16 + 4 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<2>(10, 0, out); printf("1 wave/SIMD  bf16 x32 + 2 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<4>(10, 0, out); printf("1 wave/SIMD  bf16 x32 + 4 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<8>(10, 0, out); printf("1 wave/SIMD  bf16 x32 + 8 v_fma each   : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<4>(10, 10, out);printf("2 waves/SIMD both bf16 x32 + 4 v_fma   : %7.3f ms = %6.1f cyc per MFMA pair\n", t, cyc(t));
  t = run<0>(11, 0, out); printf("1 wave/SIMD  ds_read_b128 only (16/it) : %7.3f ms = %6.1f cyc per 4 reads\n", t, cyc(t));
  t = run<0>(8, 11, out); printf("2 waves/SIMD bf16 x32 | ds_read_b128   : %7.3f ms\n", t);
  t = run<0>(1, 11, out); printf("2 waves/SIMD f32 MFMA | ds_read_b128   : %7.3f ms\n", t);
  t = run<1>(12, 0, out); printf("1 wave/SIMD  bf16 x32 + 1 ds_read_b128 : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  t = run<2>(12, 0, out); printf("1 wave/SIMD  bf16 x32 + 2 ds_read_b128 : %7.3f ms = %6.1f cyc/MFMA\n", t, cyc(t));
  return 0;
}
