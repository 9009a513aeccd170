// Micro-benchmark: how fast can a kernel store (load) 409 600 rows of 100 floats (400 B, 16-byte aligned only) of a
// [N, 100] tensor in RANDOM row order -- the output (d_output) access of the TT chain kernels -- and does the shape
// of a wave-instruction matter?
//   A: 4 lanes per row, 16 rows per instruction, a row's 25 pieces spread over 7 instructions (the chain kernels)
//   B: 25 lanes per row, 2 rows per instruction (50 lanes busy): one instruction covers 400 contiguous bytes per row
//   C: 5 lanes x 5 instructions per row, 12 rows per instruction: 80 contiguous bytes per row and instruction
// each as plain and as nt (streaming) accesses; "seq" = the same kernels with rows in order (the upper bound).
// hipcc -O3 --offload-arch=gfx950 rowstream.hip -o rowstream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>

constexpr int N = 409600, D = 100, D4 = D / 4;
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void st(float* p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
  else *reinterpret_cast<f4*>(p) = v;
}
template <bool NT>
__device__ __forceinline__ f4 ld(const float* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  return *reinterpret_cast<const f4*>(p);
}

// MODE 0 = A, 1 = B, 2 = C;  STORE: write rows / read rows
template <int MODE, bool NT, bool STORE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ rows, float* __restrict__ buf, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const f4 val = {1.f * lane, 2.f, 3.f, 4.f};
  if (MODE == 0) {
    const int b = lane >> 2, j = lane & 3;
    for (int base = gw * 16; base < N; base += nw * 16) {
      const int r = rows[base + b];
      float* row = buf + (size_t)r * D;
#pragma unroll
      for (int kk = 0; kk < 7; ++kk) {
        const int pc = 4 * kk + j;
        if (pc < D4) {
          if (STORE) st<NT>(row + 4 * pc, val);
          else acc += ld<NT>(row + 4 * pc);
        }
      }
    }
  } else if (MODE == 1) {
    const int b = lane / 25, pc = lane % 25;
    for (int base = gw * 16; base < N; base += nw * 16) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (b < 2) {
          const int r = rows[base + 2 * kk + b];
          float* row = buf + (size_t)r * D;
          if (STORE) st<NT>(row + 4 * pc, val);
          else acc += ld<NT>(row + 4 * pc);
        }
      }
    }
  } else {
    const int b = lane / 5, j = lane % 5;
    for (int base = gw * 12; base < N; base += nw * 12) {
      if (b < 12 && base + b < N) {
        const int r = rows[base + b];
        float* row = buf + (size_t)r * D;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const int pc = 5 * kk + j;
          if (STORE) st<NT>(row + 4 * pc, val);
          else acc += ld<NT>(row + 4 * pc);
        }
      }
    }
  }
  if (!STORE && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, bool NT, bool STORE>
static void run(const char* name, const int* rows, float* buf, float* sink, int wgs) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, NT, STORE>), dim3(wgs), dim3(256), 0, 0, rows, buf, sink);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE, NT, STORE>), dim3(wgs), dim3(256), 0, 0, rows, buf, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms * 100.0;
  printf("%-44s wgs %4d: %7.1f us  %6.2f TB/s of row bytes\n", name, wgs, us, (double)N * D * 4 / (us * 1e-6) / 1e12);
}

int main() {
  std::vector<int> perm(N), seq(N);
  std::iota(seq.begin(), seq.end(), 0);
  perm = seq;
  srand(7);
  for (int i = N - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);
  int *d_perm, *d_seq;
  float *buf, *sink;
  hipMalloc(&d_perm, N * 4);
  hipMalloc(&d_seq, N * 4);
  hipMalloc(&buf, (size_t)N * D * 4 + 4096);
  hipMalloc(&sink, 16);
  hipMemcpy(d_perm, perm.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_seq, seq.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(buf, 0, (size_t)N * D * 4);
  for (int wgs : {512, 1024, 2048}) {
    run<0, false, true>("store A (4 lanes/row x 7)      random", d_perm, buf, sink, wgs);
    run<0, true, true>("store A nt                      random", d_perm, buf, sink, wgs);
    run<1, false, true>("store B (25 lanes/row)          random", d_perm, buf, sink, wgs);
    run<1, true, true>("store B nt                      random", d_perm, buf, sink, wgs);
    run<2, false, true>("store C (5 lanes x 5)           random", d_perm, buf, sink, wgs);
    run<2, true, true>("store C nt                      random", d_perm, buf, sink, wgs);
    run<0, false, true>("store A                         seq", d_seq, buf, sink, wgs);
    run<1, false, true>("store B                         seq", d_seq, buf, sink, wgs);
    run<0, false, false>("load  A (4 lanes/row x 7)      random", d_perm, buf, sink, wgs);
    run<0, true, false>("load  A nt                      random", d_perm, buf, sink, wgs);
    run<1, false, false>("load  B (25 lanes/row)          random", d_perm, buf, sink, wgs);
    run<1, true, false>("load  B nt                      random", d_perm, buf, sink, wgs);
    run<2, false, false>("load  C (5 lanes x 5)           random", d_perm, buf, sink, wgs);
    run<0, false, false>("load  A                         seq", d_seq, buf, sink, wgs);
  }
  return 0;
}
