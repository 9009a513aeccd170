// Micro-benchmark: cycles per ds_add_f32 wave-instruction on gfx950 for the address patterns a fused dG2
// accumulator would see.  hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

constexpr int ACC = 140 * 80;
constexpr int ITERS = 2000;

template <int MODE>
__global__ __launch_bounds__(512) void k(const int* __restrict__ i2tab, float* out, long long* cyc) {
  __shared__ float acc[ACC];
  for (int i = threadIdx.x; i < ACC; i += 512) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 4, lo = lane & 15;
  const int* tab = i2tab + (blockIdx.x * 8 + wave) * 16;
  long long t0 = clock64();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) {
      const int col = 16 * nt + lo;           // id * 5 + kk
      const int id = col / 5, kk = col % 5;
      const int i2 = tab[id] ;
      const int i2r = (i2 + it) % 140;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int addr;
        if (MODE == 0) addr = (lane + 64 * (nt * 4 + r) + it * 7) % ACC;              // conflict-free
        else if (MODE == 1) addr = i2r * 80 + kk * 16 + 4 * hi + r;                    // [kk][c2] as the accumulators stand
        else if (MODE == 2) addr = i2r * 80 + (4 * hi + r) * 5 + kk;                   // [c2][kk]
        else addr = i2r * 81 + kk * 16 + ((4 * hi + r) ^ (kk * 4)) ;                   // padded row + xor swizzle
        atomicAdd(&acc[addr], 1.0f);
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < ACC; i += 512) s += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const int* tab, float* out, long long* cyc, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, tab, out, cyc);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, tab, out, cyc);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_cu = 8.0 * ITERS * 20;   // wave-instructions per CU
  printf("%-28s %8.3f ms  -> %.1f ns per wave-instr per CU = %.1f cycles @2.4GHz\n", name, ms, ms * 1e6 / instr_per_cu,
         ms * 1e6 / instr_per_cu * 2.4);
}

int main() {
  std::vector<int> h(256 * 8 * 16);
  srand(1);
  for (size_t w = 0; w < h.size() / 16; ++w) {   // 16 distinct i2 per chunk
    int used[140] = {0};
    for (int b = 0; b < 16; ++b) { int v; do v = rand() % 140; while (used[v]); used[v] = 1; h[w * 16 + b] = v; }
  }
  int* tab; float* out; long long* cyc;
  hipMalloc(&tab, h.size() * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(tab, out, cyc, "conflict-free");
  run<1>(tab, out, cyc, "[kk][c2] accumulator order");
  run<2>(tab, out, cyc, "[c2][kk]");
  run<3>(tab, out, cyc, "padded + swizzle");
  return 0;
}
