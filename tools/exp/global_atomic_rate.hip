// Global-memory atomic throughput on gfx950 (the L2 does them): float / double / u32 / u64 adds, no return value, (a) every lane its own
// address, a 64 MB footprint; (b) 16 workgroups piling onto the same 256 addresses (the contention of K1's gDisp: one contribution per
// channel group).  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void __launch_bounds__(256) spread(T* p, size_t n, int iters) {
  size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  for (int it = 0; it < iters; ++it, i += static_cast<size_t>(gridDim.x) * 256) atomicAdd(p + (i % n), static_cast<T>(1));
}
template <typename T>
__global__ void __launch_bounds__(256) piled(T* p, int iters) {
  const size_t base = static_cast<size_t>(blockIdx.x / 16) * 256 * iters;        // 16 consecutive workgroups share their addresses
  for (int it = 0; it < iters; ++it) atomicAdd(p + base + static_cast<size_t>(it) * 256 + threadIdx.x, static_cast<T>(1));
}
template <typename T>
void run(const char* name) {
  const size_t n = 16u << 20;
  T* d; (void)hipMalloc(&d, n * sizeof(T)); (void)hipMemset(d, 0, n * sizeof(T));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 4096, iters = 16;
  float ms[2];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(spread<T>, dim3(blocks), dim3(256), 0, 0, d, n, iters);
      else hipLaunchKernelGGL(piled<T>, dim3(blocks), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms[mode], e0, e1);
    }
  }
  const double ops = static_cast<double>(blocks) * 256 * iters;
  printf("%-8s  own addresses %7.1f G atomics/s   16 workgroups per address %7.1f G atomics/s\n", name, ops / ms[0] / 1e6, ops / ms[1] / 1e6);
  (void)hipFree(d);
}
int main() {
  run<float>("float"); run<double>("double"); run<unsigned>("u32"); run<unsigned long long>("u64");
  return 0;
}
