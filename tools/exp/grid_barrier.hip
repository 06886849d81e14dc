// Cost of a hand-rolled grid barrier between dependent "ops" inside one persistent kernel, against the launch-to-launch
// latency of separate kernels.  Each op: every workgroup writes a slice of a small buffer, the next op reads what OTHER
// workgroups wrote (so the barrier must really publish data across XCDs).
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);                       // agent scope: write back what this workgroup produced
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256)
persistent_ops(float* a, float* b, int n, int nops, unsigned* counter) {
  const int G = gridDim.x;
  for (int op = 0; op < nops; ++op) {
    const float* src = (op & 1) ? b : a;
    float* dst = (op & 1) ? a : b;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += G * blockDim.x) {
      const int j = (i + 4099) % n;                                // someone else's element
      dst[i] = src[j] * 0.5f + 1.f;
    }
    grid_barrier(counter, static_cast<unsigned>((op + 1) * G));
  }
}

__global__ void __launch_bounds__(256)
one_op(const float* src, float* dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = (i + 4099) % n;
    dst[i] = src[j] * 0.5f + 1.f;
  }
}

extern "C" int run_persistent(float* a, float* b, int n, int nops, int blocks, unsigned* counter, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipMemsetAsync(counter, 0, 4, st);
  hipLaunchKernelGGL(persistent_ops, dim3(blocks), dim3(256), 0, st, a, b, n, nops, counter);
  return static_cast<int>(hipGetLastError());
}

extern "C" int run_separate(float* a, float* b, int n, int nops, int blocks, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int op = 0; op < nops; ++op)
    hipLaunchKernelGGL(one_op, dim3(blocks), dim3(256), 0, st, (op & 1) ? b : a, (op & 1) ? a : b, n);
  return static_cast<int>(hipGetLastError());
}
