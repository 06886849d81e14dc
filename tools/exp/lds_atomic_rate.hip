// LDS atomic throughput on gfx950: cycles per wave-instruction of ds_add_f32 / ds_add_u32 / ds_add_u64 / plain read-modify-write,
// conflict-free addresses (lane i -> word i + 64 * step), one workgroup of 256 threads per CU.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* out, int iters, float v) {
  __shared__ float ldsf[8192];
  unsigned* ldsu = reinterpret_cast<unsigned*>(ldsf);
  unsigned long long* ldsq = reinterpret_cast<unsigned long long*>(ldsf);
  double* ldsd = reinterpret_cast<double*>(ldsf);
  for (int i = threadIdx.x; i < 8192; i += 256) ldsf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int idx = (lane + 256 * s + it * 7) & 4095;
      if (MODE == 0) __hip_atomic_fetch_add(ldsf + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 1) __hip_atomic_fetch_add(ldsu + idx, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 2) __hip_atomic_fetch_add(ldsq + idx, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 3) { ldsf[idx] = ldsf[idx] + v; }
      else if (MODE == 4) { ldsf[idx] = v + s; }
      else if (MODE == 5) __hip_atomic_fetch_add(ldsd + idx, static_cast<double>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (MODE == 6) { const float old = __hip_atomic_fetch_add(ldsf + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); v += old * 1e-30f; }
      else if (MODE == 7) __hip_atomic_fetch_max(reinterpret_cast<int*>(ldsu) + idx, static_cast<int>(v) + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (ldsf[lane] == 12345.f) out[0] = 0;
}

int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8);
  unsigned long long h[256];
  const int iters = 200;
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "read+add+write f32", "ds_write_b32", "ds_add_f64", "ds_add_rtn_f32", "ds_max_i32"};
  for (int mode = 0; mode < 8; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, d, iters, 1.5f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i];
    c /= 256;
    // 4 waves x iters x 16 instructions per workgroup share one LDS
    printf("%-22s %8.1f cycles per wave-instruction (4 waves issuing: %8.1f per instruction per CU)\n", names[mode], c / (iters * 16.0), c / (iters * 16.0 * 4));
  }
  return 0;
}
