// LDS atomic rate on gfx950: cycles per wave instruction of ds_add_f32 / ds_add_u32 / plain ds_write / read-add-write,
// conflict-free and with 2-way / 16-way same-address collisions.  hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND, int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx;
  if (PATTERN == 0) idx = lane;                 // conflict-free, distinct banks
  else if (PATTERN == 1) idx = lane >> 1;       // pairs of lanes on one address
  else if (PATTERN == 2) idx = lane >> 4;       // 16 lanes on one address
  else if (PATTERN == 3) idx = (lane * 4) & 63 | (lane >> 4);   // distinct addresses, 4 apart: bank conflicts (2-way on 32 banks... stride 4 -> 8 banks)
  else idx = (lane * 5) & 63;                   // permuted distinct banks
  float* p = buf + wave * 1024 + idx;
  unsigned* pu = reinterpret_cast<unsigned*>(p);
  const float v = 1.0f + lane * 1e-3f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float* q = p + u * 64;
      if (KIND == 0) __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (KIND == 1) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(q), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (KIND == 2) *reinterpret_cast<volatile float*>(q) = v;
      else if (KIND == 3) { volatile float* vq = q; *vq = *vq + v; }
      else if (KIND == 4) {               // float add as an integer compare-and-swap loop
        unsigned* qu = reinterpret_cast<unsigned*>(q);
        unsigned old = *reinterpret_cast<volatile unsigned*>(qu), assumed;
        do {
          assumed = old;
          old = atomicCAS(qu, assumed, __float_as_uint(__uint_as_float(assumed) + v));
        } while (old != assumed);
      } else {                            // 64-bit fixed point (two floats' worth of LDS per value)
        unsigned long long* q8 = reinterpret_cast<unsigned long long*>(buf + wave * 1024) + idx + u * 32;
        __hip_atomic_fetch_add(q8, static_cast<unsigned long long>(static_cast<long long>(v * 4294967296.f)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = static_cast<float>(t1 - t0) / (iters * 8.0f) + buf[5] * 0.f;
}

template <int KIND, int PATTERN>
void run(const char* name, int blocks_per_cu) {
  const int nb = 256 * blocks_per_cu, iters = 2000;
  float* d;
  hipMalloc(&d, nb * sizeof(float));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<KIND, PATTERN>), dim3(nb), dim3(256), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL((k<KIND, PATTERN>), dim3(nb), dim3(256), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<float> h(nb);
  hipMemcpy(h.data(), d, nb * sizeof(float), hipMemcpyDeviceToHost);
  // wave instructions per CU: blocks_per_cu * 4 waves * iters * 8
  const double per_cu = double(blocks_per_cu) * 4 * iters * 8;
  printf("%-34s wg/CU %d: %7.1f clk/instr seen by a wave, %6.2f ns per wave-instr per CU (%.3f ms)\n", name, blocks_per_cu, h[0], ms * 1e6 / per_cu, ms);
  hipFree(d);
}

int main() {
  for (int bpc : {1, 4}) {
    run<0, 0>("ds_add_f32 conflict-free", bpc);
    run<0, 4>("ds_add_f32 permuted banks", bpc);
    run<0, 1>("ds_add_f32 2 lanes/address", bpc);
    run<0, 2>("ds_add_f32 16 lanes/address", bpc);
    run<0, 3>("ds_add_f32 stride-4 banks", bpc);
    run<1, 0>("ds_add_u32 conflict-free", bpc);
    run<1, 1>("ds_add_u32 2 lanes/address", bpc);
    run<2, 0>("ds_write_b32 conflict-free", bpc);
    run<3, 0>("read+add+write conflict-free", bpc);
    run<4, 0>("CAS-loop f32 add conflict-free", bpc);
    run<4, 1>("CAS-loop f32 add 2 lanes/address", bpc);
    run<4, 2>("CAS-loop f32 add 16 lanes/address", bpc);
    run<5, 0>("ds_add_u64 conflict-free", bpc);
    run<5, 1>("ds_add_u64 2 lanes/address", bpc);
  }
  return 0;
}
