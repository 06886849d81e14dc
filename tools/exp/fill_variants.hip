// experiment: what write rate can a fill reach with different store flavours / grid sizes?
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX>
__global__ void __launch_bounds__(256) fill_buf(float* dst, size_t n4) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0xffffffffu, 0x00020000);
  u32x4 v; v.x = v.y = v.z = v.w = 0x3f800000u;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    __builtin_amdgcn_raw_buffer_store_b128(v, r, static_cast<unsigned>(i * 16), 0, AUX);
}
__global__ void __launch_bounds__(256) fill_plain(float4* dst, size_t n4) {
  const float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * blockDim.x) dst[i] = v;
}
__global__ void __launch_bounds__(256) fill_nt(float4* dst, size_t n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = {1.f, 1.f, 1.f, 1.f};
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst) + i);
}
// each workgroup owns a contiguous chunk (instead of the grid-stride interleave)
__global__ void __launch_bounds__(256) fill_chunk(float4* dst, size_t n4) {
  const float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[i] = v;
}
// K1's store pattern without any of its loads / LDS / ALU: workgroup = (4 rows, 8-channel group), lane = (candidate d,
// 4-pixel block); ORDER 0: rows outer, planes inner (what block_cost_fast does); ORDER 1: planes outer, rows inner.
template <int ORDER>
__global__ void __launch_bounds__(320) fill_k1(float* out, int C, int D, int H, int W) {
  const int by = blockIdx.x, g = blockIdx.y;
  const int nbx = W / 4;
  const int tid = threadIdx.x;
  const int d = tid / 64, bx = tid % 64;
  if (d >= D || bx >= nbx) return;
  const unsigned HW = H * W, dHW = D * HW;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xffffffffu, 0x00020000);
  u32x4 v; v.x = v.y = v.z = v.w = 0x3f800000u;
  const unsigned base = d * HW + by * 4 * W + bx * 4;
  if (ORDER == 0) {
    for (int rr = 0; rr < 4; ++rr)
      for (int c = 0; c < 8; ++c) {
        const unsigned plane = (g * 8 + c) * dHW;
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + rr * W) * 4u, plane * 4u, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + rr * W) * 4u, (plane + C * dHW) * 4u, 0);
      }
  } else {
    for (int c = 0; c < 16; ++c) {
      const unsigned plane = (c < 8 ? (g * 8 + c) : (C + g * 8 + c - 8)) * dHW;
      for (int rr = 0; rr < 4; ++rr) __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + rr * W) * 4u, plane * 4u, 0);
    }
  }
}
extern "C" int fill_k1_run(int order, void* dst, int C, int D, int H, int W, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(H / 4, C / 8);
  if (order == 0) hipLaunchKernelGGL(fill_k1<0>, grid, dim3(320), 0, st, reinterpret_cast<float*>(dst), C, D, H, W);
  else hipLaunchKernelGGL(fill_k1<1>, grid, dim3(320), 0, st, reinterpret_cast<float*>(dst), C, D, H, W);
  return static_cast<int>(hipGetLastError());
}
extern "C" int fill_run(int kind, void* dst, size_t nbytes, int blocks, void* stream) {
  const size_t n4 = nbytes / 16;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  switch (kind) {
    case 0: hipLaunchKernelGGL(fill_plain, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float4*>(dst), n4); break;
    case 1: hipLaunchKernelGGL(fill_nt, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float4*>(dst), n4); break;
    case 2: hipLaunchKernelGGL(fill_buf<0>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float*>(dst), n4); break;
    case 3: hipLaunchKernelGGL(fill_buf<2>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float*>(dst), n4); break;
    case 4: hipLaunchKernelGGL(fill_buf<17>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float*>(dst), n4); break;
    case 5: hipLaunchKernelGGL(fill_buf<1>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float*>(dst), n4); break;
    case 6: hipLaunchKernelGGL(fill_chunk, dim3(blocks), dim3(256), 0, st, reinterpret_cast<float4*>(dst), n4); break;
  }
  return static_cast<int>(hipGetLastError());
}
