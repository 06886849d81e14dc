// Roofline calibration streams (measurement support, not on the product path): a float4 fill
// (write-only), a float4 copy (1:1 read/write) and a float4 read-reduce (read-only).  bench.py
// times them next to the hot-path kernels so that the achieved GB/s of a write-dominated kernel
// can be placed against what this box actually sustains for that traffic mix.
#include "ts_common.hpp"

namespace {

__global__ void __launch_bounds__(256) calib_fill(float4* __restrict__ dst, size_t n4, float v) {
  const float4 val = make_float4(v, v, v, v);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] = val;
}

__global__ void __launch_bounds__(256) calib_copy(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
       i += static_cast<size_t>(gridDim.x) * blockDim.x)
    dst[i] = src[i];
}

__global__ void __launch_bounds__(256) calib_read(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
  float acc = 0.f;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
       i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1.2345e38f) *sink = acc;   // never true for finite data; keeps the loads alive
}

}  // namespace

extern "C" int ts_calib_stream(int kind, void* dst, const void* src, size_t nbytes, void* stream) {
  TS_REQUIRE(kind >= 0 && kind <= 2, TS_ERR_SHAPE, "calib: kind must be 0 (fill), 1 (copy) or 2 (read)");
  TS_REQUIRE(nbytes % 16 == 0 && nbytes > 0, TS_ERR_SHAPE, "calib: nbytes must be a positive multiple of 16");
  TS_REQUIRE_PTR(dst);
  if (kind != 0) TS_REQUIRE_PTR(src);
  const size_t n4 = nbytes / 16;
  size_t blocks = (n4 + 255) / 256;
  if (blocks > static_cast<size_t>(ts::kNumCU) * 8) blocks = static_cast<size_t>(ts::kNumCU) * 8;
  hipStream_t st = ts::as_stream(stream);
  const dim3 grid(static_cast<unsigned>(blocks));
  if (kind == 0) hipLaunchKernelGGL(calib_fill, grid, dim3(256), 0, st, reinterpret_cast<float4*>(dst), n4, 1.0f);
  else if (kind == 1) hipLaunchKernelGGL(calib_copy, grid, dim3(256), 0, st, reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), n4);
  else hipLaunchKernelGGL(calib_read, grid, dim3(256), 0, st, reinterpret_cast<const float4*>(src), n4, reinterpret_cast<float*>(dst));
  return ts::launched("calib_stream");
}
