// experiment (round 4): which store ORDER lets a [B][2C][D][H][W] volume be written at the fill rate?  Stores only, no loads.
//   kind 0: dense_warp_kernel's order -- workgroup (2 rows, 8-channel group, b): per row, 256 threads walk (d, x4) items, 16 planes each
//   kind 1: one workgroup per (b, plane, d-chunk): a purely linear stream of `len` float4
//   kind 2: workgroup (TR rows, group, b, d-chunk): per d, per plane, the TR-row run (TR * W * 4 bytes contiguous) written by consecutive lanes
#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }

__global__ void __launch_bounds__(256) pat0(float* out, int C, int D, int H, int W) {
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const unsigned HW = H * W, dHW = D * HW;
  const __amdgpu_buffer_rsrc_t r = rs(out + (size_t)b * 2 * C * dHW, 2u * C * dHW * 4u);
  u32x4 v; v.x = v.y = v.z = v.w = 0x3f800000u;
  const int nbx = W / 4, nitems = nbx * D;
  for (int rr = 0; rr < 2; ++rr) {
    const int y = by * 2 + rr;
    for (int item = threadIdx.x; item < nitems; item += 256) {
      const int d = item / nbx, x4 = (item - d * nbx) * 4;
      const unsigned off = (d * HW + y * W + x4) * 4u;
      for (int c = 0; c < 8; ++c) {
        const unsigned plane = (g * 8 + c) * dHW * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(v, r, off, plane, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, off, plane + C * dHW * 4u, 0);
      }
    }
  }
}
__global__ void __launch_bounds__(256) pat1(float4* out, size_t per) {      // blockIdx.x = chunk
  float4* p = out + per * blockIdx.x;
  const float4 v = make_float4(1.f, 1.f, 1.f, 1.f);
  for (size_t i = threadIdx.x; i < per; i += 256) p[i] = v;
}
template <int TR>
__global__ void __launch_bounds__(256) pat2(float* out, int C, int D, int H, int W, int dchunk, int planes_inner) {
  const int by = blockIdx.x, g = blockIdx.y % (C / 8), dc = blockIdx.y / (C / 8), b = blockIdx.z;
  const unsigned HW = H * W, dHW = D * HW;
  const __amdgpu_buffer_rsrc_t r = rs(out + (size_t)b * 2 * C * dHW, 2u * C * dHW * 4u);
  u32x4 v; v.x = v.y = v.z = v.w = 0x3f800000u;
  const int run4 = TR * W / 4;                       // float4 per (plane, d) run
  const int y0 = by * TR;
  if (planes_inner) {
    for (int d = dc * dchunk; d < min(D, (dc + 1) * dchunk); ++d)
      for (int i = threadIdx.x; i < run4; i += 256) {
        const unsigned off = (d * HW + y0 * W) * 4u + i * 16u;
        for (int c = 0; c < 16; ++c) {
          const unsigned plane = (c < 8 ? g * 8 + c : C + g * 8 + c - 8) * dHW * 4u;
          __builtin_amdgcn_raw_buffer_store_b128(v, r, off, plane, 0);
        }
      }
  } else {
    for (int c = 0; c < 16; ++c) {
      const unsigned plane = (c < 8 ? g * 8 + c : C + g * 8 + c - 8) * dHW * 4u;
      for (int d = dc * dchunk; d < min(D, (dc + 1) * dchunk); ++d)
        for (int i = threadIdx.x; i < run4; i += 256)
          __builtin_amdgcn_raw_buffer_store_b128(v, r, (d * HW + y0 * W) * 4u + i * 16u, plane, 0);
    }
  }
}
// K1's store pattern (block_cost_fast): workgroup = (4 rows, 8-channel group, b), lane = (candidate d, 4-pixel block); 16 planes + 1
// ORDER 0: rows outer, planes inner (what block_cost_fast does); 1: planes outer, rows inner; 2: two passes of 2 rows, planes inner
template <int ORDER>
__global__ void __launch_bounds__(320) pat_k1(float* out, int C, int D, int H, int W) {
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int nbx = W / 4;
  const int tid = threadIdx.x;
  const int d = tid / 64, bx = tid % 64;
  if (d >= D || bx >= nbx) return;
  const unsigned HW = H * W, dHW = D * HW;
  const __amdgpu_buffer_rsrc_t r = rs(out + (size_t)b * 2 * C * dHW, 2u * C * dHW * 4u);
  u32x4 v; v.x = v.y = v.z = v.w = 0x3f800000u;
  const unsigned base = d * HW + by * 4 * W + bx * 4;
  if (ORDER == 0) {
    for (int rr = 0; rr < 4; ++rr)
      for (int c = 0; c < 16; ++c) {
        const unsigned plane = (c < 8 ? (g * 8 + c) : (C + g * 8 + c - 8)) * dHW;
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + rr * W) * 4u, plane * 4u, 0);
      }
  } else if (ORDER == 1) {
    for (int c = 0; c < 16; ++c) {
      const unsigned plane = (c < 8 ? (g * 8 + c) : (C + g * 8 + c - 8)) * dHW;
      for (int rr = 0; rr < 4; ++rr) __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + rr * W) * 4u, plane * 4u, 0);
    }
  } else if (ORDER == 2) {      // row pairs; per pair, per plane: the two rows back to back
    for (int rp = 0; rp < 2; ++rp)
      for (int c = 0; c < 16; ++c) {
        const unsigned plane = (c < 8 ? (g * 8 + c) : (C + g * 8 + c - 8)) * dHW;
        for (int q = 0; q < 2; ++q) __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + (2 * rp + q) * W) * 4u, plane * 4u, 0);
      }
  } else {                      // row pairs; per pair, per block of 4 planes: row a's four planes, then row b's four planes
    for (int rp = 0; rp < 2; ++rp)
      for (int h = 0; h < 4; ++h)
        for (int q = 0; q < 2; ++q)
          for (int cc = 0; cc < 4; ++cc) {
            const int c = h * 4 + cc;
            const unsigned plane = (c < 8 ? (g * 8 + c) : (C + g * 8 + c - 8)) * dHW;
            __builtin_amdgcn_raw_buffer_store_b128(v, r, (base + (2 * rp + q) * W) * 4u, plane * 4u, 0);
          }
  }
}
extern "C" int run_k1(int order, void* out, int B, int C, int D, int H, int W, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(H / 4, C / 8, B);
  if (order == 0) hipLaunchKernelGGL(pat_k1<0>, grid, dim3(320), 0, st, reinterpret_cast<float*>(out), C, D, H, W);
  else if (order == 1) hipLaunchKernelGGL(pat_k1<1>, grid, dim3(320), 0, st, reinterpret_cast<float*>(out), C, D, H, W);
  else if (order == 2) hipLaunchKernelGGL(pat_k1<2>, grid, dim3(320), 0, st, reinterpret_cast<float*>(out), C, D, H, W);
  else hipLaunchKernelGGL(pat_k1<3>, grid, dim3(320), 0, st, reinterpret_cast<float*>(out), C, D, H, W);
  return static_cast<int>(hipGetLastError());
}
extern "C" int run_pat(int kind, int tr, int dchunk, int planes_inner, void* out, int B, int C, int D, int H, int W, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* o = reinterpret_cast<float*>(out);
  const int dcs = (D + dchunk - 1) / dchunk;
  if (kind == 0) hipLaunchKernelGGL(pat0, dim3(H / 2, C / 8, B), dim3(256), 0, st, o, C, D, H, W);
  else if (kind == 1) {
    const size_t total4 = (size_t)B * 2 * C * D * H * W / 4;
    const int chunks = tr;                          // number of workgroups
    hipLaunchKernelGGL(pat1, dim3(chunks), dim3(256), 0, st, reinterpret_cast<float4*>(out), total4 / chunks);
  } else if (tr == 2) hipLaunchKernelGGL(pat2<2>, dim3(H / 2, C / 8 * dcs, B), dim3(256), 0, st, o, C, D, H, W, dchunk, planes_inner);
  else if (tr == 4) hipLaunchKernelGGL(pat2<4>, dim3(H / 4, C / 8 * dcs, B), dim3(256), 0, st, o, C, D, H, W, dchunk, planes_inner);
  else if (tr == 8) hipLaunchKernelGGL(pat2<8>, dim3(H / 8, C / 8 * dcs, B), dim3(256), 0, st, o, C, D, H, W, dchunk, planes_inner);
  else if (tr == 136) hipLaunchKernelGGL(pat2<136>, dim3(1, C / 8 * dcs, B), dim3(256), 0, st, o, C, D, H, W, dchunk, planes_inner);
  return static_cast<int>(hipGetLastError());
}
