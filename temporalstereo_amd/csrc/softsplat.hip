// K2 -- temporal cost warp for gfx950 (MI355X): rigid re-projection + forward (soft-max) splatting.
//
//   K2a  ts_project_to_3d_fwd  replaces project_to_3d()  architecture/modeling/layers/inverse_warp.py:92-178
//        (the triangular_depth / optical_flow / flow_mask products used by update_map).
//   K2b  ts_softsplat_*        replaces the three cupy/NVRTC kernels of
//        architecture/modeling/layers/softsplat.py:8-177 (updateOutput / updateGradInput /
//        updateGradFlow) and, for the always-detached inference use of the project
//        (projects/TemporalStereo/TemporalStereo.py:388-389,415-419), a fused
//        softmax-splat (exp(metric) weighting + scatter + normalisation, softsplat.py:334-360).
//
// The tensors are tiny (B x <=5 x 68 x 120 at config 3), so these kernels are latency-bound; the
// design goal is ONE launch per op with no per-shape recompilation (the reference bakes sizes and
// strides into NVRTC source, softsplat.py:179-232).  Scatter uses fp32 hardware atomics
// (global_atomic_add_f32), like the reference's atomicAdd: summation order is not deterministic.
#include "ts_common.hpp"

namespace {

struct Taps {
  int x0, y0;
  float nw, ne, sw, se;
};

// bilinear footprint of a source pixel pushed to (x+fx, y+fy); softsplat.py:19-35
__device__ __forceinline__ Taps taps_of(float ox, float oy) {
  Taps t;
  const float fx0 = floorf(ox), fy0 = floorf(oy);
  // keep the int conversion defined for absurd flows; such taps are out of frame anyway
  t.x0 = static_cast<int>(fminf(fmaxf(fx0, -2.f), 1.0e9f));
  t.y0 = static_cast<int>(fminf(fmaxf(fy0, -2.f), 1.0e9f));
  const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
  t.nw = (x1 - ox) * (y1 - oy);
  t.ne = (ox - fx0) * (y1 - oy);
  t.sw = (x1 - ox) * (oy - fy0);
  t.se = (ox - fx0) * (oy - fy0);
  return t;
}

__device__ __forceinline__ bool inside(int x, int y, int W, int H) {
  return (x >= 0) & (x < W) & (y >= 0) & (y < H);
}

// MODE 0: plain summation of `input` (C channels).
// MODE 1: softmax: channels c<C carry input*exp(metric), channel C carries exp(metric).
template <int MODE>
__global__ void __launch_bounds__(256)
splat_scatter(const float* __restrict__ input, const float* __restrict__ flow, const float* __restrict__ metric,
              float* __restrict__ accum, int B, int C, int H, int W) {
  const int HW = H * W;
  const int CO = (MODE == 1) ? C + 1 : C;
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * HW + p;
    const Taps t = taps_of(static_cast<float>(x) + fl[0], static_cast<float>(y) + fl[HW]);
    const bool inw = inside(t.x0, t.y0, W, H), ine = inside(t.x0 + 1, t.y0, W, H);
    const bool isw = inside(t.x0, t.y0 + 1, W, H), ise = inside(t.x0 + 1, t.y0 + 1, W, H);
    if (!(inw | ine | isw | ise)) continue;
    const float e = (MODE == 1) ? expf(metric[i]) : 1.f;
    float* ob = accum + static_cast<size_t>(b) * CO * HW;
    const int q = t.y0 * W + t.x0;
    for (int c = 0; c < CO; ++c) {
      float v;
      if (MODE == 1) v = (c < C) ? input[(static_cast<size_t>(b) * C + c) * HW + p] * e : e;
      else v = input[(static_cast<size_t>(b) * C + c) * HW + p];
      float* o = ob + static_cast<size_t>(c) * HW + q;
      if (inw) unsafeAtomicAdd(o, v * t.nw);
      if (ine) unsafeAtomicAdd(o + 1, v * t.ne);
      if (isw) unsafeAtomicAdd(o + W, v * t.sw);
      if (ise) unsafeAtomicAdd(o + W + 1, v * t.se);
    }
  }
}

// Deterministic summation splat: contributions are accumulated as 64-bit fixed-point integers (2^-40
// resolution) with integer atomics -- integer addition is associative, so the result does not depend on
// the order in which the hardware retires the atomics.
constexpr double kFix = 1099511627776.0;     // 2^40

__global__ void __launch_bounds__(256)
splat_scatter_fixed(const float* __restrict__ input, const float* __restrict__ flow, unsigned long long* __restrict__ accum,
                    int B, int C, int H, int W) {
  const int HW = H * W;
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * HW + p;
    const Taps t = taps_of(static_cast<float>(x) + fl[0], static_cast<float>(y) + fl[HW]);
    const bool inw = inside(t.x0, t.y0, W, H), ine = inside(t.x0 + 1, t.y0, W, H);
    const bool isw = inside(t.x0, t.y0 + 1, W, H), ise = inside(t.x0 + 1, t.y0 + 1, W, H);
    if (!(inw | ine | isw | ise)) continue;
    unsigned long long* ob = accum + static_cast<size_t>(b) * C * HW;
    const int q = t.y0 * W + t.x0;
    auto fix = [](float v) { return static_cast<unsigned long long>(__double2ll_rn(static_cast<double>(v) * kFix)); };
    for (int c = 0; c < C; ++c) {
      const float v = input[(static_cast<size_t>(b) * C + c) * HW + p];
      unsigned long long* o = ob + static_cast<size_t>(c) * HW + q;
      if (inw) atomicAdd(o, fix(v * t.nw));               // two's complement: unsigned wrap-around == signed sum
      if (ine) atomicAdd(o + 1, fix(v * t.ne));
      if (isw) atomicAdd(o + W, fix(v * t.sw));
      if (ise) atomicAdd(o + W + 1, fix(v * t.se));
    }
  }
}

__global__ void __launch_bounds__(256)
splat_from_fixed(const unsigned long long* __restrict__ accum, float* __restrict__ out, long long n) {
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    out[i] = static_cast<float>(static_cast<double>(static_cast<long long>(accum[i])) / kFix);
}

// out[c] = accum[c] / (accum[C] + 1e-22)   (softsplat.py:352-357)
__global__ void __launch_bounds__(256)
splat_normalize(const float* __restrict__ accum, float* __restrict__ out, int B, int C, int HW) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const float* a = accum + static_cast<size_t>(b) * (C + 1) * HW + p;
    const float den = a[static_cast<size_t>(C) * HW] + 1e-22f;
    for (int c = 0; c < C; ++c) out[(static_cast<size_t>(b) * C + c) * HW + p] = a[static_cast<size_t>(c) * HW] / den;
  }
}

// gradInput[n,c,y,x] = sum over the 4 taps of gradOutput * weight   (softsplat.py:63-105)
__global__ void __launch_bounds__(256)
splat_grad_input(const float* __restrict__ flow, const float* __restrict__ gout, float* __restrict__ gin,
                 int B, int C, int H, int W) {
  const int HW = H * W;
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * HW + p;
    const Taps t = taps_of(static_cast<float>(x) + fl[0], static_cast<float>(y) + fl[HW]);
    const bool inw = inside(t.x0, t.y0, W, H), ine = inside(t.x0 + 1, t.y0, W, H);
    const bool isw = inside(t.x0, t.y0 + 1, W, H), ise = inside(t.x0 + 1, t.y0 + 1, W, H);
    const int q = t.y0 * W + t.x0;
    for (int c = 0; c < C; ++c) {
      const float* go = gout + (static_cast<size_t>(b) * C + c) * HW + q;
      float g = 0.f;
      if (inw) g += go[0] * t.nw;
      if (ine) g += go[1] * t.ne;
      if (isw) g += go[W] * t.sw;
      if (ise) g += go[W + 1] * t.se;
      gin[(static_cast<size_t>(b) * C + c) * HW + p] = g;
    }
  }
}

// gradFlow[n,{x,y},y,x] = sum_c input * sum_taps gradOutput * d weight   (softsplat.py:116-176)
__global__ void __launch_bounds__(256)
splat_grad_flow(const float* __restrict__ input, const float* __restrict__ flow, const float* __restrict__ gout,
                float* __restrict__ gflow, int B, int C, int H, int W) {
  const int HW = H * W;
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const int y = p / W, x = p - y * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * HW + p;
    const float ox = static_cast<float>(x) + fl[0], oy = static_cast<float>(y) + fl[HW];
    const Taps t = taps_of(ox, oy);
    const float fx0 = floorf(ox), fy0 = floorf(oy);
    const float ax = ox - fx0, ay = oy - fy0;         // fractional parts
    const bool inw = inside(t.x0, t.y0, W, H), ine = inside(t.x0 + 1, t.y0, W, H);
    const bool isw = inside(t.x0, t.y0 + 1, W, H), ise = inside(t.x0 + 1, t.y0 + 1, W, H);
    const int q = t.y0 * W + t.x0;
    float gx = 0.f, gy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = input[(static_cast<size_t>(b) * C + c) * HW + p];
      const float* go = gout + (static_cast<size_t>(b) * C + c) * HW + q;
      const float a = inw ? go[0] : 0.f, bq = ine ? go[1] : 0.f, cq = isw ? go[W] : 0.f, dq = ise ? go[W + 1] : 0.f;
      // d/dox: nw -(1-ay), ne +(1-ay), sw -ay, se +ay ; d/doy: nw -(1-ax), ne -ax, sw +(1-ax), se +ax
      gx += v * ((bq - a) * (1.f - ay) + (dq - cq) * ay);
      gy += v * ((cq - a) * (1.f - ax) + (dq - bq) * ax);
    }
    gflow[static_cast<size_t>(b) * 2 * HW + p] = gx;
    gflow[static_cast<size_t>(b) * 2 * HW + HW + p] = gy;
  }
}

// ------------------------------------------------------------------------------------------ K2a
// per pixel and depth plane: X = invK * [u v 1]^T * depth; cam = (K*T)[:3] * [X 1]; inverse_warp.py:119-170
__global__ void __launch_bounds__(256)
project_kernel(const float* __restrict__ depth, const float* __restrict__ K, const float* __restrict__ invK,
               const float* __restrict__ T, float* __restrict__ tri, float* __restrict__ flowo,
               unsigned char* __restrict__ mask, int B, int C, int H, int W, int kdim, int ikdim, float eps) {
  __shared__ float P[3][4];
  __shared__ float iK[3][3];
  const int b = blockIdx.y;
  if (threadIdx.x < 12) {
    const int r = threadIdx.x / 4, c = threadIdx.x % 4;
    // new_K (4x4, identity-padded when K is 3x3) times T, first three rows (:138-146)
    float acc = 0.f;
    for (int k = 0; k < 4; ++k) {
      float kv;
      if (k < kdim && r < kdim) kv = K[(static_cast<size_t>(b) * kdim + r) * kdim + k];
      else kv = (r == k) ? 1.f : 0.f;
      acc += kv * T[(static_cast<size_t>(b) * 4 + k) * 4 + c];
    }
    P[r][c] = acc;
  } else if (threadIdx.x >= 16 && threadIdx.x < 25) {
    const int j = threadIdx.x - 16;
    iK[j / 3][j % 3] = invK[(static_cast<size_t>(b) * ikdim + j / 3) * ikdim + j % 3];
  }
  __syncthreads();
  const int HW = H * W;
  const int n = C * HW;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int c = i / HW, p = i - c * HW;
    const int y = p / W, x = p - y * W;
    const float u = static_cast<float>(x), v = static_cast<float>(y);
    const float z = depth[static_cast<size_t>(b) * n + i];
    const float X = (iK[0][0] * u + iK[0][1] * v + iK[0][2]) * z;
    const float Y = (iK[1][0] * u + iK[1][1] * v + iK[1][2]) * z;
    const float Z = (iK[2][0] * u + iK[2][1] * v + iK[2][2]) * z;
    const float cx = P[0][0] * X + P[0][1] * Y + P[0][2] * Z + P[0][3];
    const float cy = P[1][0] * X + P[1][1] * Y + P[1][2] * Z + P[1][3];
    const float cz = P[2][0] * X + P[2][1] * Y + P[2][2] * Z + P[2][3];
    const float sx = cx / (cz + eps), sy = cy / (cz + eps);
    if (tri) tri[static_cast<size_t>(b) * n + i] = cz;
    if (flowo) {
      flowo[(static_cast<size_t>(b) * C * 2 + 2 * c) * HW + p] = sx - u;
      flowo[(static_cast<size_t>(b) * C * 2 + 2 * c + 1) * HW + p] = sy - v;
    }
    if (mask) mask[static_cast<size_t>(b) * n + i] = (sx >= 0.f) & (sx <= static_cast<float>(W - 1)) & (sy >= 0.f) & (sy <= static_cast<float>(H - 1));
  }
}

unsigned grid_for(long long n, int threads) {
  long long blocks = (n + threads - 1) / threads;
  const long long cap = static_cast<long long>(ts::kNumCU) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

int check_bchw(int B, int C, int H, int W, const char* who) {
  TS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "%s: non-positive size", who);
  TS_REQUIRE(static_cast<long long>(C + 1) * H * W < (1ll << 31), TS_ERR_UNSUPPORTED, "%s: tensor too large", who);
  return TS_OK;
}

}  // namespace

extern "C" int ts_softsplat_sum_fwd(const float* input, const float* flow, float* output,
                                    int B, int C, int H, int W, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "softsplat_sum_fwd")) return rc;
  TS_REQUIRE_PTR(input); TS_REQUIRE_PTR(flow); TS_REQUIRE_PTR(output);
  hipStream_t st = ts::as_stream(stream);
  if (hipError_t e = hipMemsetAsync(output, 0, static_cast<size_t>(B) * C * H * W * sizeof(float), st))
    return ts::fail(e, "softsplat: memset");
  hipLaunchKernelGGL(splat_scatter<0>, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0, st,
                     input, flow, nullptr, output, B, C, H, W);
  return ts::launched("splat_scatter");
}

// Order-independent variant of ts_softsplat_sum_fwd (bit-identical from run to run): workspace of
// B*C*H*W*8 bytes; |value| must stay below 2^22 (fixed point with 40 fractional bits).
extern "C" int ts_softsplat_sum_fwd_deterministic(const float* input, const float* flow, float* output, void* workspace,
                                                  int B, int C, int H, int W, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "softsplat_sum_fwd_deterministic")) return rc;
  TS_REQUIRE_PTR(input); TS_REQUIRE_PTR(flow); TS_REQUIRE_PTR(output); TS_REQUIRE_PTR(workspace);
  hipStream_t st = ts::as_stream(stream);
  const long long n = static_cast<long long>(B) * C * H * W;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace);
  if (hipError_t e = hipMemsetAsync(acc, 0, static_cast<size_t>(n) * sizeof(unsigned long long), st))
    return ts::fail(e, "softsplat: memset");
  hipLaunchKernelGGL(splat_scatter_fixed, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0, st,
                     input, flow, acc, B, C, H, W);
  if (int rc = ts::launched("splat_scatter_fixed")) return rc;
  hipLaunchKernelGGL(splat_from_fixed, dim3(grid_for(n, 256)), dim3(256), 0, st, acc, output, n);
  return ts::launched("splat_from_fixed");
}

extern "C" size_t ts_softsplat_softmax_workspace_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return ts::round_up(static_cast<size_t>(B) * (C + 1) * H * W * sizeof(float), 256);
}

extern "C" int ts_softsplat_softmax_fwd(const float* input, const float* flow, const float* metric, float* output,
                                        void* workspace, int B, int C, int H, int W, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "softsplat_softmax_fwd")) return rc;
  TS_REQUIRE_PTR(input); TS_REQUIRE_PTR(flow); TS_REQUIRE_PTR(metric); TS_REQUIRE_PTR(output); TS_REQUIRE_PTR(workspace);
  hipStream_t st = ts::as_stream(stream);
  float* accum = reinterpret_cast<float*>(workspace);
  if (hipError_t e = hipMemsetAsync(accum, 0, static_cast<size_t>(B) * (C + 1) * H * W * sizeof(float), st))
    return ts::fail(e, "softsplat: memset");
  const dim3 grid(grid_for(static_cast<long long>(B) * H * W, 256));
  hipLaunchKernelGGL(splat_scatter<1>, grid, dim3(256), 0, st, input, flow, metric, accum, B, C, H, W);
  if (int rc = ts::launched("splat_scatter")) return rc;
  hipLaunchKernelGGL(splat_normalize, grid, dim3(256), 0, st, accum, output, B, C, H * W);
  return ts::launched("splat_normalize");
}

extern "C" int ts_softsplat_sum_bwd_input(const float* flow, const float* grad_output, float* grad_input,
                                          int B, int C, int H, int W, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "softsplat_sum_bwd_input")) return rc;
  TS_REQUIRE_PTR(flow); TS_REQUIRE_PTR(grad_output); TS_REQUIRE_PTR(grad_input);
  hipLaunchKernelGGL(splat_grad_input, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0,
                     ts::as_stream(stream), flow, grad_output, grad_input, B, C, H, W);
  return ts::launched("splat_grad_input");
}

extern "C" int ts_softsplat_sum_bwd_flow(const float* input, const float* flow, const float* grad_output,
                                         float* grad_flow, int B, int C, int H, int W, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "softsplat_sum_bwd_flow")) return rc;
  TS_REQUIRE_PTR(input); TS_REQUIRE_PTR(flow); TS_REQUIRE_PTR(grad_output); TS_REQUIRE_PTR(grad_flow);
  hipLaunchKernelGGL(splat_grad_flow, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0,
                     ts::as_stream(stream), input, flow, grad_output, grad_flow, B, C, H, W);
  return ts::launched("splat_grad_flow");
}

extern "C" int ts_project_to_3d_fwd(const float* depth, const float* K, const float* inv_K, const float* T,
                                    float* triangular_depth, float* optical_flow, unsigned char* flow_mask,
                                    int B, int C, int H, int W, int k_dim, int inv_k_dim, float eps, void* stream) {
  if (int rc = check_bchw(B, C, H, W, "project_to_3d")) return rc;
  TS_REQUIRE(k_dim == 3 || k_dim == 4, TS_ERR_SHAPE, "project_to_3d: K must be 3x3 or 4x4");
  TS_REQUIRE(inv_k_dim == 3 || inv_k_dim == 4, TS_ERR_SHAPE, "project_to_3d: inv_K must be 3x3 or 4x4");
  TS_REQUIRE(B <= 65535, TS_ERR_UNSUPPORTED, "project_to_3d: batch too large");
  TS_REQUIRE_PTR(depth); TS_REQUIRE_PTR(K); TS_REQUIRE_PTR(inv_K); TS_REQUIRE_PTR(T);
  const dim3 grid(grid_for(static_cast<long long>(C) * H * W, 256), B);
  hipLaunchKernelGGL(project_kernel, grid, dim3(256), 0, ts::as_stream(stream), depth, K, inv_K, T,
                     triangular_depth, optical_flow, flow_mask, B, C, H, W, k_dim, inv_k_dim, eps);
  return ts::launched("project_kernel");
}
