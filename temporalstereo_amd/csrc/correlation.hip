// Native correlation volumes: what the reference obtains from the third-party
// `spatial_correlation_sampler.SpatialCorrelationSampler` (unvendored, unpinned: architecture/modeling/aggregation/utils/
// correlation.py:4-7 imports it inside try/except) in
//   correlation      correlation.py:10-29   out[b, ph*pW+pw, y, x] = lrelu_0.1( sum_c L[b,c,y,x] * R[b,c,y+ph-pH/2, x+pw-pW/2] )
//   correlation1d    correlation.py:32-57   the same with patch (1, 2*max_disp-1), first max_disp planes kept:
//                                           plane k = shift x + k - (max_disp-1), i.e. disparity max_disp-1-k
// for the arguments the reference uses (kernel_size 1, stride 1, padding 0, dilation 1, dilation_patch 1: the sampler's
// published definition -- a plain sum over channels, zeros outside the image).  The literal "shift-and-correlate over D
// candidate disparities" of the north star; no shipped configuration calls it.
//
// One lane = 4 consecutive pixels of one output plane: the left values are one 16-byte load per channel (the same for every
// plane of the pixel: served by L1 / L2 after the first plane), the shifted right values four clamped scalar loads selected
// afterwards (no load sits behind a branch).  Backward: two gather kernels (deterministic, no atomics).
#include "ts_common.hpp"

namespace {

struct Corr {
  int B, C, H, W, pH, pW, keep;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.1f * v; }

__global__ void __launch_bounds__(256)
correlation_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ out, const Corr p) {
  const int Wq = (p.W + 3) / 4;
  const long long n = static_cast<long long>(p.B) * p.keep * p.H * Wq;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xq = static_cast<int>(i % Wq);
  long long t = i / Wq;
  const int y = static_cast<int>(t % p.H); t /= p.H;
  const int k = static_cast<int>(t % p.keep), b = static_cast<int>(t / p.keep);
  const int dy = k / p.pW - p.pH / 2, dx = k % p.pW - p.pW / 2;
  const int x0 = xq * 4, ys = y + dy;
  const bool rowok = ys >= 0 && ys < p.H;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* Lp = L + static_cast<size_t>(b) * p.C * HW + static_cast<size_t>(y) * p.W;
  const float* Rp = R + static_cast<size_t>(b) * p.C * HW + static_cast<size_t>(rowok ? ys : 0) * p.W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int xs[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int xr = x0 + q + dx;
    ok[q] = rowok && xr >= 0 && xr < p.W && x0 + q < p.W;
    xs[q] = min(max(xr, 0), p.W - 1);
  }
  for (int c = 0; c < p.C; ++c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float l = Lp[c * HW + min(x0 + q, p.W - 1)];
      const float r = Rp[c * HW + xs[q]];
      acc[q] += ok[q] ? l * r : 0.f;
    }
  }
  float* op = out + ((static_cast<size_t>(b) * p.keep + k) * p.H + y) * p.W;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (x0 + q < p.W) op[x0 + q] = lrelu(acc[q]);
}

// grad wrt the left map: gL[c,y,x] = sum_k g'[k,y,x] R[c, y+dy_k, x+dx_k],   g' = g * (out > 0 ? 1 : 0.1)
// grad wrt the right map: gR[c,y,x] = sum_k g'[k, y-dy_k, x-dx_k] L[c, y-dy_k, x-dx_k]
template <bool RIGHT>
__global__ void __launch_bounds__(256)
correlation_bwd_kernel(const float* __restrict__ other, const float* __restrict__ out, const float* __restrict__ g,
                       float* __restrict__ grad, const Corr p) {
  const long long n = static_cast<long long>(p.B) * p.C * p.H * p.W;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = static_cast<int>(i % p.W);
  long long t = i / p.W;
  const int y = static_cast<int>(t % p.H); t /= p.H;
  const int c = static_cast<int>(t % p.C), b = static_cast<int>(t / p.C);
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* op = other + (static_cast<size_t>(b) * p.C + c) * HW;
  float acc = 0.f;
  for (int k = 0; k < p.keep; ++k) {
    const int dy = k / p.pW - p.pH / 2, dx = k % p.pW - p.pW / 2;
    // LEFT: the output pixel is (y, x), the partner R(y+dy, x+dx); RIGHT: the output pixel is (y-dy, x-dx), the partner L there
    const int oy = RIGHT ? y - dy : y, ox = RIGHT ? x - dx : x;
    const int py = RIGHT ? oy : y + dy, px = RIGHT ? ox : x + dx;
    const bool ok = oy >= 0 && oy < p.H && ox >= 0 && ox < p.W && py >= 0 && py < p.H && px >= 0 && px < p.W;
    const size_t oi = ((static_cast<size_t>(b) * p.keep + k) * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0);
    const float gv = g[oi] * (out[oi] > 0.f ? 1.f : 0.1f);
    const float ov = op[static_cast<size_t>(ok ? py : 0) * p.W + (ok ? px : 0)];
    acc += ok ? gv * ov : 0.f;
  }
  grad[i] = acc;
}

int check(const Corr& p) {
  TS_REQUIRE(p.B > 0 && p.C > 0 && p.H > 0 && p.W > 0, TS_ERR_SHAPE, "correlation: non-positive size");
  TS_REQUIRE(p.pH >= 1 && p.pW >= 1 && (p.pH & 1) && (p.pW & 1), TS_ERR_SHAPE, "correlation: patch sizes must be odd and >= 1");
  TS_REQUIRE(p.keep >= 1 && p.keep <= p.pH * p.pW, TS_ERR_SHAPE, "correlation: keep outside 1..pH*pW");
  return TS_OK;
}

}  // namespace

extern "C" int ts_correlation_fwd(const float* left, const float* right, float* out, int B, int C, int H, int W, int patch_h,
                                  int patch_w, int keep, void* stream) {
  const Corr p{B, C, H, W, patch_h, patch_w, keep};
  if (int rc = check(p)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(out);
  const long long n = static_cast<long long>(B) * keep * H * ((W + 3) / 4);
  hipLaunchKernelGGL(correlation_fwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ts::as_stream(stream),
                     left, right, out, p);
  return ts::launched("correlation_fwd_kernel");
}

extern "C" int ts_correlation_bwd(const float* left, const float* right, const float* out, const float* grad_out,
                                  float* grad_left, float* grad_right, int B, int C, int H, int W, int patch_h, int patch_w,
                                  int keep, void* stream) {
  const Corr p{B, C, H, W, patch_h, patch_w, keep};
  if (int rc = check(p)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(out); TS_REQUIRE_PTR(grad_out);
  const long long n = static_cast<long long>(B) * C * H * W;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (grad_left) {
    hipLaunchKernelGGL(correlation_bwd_kernel<false>, grid, dim3(256), 0, ts::as_stream(stream), right, out, grad_out, grad_left, p);
    if (int rc = ts::launched("correlation_bwd_kernel<left>")) return rc;
  }
  if (grad_right) {
    hipLaunchKernelGGL(correlation_bwd_kernel<true>, grid, dim3(256), 0, ts::as_stream(stream), left, out, grad_out, grad_right, p);
    if (int rc = ts::launched("correlation_bwd_kernel<right>")) return rc;
  }
  return TS_OK;
}
