// K4 -- disparity regression for gfx950 (MI355X).
//
//   K4a  top-k soft-argmax with learned per-candidate offset
//        replaces predict_disp()  architecture/modeling/aggregation/TemporalStereo/coarse.py:69-75
//        (== fine.py:70-76, precise.py:61-67): topk -> softmax -> gather(sample+offset) -> sum.
//   K4b  full soft-argmin / argmin over D
//        replaces SOFTARGMIN.forward  architecture/modeling/prediction/soft_argmin.py:38-59 and
//        ARGMIN.forward  architecture/modeling/prediction/argmin.py:35-46.
//
// Layout is [B, D, H, W] with W contiguous, so consecutive lanes own consecutive pixels and every
// load of a candidate plane is a coalesced row segment; the D axis is a stride of H*W.  K4a keeps
// its k best candidates in registers (D <= 14 in the shipped configs).  K4b splits the D axis of a
// pixel over 4 lanes (16 pixels x 4 slices per wavefront) and merges the online-softmax partials
// (max, sum, weighted sum) with wavefront shuffles, which quarters the serial chain at D=192.
// All of it is HBM-bound: inputs read once, outputs written once.
#include "ts_common.hpp"

namespace {

constexpr int KMAX = 8;

// ---------------------------------------------------------------------------------------- K4a
template <int K>
__global__ void __launch_bounds__(256)
topk_softargmax_fwd(const float* __restrict__ cost, const float* __restrict__ samp, const float* __restrict__ off,
                    float* __restrict__ disp, float* __restrict__ tdisp, float* __restrict__ tcost,
                    int* __restrict__ tidx, int B, int D, int HW) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const size_t base = static_cast<size_t>(b) * D * HW + p;
    float bc[K];
    int bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bc[j] = -INFINITY; bi[j] = -1; }
    // eight candidates requested at a time (clamped index: unconditional loads), then inserted in order -- a
    // load-per-iteration loop serialises D global round trips on the small levels
    for (int d0 = 0; d0 < D; d0 += 8) {
      float cc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cc[u] = cost[base + static_cast<size_t>(min(d0 + u, D - 1)) * HW];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float c = cc[u];
        int ci = d0 + u;
        const bool real = ci < D;
        // insertion into the descending list; strict '>' keeps the lowest index first among ties
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const bool take = real && ((c > bc[j]) || (bi[j] < 0));
          const float tc = bc[j];
          const int ti = bi[j];
          if (take) { bc[j] = c; bi[j] = ci; c = tc; ci = ti; }
        }
      }
    }
    // softmax over the k kept costs (bc[0] is the maximum)
    float w[K], den = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { w[j] = expf(bc[j] - bc[0]); den += w[j]; }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const size_t o = base + static_cast<size_t>(bi[j]) * HW;
      const float td = samp[o] + off[o];
      acc += (w[j] / den) * td;
      const size_t ko = (static_cast<size_t>(b) * K + j) * HW + p;
      tdisp[ko] = td;
      tcost[ko] = bc[j];
      if (tidx) tidx[ko] = bi[j];
    }
    disp[i] = acc;
  }
}

// grads of (disp, topk_disp, topk_cost) -> grads of (cost, sample (+offset: same tensor))
template <int K>
__global__ void __launch_bounds__(256)
topk_softargmax_bwd(const float* __restrict__ tdisp, const float* __restrict__ tcost, const int* __restrict__ tidx,
                    const float* __restrict__ disp, const float* __restrict__ gdisp, const float* __restrict__ gtd,
                    const float* __restrict__ gtc, float* __restrict__ gcost, float* __restrict__ gsamp,
                    int B, int D, int HW) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const size_t base = static_cast<size_t>(b) * D * HW + p;
    for (int d = 0; d < D; ++d) {
      if (gcost) gcost[base + static_cast<size_t>(d) * HW] = 0.f;
      if (gsamp) gsamp[base + static_cast<size_t>(d) * HW] = 0.f;
    }
    float c[K], td[K], w[K], den = 0.f;
    int idx[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const size_t ko = (static_cast<size_t>(b) * K + j) * HW + p;
      c[j] = tcost[ko]; td[j] = tdisp[ko]; idx[j] = tidx[ko];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) { w[j] = expf(c[j] - c[0]); den += w[j]; }
    const float g = gdisp ? gdisp[i] : 0.f;
    const float dv = disp[i];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const size_t ko = (static_cast<size_t>(b) * K + j) * HW + p;
      const float pj = w[j] / den;
      const size_t o = base + static_cast<size_t>(idx[j]) * HW;
      if (gcost) gcost[o] = g * pj * (td[j] - dv) + (gtc ? gtc[ko] : 0.f);
      if (gsamp) gsamp[o] = g * pj + (gtd ? gtd[ko] : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------- K4b
// One wavefront = 16 consecutive pixels x 4 slices of the D axis.
struct Partial { float m, s, ws; };   // running max, sum exp, sum exp*sample

__device__ __forceinline__ Partial merge(Partial a, Partial b) {
  const float m = fmaxf(a.m, b.m);
  const float ea = (a.m == -INFINITY) ? 0.f : expf(a.m - m);
  const float eb = (b.m == -INFINITY) ? 0.f : expf(b.m - m);
  return Partial{m, a.s * ea + b.s * eb, a.ws * ea + b.ws * eb};
}

template <bool NORMALIZE>
__global__ void __launch_bounds__(256)
soft_argmin_fwd(const float* __restrict__ cost, const float* __restrict__ samp, float* __restrict__ disp,
                float temperature, int B, int D, int HW) {
  const int lane = threadIdx.x & 63;
  const int slice = lane >> 4;                 // 0..3
  const long long wave = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const long long nwave = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
  const long long npix = static_cast<long long>(B) * HW;
  for (long long p0 = wave * 16; p0 < npix; p0 += nwave * 16) {
    const long long i = p0 + (lane & 15);
    const bool ok = i < npix;
    const int b = ok ? static_cast<int>(i / HW) : 0;
    const int p = ok ? static_cast<int>(i - static_cast<long long>(b) * HW) : 0;
    const size_t base = static_cast<size_t>(b) * D * HW + p;
    Partial a{-INFINITY, 0.f, 0.f};
    float lin = 0.f;
    if (ok) {
      for (int d = slice; d < D; d += 4) {
        const float c = cost[base + static_cast<size_t>(d) * HW] * temperature;
        const float sv = samp[base + static_cast<size_t>(d) * HW];
        if constexpr (NORMALIZE) {
          a = merge(a, Partial{c, 1.f, sv});
        } else {
          lin += c * sv;
        }
      }
    }
    if constexpr (NORMALIZE) {
      // wavefront-shuffle reduction across the 4 slices of a pixel (lanes l, l^16, l^32, l^48)
#pragma unroll
      for (int s = 16; s <= 32; s <<= 1) {
        Partial o{__shfl_xor(a.m, s), __shfl_xor(a.s, s), __shfl_xor(a.ws, s)};
        a = merge(a, o);
      }
      if (ok && slice == 0) disp[i] = a.ws / a.s;
    } else {
      lin += __shfl_xor(lin, 16);
      lin += __shfl_xor(lin, 32);
      if (ok && slice == 0) disp[i] = lin;
    }
  }
}

template <bool NORMALIZE>
__global__ void __launch_bounds__(256)
soft_argmin_bwd(const float* __restrict__ cost, const float* __restrict__ samp, const float* __restrict__ disp,
                const float* __restrict__ gdisp, float* __restrict__ gcost, float* __restrict__ gsamp,
                float temperature, int B, int D, int HW) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const size_t base = static_cast<size_t>(b) * D * HW + p;
    const float g = gdisp[i];
    if constexpr (NORMALIZE) {
      float m = -INFINITY, s = 0.f;
      for (int d = 0; d < D; ++d) {
        const float c = cost[base + static_cast<size_t>(d) * HW] * temperature;
        const float mn = fmaxf(m, c);
        s = s * expf(m - mn) + expf(c - mn);
        m = mn;
      }
      const float dv = disp[i];
      for (int d = 0; d < D; ++d) {
        const size_t o = base + static_cast<size_t>(d) * HW;
        const float pd = expf(cost[o] * temperature - m) / s;
        if (gcost) gcost[o] = g * temperature * pd * (samp[o] - dv);
        if (gsamp) gsamp[o] = g * pd;
      }
    } else {
      for (int d = 0; d < D; ++d) {
        const size_t o = base + static_cast<size_t>(d) * HW;
        if (gcost) gcost[o] = g * temperature * samp[o];
        if (gsamp) gsamp[o] = g * temperature * cost[o];
      }
    }
  }
}

__global__ void __launch_bounds__(256)
argmax_select_fwd(const float* __restrict__ cost, const float* __restrict__ samp, float* __restrict__ disp,
                  int* __restrict__ idx, int B, int D, int HW) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int p = static_cast<int>(i - static_cast<long long>(b) * HW);
    const size_t base = static_cast<size_t>(b) * D * HW + p;
    float best = -INFINITY;
    int bi = 0;
    for (int d = 0; d < D; ++d) {
      const float c = cost[base + static_cast<size_t>(d) * HW];
      if (c > best) { best = c; bi = d; }
    }
    disp[i] = samp[base + static_cast<size_t>(bi) * HW];
    if (idx) idx[i] = bi;
  }
}

int check_bdhw(int B, int D, int H, int W, const char* who) {
  TS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "%s: non-positive size", who);
  TS_REQUIRE(static_cast<long long>(H) * W < (1ll << 31), TS_ERR_UNSUPPORTED, "%s: plane too large", who);
  return TS_OK;
}

unsigned grid_for(long long n, int threads) {
  long long blocks = (n + threads - 1) / threads;
  const long long cap = static_cast<long long>(ts::kNumCU) * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

}  // namespace

extern "C" int ts_topk_softargmax_fwd(const float* cost, const float* sample, const float* offset, float* disp,
                                      float* topk_disp, float* topk_cost, int* topk_index,
                                      int B, int D, int H, int W, int k, void* stream) {
  if (int rc = check_bdhw(B, D, H, W, "topk_softargmax")) return rc;
  TS_REQUIRE(k >= 1 && k <= KMAX && k <= D, TS_ERR_UNSUPPORTED, "topk_softargmax: k=%d outside 1..min(%d, D)", k, KMAX);
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(offset);
  TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(topk_disp); TS_REQUIRE_PTR(topk_cost);
  const int HW = H * W;
  const dim3 grid(grid_for(static_cast<long long>(B) * HW, 256));
  hipStream_t st = ts::as_stream(stream);
#define TS_CASE(KK)                                                                                     \
  case KK:                                                                                              \
    hipLaunchKernelGGL(topk_softargmax_fwd<KK>, grid, dim3(256), 0, st, cost, sample, offset, disp,      \
                       topk_disp, topk_cost, topk_index, B, D, HW);                                     \
    break;
  switch (k) { TS_CASE(1) TS_CASE(2) TS_CASE(3) TS_CASE(4) TS_CASE(5) TS_CASE(6) TS_CASE(7) TS_CASE(8) }
#undef TS_CASE
  return ts::launched("topk_softargmax_fwd");
}

extern "C" int ts_topk_softargmax_bwd(const float* topk_disp, const float* topk_cost, const int* topk_index,
                                      const float* disp, const float* grad_disp, const float* grad_topk_disp,
                                      const float* grad_topk_cost, float* grad_cost, float* grad_sample,
                                      int B, int D, int H, int W, int k, void* stream) {
  if (int rc = check_bdhw(B, D, H, W, "topk_softargmax_bwd")) return rc;
  TS_REQUIRE(k >= 1 && k <= KMAX && k <= D, TS_ERR_UNSUPPORTED, "topk_softargmax_bwd: k=%d outside 1..min(%d, D)", k, KMAX);
  TS_REQUIRE_PTR(topk_disp); TS_REQUIRE_PTR(topk_cost); TS_REQUIRE_PTR(topk_index); TS_REQUIRE_PTR(disp);
  const int HW = H * W;
  const dim3 grid(grid_for(static_cast<long long>(B) * HW, 256));
  hipStream_t st = ts::as_stream(stream);
#define TS_CASE(KK)                                                                                      \
  case KK:                                                                                               \
    hipLaunchKernelGGL(topk_softargmax_bwd<KK>, grid, dim3(256), 0, st, topk_disp, topk_cost, topk_index, \
                       disp, grad_disp, grad_topk_disp, grad_topk_cost, grad_cost, grad_sample, B, D, HW); \
    break;
  switch (k) { TS_CASE(1) TS_CASE(2) TS_CASE(3) TS_CASE(4) TS_CASE(5) TS_CASE(6) TS_CASE(7) TS_CASE(8) }
#undef TS_CASE
  return ts::launched("topk_softargmax_bwd");
}

extern "C" int ts_softargmin_fwd(const float* cost, const float* sample, float* disp, float temperature,
                                 int normalize, int B, int D, int H, int W, void* stream) {
  if (int rc = check_bdhw(B, D, H, W, "softargmin")) return rc;
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(disp);
  const int HW = H * W;
  const long long nthreads = (static_cast<long long>(B) * HW + 15) / 16 * 64;    // 4 lanes per pixel
  const dim3 grid(grid_for(nthreads, 256));
  hipStream_t st = ts::as_stream(stream);
  if (normalize) hipLaunchKernelGGL(soft_argmin_fwd<true>, grid, dim3(256), 0, st, cost, sample, disp, temperature, B, D, HW);
  else hipLaunchKernelGGL(soft_argmin_fwd<false>, grid, dim3(256), 0, st, cost, sample, disp, temperature, B, D, HW);
  return ts::launched("softargmin_fwd");
}

extern "C" int ts_softargmin_bwd(const float* cost, const float* sample, const float* disp, const float* grad_disp,
                                 float* grad_cost, float* grad_sample, float temperature, int normalize,
                                 int B, int D, int H, int W, void* stream) {
  if (int rc = check_bdhw(B, D, H, W, "softargmin_bwd")) return rc;
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(grad_disp);
  const int HW = H * W;
  const dim3 grid(grid_for(static_cast<long long>(B) * HW, 256));
  hipStream_t st = ts::as_stream(stream);
  if (normalize) hipLaunchKernelGGL(soft_argmin_bwd<true>, grid, dim3(256), 0, st, cost, sample, disp, grad_disp, grad_cost, grad_sample, temperature, B, D, HW);
  else hipLaunchKernelGGL(soft_argmin_bwd<false>, grid, dim3(256), 0, st, cost, sample, disp, grad_disp, grad_cost, grad_sample, temperature, B, D, HW);
  return ts::launched("softargmin_bwd");
}

extern "C" int ts_argmax_select_fwd(const float* cost, const float* sample, float* disp, int* index,
                                    int B, int D, int H, int W, void* stream) {
  if (int rc = check_bdhw(B, D, H, W, "argmax_select")) return rc;
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(disp);
  const int HW = H * W;
  hipLaunchKernelGGL(argmax_select_fwd, dim3(grid_for(static_cast<long long>(B) * HW, 256)), dim3(256), 0,
                     ts::as_stream(stream), cost, sample, disp, index, B, D, HW);
  return ts::launched("argmax_select_fwd");
}
