// Train-mode BatchNorm + activation for the convolution wrappers of the path (training form), gfx950.
//
// The reference's Conv3d / Conv2d / ConvTranspose wrappers are conv -> BatchNorm -> activation
// (architecture/modeling/layers/basic_layers.py:194-235: `self.norm`, `self.activation` applied in forward); in train()
// mode BatchNorm uses the batch statistics and cannot be folded into the convolution.  As framework ops that is a
// statistics kernel, a normalisation kernel and an activation kernel per layer forward, and their three autograd
// nodes backward -- 88 layers per frame.  Here:
//   bn_stats      per channel mean / biased variance of x [B,C,N] in two deterministic stages (per-chunk sums of
//                 x - pivot and (x - pivot)^2, pivot = the channel's first element, so that E[x'^2] - E[x']^2 does
//                 not cancel; fixed-order finish in double), optionally updating the running statistics
//   bn_apply_act  out = act((x - mean) * rsqrt(var + eps) * gamma + beta)            (act: none | SiLU | ReLU)
//   bn_bwd_reduce s1[c] = sum dz, s2[c] = sum dz * xhat,  dz = dy * act'(z), z recomputed from x (no z / xhat kept)
//   bn_bwd_apply  dx = (dz - s1/n - xhat * s2/n) * invstd * gamma     (train);  dx = dz * invstd * gamma   (eval)
// HBM-bound element kernels: 16 bytes per lane where the plane length allows it.  Cross-rank statistics
// (SyncBatchNorm, dist.py) enter between bn_stats and bn_apply_act / between bn_bwd_reduce and bn_bwd_apply as
// all-reduced [C] vectors: the kernels take mean / var / s1 / s2 / n as arguments.
#include <cstdlib>

#include "ts_common.hpp"

namespace {

enum { BN_ACT_NONE = 0, BN_ACT_SILU = 1, BN_ACT_RELU = 2 };

// 1 / (1 + e^-z) as v_exp_f32 + v_rcp_f32 + one Newton step (conv_common.hpp silu_fast: the inference epilogues' form; ~1.5 ulp).  The
// library expf + IEEE division are ~40 instructions per element: on a 32 x 272 x 480 layer that alone was 6.7 us of the 13.9 us
// bn_finish_apply_act_kernel took (and of every backward kernel's, which evaluates it again for the derivative).  e^-z = inf
// (z < -88.7): the unrefined reciprocal 0 is kept, the Newton step there would be inf * 0.
__device__ __forceinline__ float sigmoid_fast(float z) {
#ifdef TS_EXACT_SILU
  return 1.f / (1.f + expf(-z));
#else
  const float d = 1.f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
  const float r = __builtin_amdgcn_rcpf(d);
  const float rn = fmaf(fmaf(-d, r, 1.f), r, r);
  return d < 3.0e38f ? rn : r;
#endif
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == BN_ACT_SILU) return z * sigmoid_fast(z);
  if (act == BN_ACT_RELU) return fmaxf(z, 0.f);
  return z;
}

__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == BN_ACT_SILU) {
    const float s = sigmoid_fast(z);
    return s * (1.f + z * (1.f - s));
  }
  if (act == BN_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ void block_sum2(float& a, float& b) {      // 256 threads -> thread 0
  __shared__ float sa[4], sb[4];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sa[0] + sa[1] + sa[2] + sa[3];
    b = sb[0] + sb[1] + sb[2] + sb[3];
  }
}

struct BN {
  int B, C;
  long long N;                  // elements per (batch, channel) plane block (D*H*W), contiguous
  long long bstride, cstride;   // of x (elements)
  int nchunk;                   // chunks per (batch item, channel)
  long long chunk;              // elements per chunk
};

// grid (nchunk, C, B): partial[(c * B + b) * nchunk + k] = (sum x', sum x'^2), x' = x - x[b=0, c, 0]
__global__ void __launch_bounds__(256)
bn_stats_partial(const float* __restrict__ x, float2* __restrict__ partial, const BN p, int vec) {
  const int k = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float pivot = x[static_cast<size_t>(c) * p.cstride];
  const float* xp = x + static_cast<size_t>(b) * p.bstride + static_cast<size_t>(c) * p.cstride;
  const long long lo = k * p.chunk, hi = min(p.N, lo + p.chunk);
  float s = 0.f, q = 0.f;
  if (vec) {                  // N, the chunk and the strides are multiples of 4, the base 16-byte aligned: a quad per lane and trip
    for (long long i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
      const float4 t = *reinterpret_cast<const float4*>(xp + i);
      const float v0 = t.x - pivot, v1 = t.y - pivot, v2 = t.z - pivot, v3 = t.w - pivot;
      s += (v0 + v1) + (v2 + v3);
      q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
  } else
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float v = xp[i] - pivot;
    s += v;
    q += v * v;
  }
  block_sum2(s, q);
  if (threadIdx.x == 0) partial[(static_cast<size_t>(c) * p.B + b) * p.nchunk + k] = make_float2(s, q);
}

// one workgroup per channel: fixed-order sum in double -> mean, biased variance; optional running update
// (running_var takes the unbiased estimate, as nn.BatchNorm does)
__global__ void __launch_bounds__(64)
bn_stats_finish(const float* __restrict__ x, const float2* __restrict__ partial, float* __restrict__ mean, float* __restrict__ var,
                float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                long long* __restrict__ num_batches_tracked, const BN p) {
  const int c = blockIdx.x;
  if (num_batches_tracked && c == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  const int n = p.B * p.nchunk;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float2 v = partial[static_cast<size_t>(c) * n + i];
    s += v.x;
    q += v.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (threadIdx.x == 0) {
    const double cnt = static_cast<double>(p.B) * static_cast<double>(p.N);
    const double m1 = s / cnt;
    const double v = fmax(q / cnt - m1 * m1, 0.0);
    const float m = static_cast<float>(m1 + static_cast<double>(x[static_cast<size_t>(c) * p.cstride]));
    mean[c] = m;
    var[c] = static_cast<float>(v);
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      const double unb = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * static_cast<float>(unb);
    }
  }
}

struct BNA {
  int B, C;
  long long N;
  long long xb, xc, ob, oc;     // strides (elements) of x / dy and of out / dx
  int act, train;
  float eps, inv_n;
};

// grid (chunks of 1024 elements, C, B)
__global__ void __launch_bounds__(256)
bn_apply_act_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out, const BNA p) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float m = mean[c], is = rsqrtf(var[c] + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float sc = is * g, sh = be - m * sc;
  const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
  float* op = out + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
  const long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + k < p.N) op[i0 + k] = act_fwd(xp[i0 + k] * sc + sh, p.act);
}

// grid (nchunk, C, B): partial sums of dz and dz * xhat
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                     const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float2* __restrict__ partial, const BNA p, int nchunk, long long chunk, int vec) {
  const int k = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float m = mean[c], is = rsqrtf(var[c] + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
  const float* gp = dy + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
  const long long lo = k * chunk, hi = min(p.N, lo + chunk);
  float s1 = 0.f, s2 = 0.f;
  if (vec) {
    for (long long i = lo + 4 * threadIdx.x; i < hi; i += 1024) {
      const float4 xt = *reinterpret_cast<const float4*>(xp + i);
      const float4 gt = *reinterpret_cast<const float4*>(gp + i);
      const float xs[4] = {xt.x, xt.y, xt.z, xt.w}, gs[4] = {gt.x, gt.y, gt.z, gt.w};
      float d1 = 0.f, d2 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xs[e] - m) * is;
        const float dz = gs[e] * act_grad(xh * g + be, p.act);
        d1 += dz;
        d2 += dz * xh;
      }
      s1 += d1;
      s2 += d2;
    }
  } else
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const float xh = (xp[i] - m) * is;
    const float dz = gp[i] * act_grad(xh * g + be, p.act);
    s1 += dz;
    s2 += dz * xh;
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) partial[(static_cast<size_t>(c) * p.B + b) * nchunk + k] = make_float2(s1, s2);
}

__global__ void __launch_bounds__(64)
bn_bwd_finish_kernel(const float2* __restrict__ partial, float* __restrict__ s1, float* __restrict__ s2, int n) {
  const int c = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) {
    const float2 v = partial[static_cast<size_t>(c) * n + i];
    a += v.x;
    b += v.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (threadIdx.x == 0) { s1[c] = static_cast<float>(a); s2[c] = static_cast<float>(b); }
}

__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                    const float* __restrict__ s1, const float* __restrict__ s2, float* __restrict__ dx, const BNA p) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float m = mean[c], is = rsqrtf(var[c] + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float a1 = p.train ? s1[c] * p.inv_n : 0.f, a2 = p.train ? s2[c] * p.inv_n : 0.f;
  const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
  const float* gp = dy + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
  float* op = dx + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
  const long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (i0 + k < p.N) {
      const float xh = (xp[i0 + k] - m) * is;
      const float dz = gp[i0 + k] * act_grad(xh * g + be, p.act);
      op[i0 + k] = (dz - a1 - xh * a2) * is * g;
    }
}


// Sum of one channel's `n` partials in the fixed order of the finish kernels (64 lanes striding, xor tree in double), by
// wave 0 of the calling workgroup; every thread gets the result.
__device__ __forceinline__ void channel_partial_sum(const float2* __restrict__ partial, int c, int n, double& a, double& b) {
  __shared__ double bc[2];
  if (threadIdx.x < 64) {
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) {
      const float2 v = partial[static_cast<size_t>(c) * n + i];
      s += v.x;
      q += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (threadIdx.x == 0) { bc[0] = s; bc[1] = q; }
  }
  __syncthreads();
  a = bc[0];
  b = bc[1];
}

// bn_stats_finish + bn_apply_act in one launch (single-rank training: nothing is exchanged between the two): every workgroup
// sums its channel's partials itself (a few hundred bytes from L2, the same fixed order, so all of them agree to the bit) and
// workgroup (0, c, 0) leaves mean / var for the backward pass and updates the running statistics.
// four consecutive elements of a plane block, requested BEFORE anything that waits (VEC: one 16-byte access; N % 4 == 0 then, so a
// quad is inside or outside as a whole)
template <bool VEC>
__device__ __forceinline__ void ld4n(const float* __restrict__ p, long long i0, long long N, float (&v)[4]) {
  if constexpr (VEC) {
    const float4 t = i0 < N ? *reinterpret_cast<const float4*>(p + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < N ? p[i0 + k] : 0.f;
  }
}
template <bool VEC>
__device__ __forceinline__ void st4n(float* __restrict__ p, long long i0, long long N, const float (&v)[4]) {
  if constexpr (VEC) {
    if (i0 < N) *reinterpret_cast<float4*>(p + i0) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i0 + k < N) p[i0 + k] = v[k];
  }
}

// (the workgroup's own elements and the pivot are requested first: the partial sums' round trip, the pivot's and the elements' used
// to follow one another -- three serialized round trips in a workgroup that lives for one)
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_finish_apply_act_kernel(const float* __restrict__ x, const float2* __restrict__ partial, int npart, float* __restrict__ mean,
                           float* __restrict__ var, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                           long long* __restrict__ num_batches_tracked, const float* __restrict__ gamma,
                           const float* __restrict__ beta, float* __restrict__ out, const BNA p) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  float xv[4];
  ld4n<VEC>(x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc, i0, p.N, xv);
  const float pivot = x[static_cast<size_t>(c) * p.xc];
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  double s, q;
  channel_partial_sum(partial, c, npart, s, q);
  const double cnt = static_cast<double>(p.B) * static_cast<double>(p.N);
  const double m1 = s / cnt;
  const double v = fmax(q / cnt - m1 * m1, 0.0);
  const float m = static_cast<float>(m1 + static_cast<double>(pivot));
  const float vf = static_cast<float>(v);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    mean[c] = m;
    var[c] = vf;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      const double unb = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * static_cast<float>(unb);
    }
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
  }
  const float is = rsqrtf(vf + p.eps);
  const float sc = is * g, sh = be - m * sc;
  float ov[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ov[k] = act_fwd(xv[k] * sc + sh, p.act);
  st4n<VEC>(out + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc, i0, p.N, ov);
}

// bn_bwd_finish + bn_bwd_apply in one launch, the same way; workgroup (0, c, 0) writes the two sums (grad beta, grad gamma)
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_bwd_finish_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                           const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                           const float2* __restrict__ partial, int npart, float* __restrict__ s1o, float* __restrict__ s2o,
                           float* __restrict__ dx, const BNA p) {
  const int c = blockIdx.y, b = blockIdx.z;
  const long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  float xv[4], gv[4];
  ld4n<VEC>(x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc, i0, p.N, xv);
  ld4n<VEC>(dy + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc, i0, p.N, gv);
  const float m = mean[c], is = rsqrtf(var[c] + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  double sa, sb;
  channel_partial_sum(partial, c, npart, sa, sb);
  const float s1 = static_cast<float>(sa), s2 = static_cast<float>(sb);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) { s1o[c] = s1; s2o[c] = s2; }
  const float a1 = s1 * p.inv_n, a2 = s2 * p.inv_n;
  float ov[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float xh = (xv[k] - m) * is;
    const float dz = gv[k] * act_grad(xh * g + be, p.act);
    ov[k] = (dz - a1 - xh * a2) * is * g;
  }
  st4n<VEC>(dx + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc, i0, p.N, ov);
}

// ---- one-launch forms for SMALL layers ------------------------------------------------------------------------------------------
// Most train-mode layers of the pyramid are small (a channel of the 1/16 ... 1/64 hourglass levels is 405 ... 28,560 elements): their
// two launches each way are two launch latencies around a few microseconds of traffic.  One workgroup of 1024 threads per channel
// does both passes itself -- sums (per-thread fp32 partials of x - pivot, combined in double in a fixed order), then the
// normalisation / input gradient out of L2 -- so such a layer costs ONE launch forward and ONE backward.
constexpr int kSmallThreads = 1024;
constexpr long long kSmallElems = 10240;          // per channel (B * N).  Measured (tools/exp/bn_small_bench.py, us per call, one / two launches): N = 405 ... 8160: 6.0-6.7 / 8.8-9.5 forward, 7.3-7.8 / 10.3-10.6 backward; N = 24480: 10.3 / 9.4 and 16.2 / 10.5 (one CU streams the channel twice); 65280: 19.9 / 9.5

__device__ __forceinline__ void block_sum2_double(double& a, double& b) {      // kSmallThreads threads -> every thread
  __shared__ double sa[kSmallThreads / 64], sb[kSmallThreads / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  double ra = 0.0, rb = 0.0;
#pragma unroll
  for (int i = 0; i < kSmallThreads / 64; ++i) { ra += sa[i]; rb += sb[i]; }
  a = ra;
  b = rb;
}

template <bool VEC>
__global__ void __launch_bounds__(kSmallThreads)
bn_train_fwd_small_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ var,
                          float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                          long long* __restrict__ num_batches_tracked, const float* __restrict__ gamma,
                          const float* __restrict__ beta, float* __restrict__ out, const BNA p) {
  const int c = blockIdx.x;
  const float pivot = x[static_cast<size_t>(c) * p.xc];
  float s = 0.f, q = 0.f;
  const long long n4 = VEC ? p.N / 4 : 0;          // VEC: N, the strides and the bases are multiples of 4 elements
  for (int b = 0; b < p.B; ++b) {
    const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
    for (long long i = threadIdx.x; i < n4; i += kSmallThreads) {
      const float4 u = reinterpret_cast<const float4*>(xp)[i];
      const float v0 = u.x - pivot, v1 = u.y - pivot, v2 = u.z - pivot, v3 = u.w - pivot;
      s += (v0 + v1) + (v2 + v3);
      q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
    for (long long i = 4 * n4 + threadIdx.x; i < p.N; i += kSmallThreads) {
      const float v = xp[i] - pivot;
      s += v;
      q += v * v;
    }
  }
  double ds = s, dq = q;
  block_sum2_double(ds, dq);
  const double cnt = static_cast<double>(p.B) * static_cast<double>(p.N);
  const double m1 = ds / cnt;
  const double v = fmax(dq / cnt - m1 * m1, 0.0);
  const float m = static_cast<float>(m1 + static_cast<double>(pivot));
  const float vf = static_cast<float>(v);
  if (threadIdx.x == 0) {
    mean[c] = m;
    var[c] = vf;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
      const double unb = cnt > 1.0 ? v * cnt / (cnt - 1.0) : v;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * static_cast<float>(unb);
    }
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
  }
  const float is = rsqrtf(vf + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  const float sc = is * g, sh = be - m * sc;
  for (int b = 0; b < p.B; ++b) {
    const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
    float* op = out + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
    for (long long i = threadIdx.x; i < n4; i += kSmallThreads) {
      const float4 u = reinterpret_cast<const float4*>(xp)[i];
      reinterpret_cast<float4*>(op)[i] = make_float4(act_fwd(u.x * sc + sh, p.act), act_fwd(u.y * sc + sh, p.act),
                                                      act_fwd(u.z * sc + sh, p.act), act_fwd(u.w * sc + sh, p.act));
    }
    for (long long i = 4 * n4 + threadIdx.x; i < p.N; i += kSmallThreads) op[i] = act_fwd(xp[i] * sc + sh, p.act);
  }
}

// p.xb / p.xc: strides of x AND of dx;  p.ob / p.oc: strides of dy
template <bool VEC>
__global__ void __launch_bounds__(kSmallThreads)
bn_train_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                          const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta,
                          float* __restrict__ s1o, float* __restrict__ s2o, float* __restrict__ dx, const BNA p) {
  const int c = blockIdx.x;
  const float m = mean[c], is = rsqrtf(var[c] + p.eps);
  const float g = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
  float a = 0.f, bsum = 0.f;
  const long long n4 = VEC ? p.N / 4 : 0;
  auto term = [&](float xv, float gv, float& dz, float& xh) {
    xh = (xv - m) * is;
    dz = gv * act_grad(xh * g + be, p.act);
  };
  for (int b = 0; b < p.B; ++b) {
    const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
    const float* gp = dy + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
    for (long long i = threadIdx.x; i < n4; i += kSmallThreads) {
      const float4 u = reinterpret_cast<const float4*>(xp)[i], w = reinterpret_cast<const float4*>(gp)[i];
      float d0, d1, d2, d3, h0, h1, h2, h3;
      term(u.x, w.x, d0, h0); term(u.y, w.y, d1, h1); term(u.z, w.z, d2, h2); term(u.w, w.w, d3, h3);
      a += (d0 + d1) + (d2 + d3);
      bsum += (d0 * h0 + d1 * h1) + (d2 * h2 + d3 * h3);
    }
    for (long long i = 4 * n4 + threadIdx.x; i < p.N; i += kSmallThreads) {
      float dz, xh;
      term(xp[i], gp[i], dz, xh);
      a += dz;
      bsum += dz * xh;
    }
  }
  double da = a, db = bsum;
  block_sum2_double(da, db);
  const float s1 = static_cast<float>(da), s2 = static_cast<float>(db);
  if (threadIdx.x == 0) { s1o[c] = s1; s2o[c] = s2; }
  const float a1 = s1 * p.inv_n, a2 = s2 * p.inv_n;
  for (int b = 0; b < p.B; ++b) {
    const float* xp = x + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
    const float* gp = dy + static_cast<size_t>(b) * p.ob + static_cast<size_t>(c) * p.oc;
    float* op = dx + static_cast<size_t>(b) * p.xb + static_cast<size_t>(c) * p.xc;
    for (long long i = threadIdx.x; i < n4; i += kSmallThreads) {
      const float4 u = reinterpret_cast<const float4*>(xp)[i], w = reinterpret_cast<const float4*>(gp)[i];
      float d0, d1, d2, d3, h0, h1, h2, h3;
      term(u.x, w.x, d0, h0); term(u.y, w.y, d1, h1); term(u.z, w.z, d2, h2); term(u.w, w.w, d3, h3);
      reinterpret_cast<float4*>(op)[i] = make_float4((d0 - a1 - h0 * a2) * is * g, (d1 - a1 - h1 * a2) * is * g,
                                                      (d2 - a1 - h2 * a2) * is * g, (d3 - a1 - h3 * a2) * is * g);
    }
    for (long long i = 4 * n4 + threadIdx.x; i < p.N; i += kSmallThreads) {
      float dz, xh;
      term(xp[i], gp[i], dz, xh);
      op[i] = (dz - a1 - xh * a2) * is * g;
    }
  }
}

long long g_bn_small_elems = -1;
bool bn_vec(const void* a, const void* b, const void* c, long long N, long long s0, long long s1, long long s2, long long s3) {
  return N % 4 == 0 && s0 % 4 == 0 && s1 % 4 == 0 && s2 % 4 == 0 && s3 % 4 == 0 && ts::aligned16(a) && ts::aligned16(b) && (!c || ts::aligned16(c));
}

bool bn_small(int B, long long N) {
  if (g_bn_small_elems < 0) { const char* e = getenv("TS_BN_SMALL_ELEMS"); g_bn_small_elems = e ? atoll(e) : kSmallElems; }
  return static_cast<long long>(B) * N <= g_bn_small_elems;
}

int chunks_for(long long N, int B, int C, long long& chunk) {
  // enough workgroups to fill the chip (C * B * nchunk >= ~1024) without making chunks shorter than 1024 elements
  int n = static_cast<int>((1024 + static_cast<long long>(B) * C - 1) / (static_cast<long long>(B) * C));
  const long long maxn = (N + 1023) / 1024;
  if (n > maxn) n = static_cast<int>(maxn);
  if (n < 1) n = 1;
  chunk = ((N + n - 1) / n + 3) & ~3ll;          // a multiple of 4: the quad loops of the reduction kernels start every chunk aligned
  return n;
}

}  // namespace

// experiments: the size (B * N elements per channel) up to which ts_bn_train_{fwd,bwd} take their one-launch form; < 0 only queries
extern "C" long long ts_bn_set_small_elems(long long n) {
  (void)bn_small(1, 1);
  const long long old = g_bn_small_elems;
  if (n >= 0) g_bn_small_elems = n;
  return old;
}

// Per-channel sum of x [B,C,N] -> out [C]: the bias gradient of a convolution (sum of dy over batch and pixels), two deterministic
// stages on the partial-sum workspace of the BatchNorm kernels (ts_bn_workspace_bytes(B, C, N)).  The framework's reduction took
// 70 us for the [1,9,544,960] gradient of UNet.deconv2's bias (module.py:457) and 12-19 us for the small ones.
__global__ void __launch_bounds__(256)
channel_sum_partial(const float* __restrict__ x, float2* __restrict__ partial, const BN p) {
  const int k = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const float* xp = x + static_cast<size_t>(b) * p.bstride + static_cast<size_t>(c) * p.cstride;
  const long long lo = k * p.chunk, hi = min(p.N, lo + p.chunk);
  float s = 0.f, q = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) s += xp[i];
  block_sum2(s, q);
  if (threadIdx.x == 0) partial[(static_cast<size_t>(c) * p.B + b) * p.nchunk + k] = make_float2(s, 0.f);
}

__global__ void __launch_bounds__(64)
channel_sum_finish(const float2* __restrict__ partial, float* __restrict__ out, int n) {
  const int c = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) a += partial[static_cast<size_t>(c) * n + i].x;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (threadIdx.x == 0) out[c] = static_cast<float>(a);
}

// the same in ONE launch for small tensors (the bias gradients of the heads: 2 channels x 28,560 elements): a 1024-thread workgroup
// per channel, per-thread fp32 partials combined in double in a fixed order (as the one-launch BatchNorm forms)
__global__ void __launch_bounds__(kSmallThreads)
channel_sum_small_kernel(const float* __restrict__ x, float* __restrict__ out, const BN p) {
  const int c = blockIdx.x;
  float s = 0.f;
  for (int b = 0; b < p.B; ++b) {
    const float* xp = x + static_cast<size_t>(b) * p.bstride + static_cast<size_t>(c) * p.cstride;
    for (long long i = threadIdx.x; i < p.N; i += kSmallThreads) s += xp[i];
  }
  double a = static_cast<double>(s), z = 0.0;
  block_sum2_double(a, z);
  if (threadIdx.x == 0) out[c] = static_cast<float>(a);
}

extern "C" int ts_channel_sum_fwd(const float* x, float* out, void* workspace, int B, int C, long long N, long long bstride,
                                  long long cstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "channel_sum: bad size");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(out); TS_REQUIRE_PTR(workspace);
  if (static_cast<long long>(B) * N <= 32768) {
    const BN ps{B, C, N, bstride, cstride, 0, 0};
    hipLaunchKernelGGL(channel_sum_small_kernel, dim3(C), dim3(kSmallThreads), 0, ts::as_stream(stream), x, out, ps);
    return ts::launched("channel_sum_small_kernel");
  }
  BN p{B, C, N, bstride, cstride, 0, 0};
  p.nchunk = chunks_for(N, B, C, p.chunk);
  float2* partial = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(channel_sum_partial, dim3(p.nchunk, C, B), dim3(256), 0, ts::as_stream(stream), x, partial, p);
  if (int rc = ts::launched("channel_sum_partial")) return rc;
  hipLaunchKernelGGL(channel_sum_finish, dim3(C), dim3(64), 0, ts::as_stream(stream), partial, out, B * p.nchunk);
  return ts::launched("channel_sum_finish");
}

extern "C" size_t ts_bn_workspace_bytes(int B, int C, long long N) {
  if (B <= 0 || C <= 0 || N <= 0) return 0;
  long long chunk;
  const int n = chunks_for(N, B, C, chunk);
  return ts::round_up(static_cast<size_t>(B) * C * n * sizeof(float2), 256);
}

extern "C" int ts_bn_stats_fwd(const float* x, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                               long long* num_batches_tracked, void* workspace, int B, int C, long long N, long long bstride,
                               long long cstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "bn_stats: bad size");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(workspace);
  BN p{B, C, N, bstride, cstride, 0, 0};
  p.nchunk = chunks_for(N, B, C, p.chunk);
  float2* partial = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(bn_stats_partial, dim3(p.nchunk, C, B), dim3(256), 0, ts::as_stream(stream), x, partial, p,
                     bn_vec(x, nullptr, nullptr, N, bstride, cstride, 0, 0) ? 1 : 0);
  if (int rc = ts::launched("bn_stats_partial")) return rc;
  hipLaunchKernelGGL(bn_stats_finish, dim3(C), dim3(64), 0, ts::as_stream(stream), x, partial, mean, var, running_mean, running_var,
                     momentum, num_batches_tracked, p);
  return ts::launched("bn_stats_finish");
}

extern "C" int ts_bn_apply_act_fwd(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                                   float* out, int B, int C, long long N, long long x_bstride, long long x_cstride,
                                   long long out_bstride, long long out_cstride, float eps, int act, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "bn_apply: bad size");
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_UNSUPPORTED, "bn_apply: activation %d", act);
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(out);
  const BNA p{B, C, N, x_bstride, x_cstride, out_bstride, out_cstride, act, 0, eps, 0.f};
  hipLaunchKernelGGL(bn_apply_act_kernel, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0, ts::as_stream(stream),
                     x, mean, var, gamma, beta, out, p);
  return ts::launched("bn_apply_act_kernel");
}

extern "C" int ts_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float* sum_dz, float* sum_dz_xhat, void* workspace, int B, int C,
                                    long long N, long long x_bstride, long long x_cstride, long long dy_bstride,
                                    long long dy_cstride, float eps, int act, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "bn_bwd: bad size");
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_UNSUPPORTED, "bn_bwd: activation %d", act);
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(dy); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(sum_dz);
  TS_REQUIRE_PTR(sum_dz_xhat); TS_REQUIRE_PTR(workspace);
  const BNA p{B, C, N, x_bstride, x_cstride, dy_bstride, dy_cstride, act, 1, eps, 0.f};
  long long chunk;
  const int n = chunks_for(N, B, C, chunk);
  float2* partial = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(n, C, B), dim3(256), 0, ts::as_stream(stream), x, dy, mean, var, gamma, beta, partial,
                     p, n, chunk, bn_vec(x, dy, nullptr, N, x_bstride, x_cstride, dy_bstride, dy_cstride) ? 1 : 0);
  if (int rc = ts::launched("bn_bwd_reduce_kernel")) return rc;
  hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3(C), dim3(64), 0, ts::as_stream(stream), partial, sum_dz, sum_dz_xhat, B * n);
  return ts::launched("bn_bwd_finish_kernel");
}

extern "C" int ts_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* var, const float* gamma,
                                   const float* beta, const float* sum_dz, const float* sum_dz_xhat, float* dx, int B, int C,
                                   long long N, long long x_bstride, long long x_cstride, long long dy_bstride,
                                   long long dy_cstride, float eps, int act, int train, float count, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "bn_bwd: bad size");
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_UNSUPPORTED, "bn_bwd: activation %d", act);
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(dy); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(dx);
  if (train) { TS_REQUIRE_PTR(sum_dz); TS_REQUIRE_PTR(sum_dz_xhat); TS_REQUIRE(count > 0.f, TS_ERR_SHAPE, "bn_bwd: count"); }
  const BNA p{B, C, N, x_bstride, x_cstride, dy_bstride, dy_cstride, act, train, eps, train ? 1.f / count : 0.f};
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0, ts::as_stream(stream),
                     x, dy, mean, var, gamma, beta, sum_dz, sum_dz_xhat, dx, p);
  return ts::launched("bn_bwd_apply_kernel");
}

// SyncBatchNorm: combine the ranks' [mean | biased var | count] records (one all_gather) into the statistics of the whole batch with
// the parallel-variance formula, update the running statistics (unbiased variance of the global count) and leave 1 / count on the
// device -- the backward pass scales its all-reduced sums with it, so the host never reads a count (round 2 read one per layer).
__global__ void __launch_bounds__(64)
bn_sync_merge_kernel(const float* __restrict__ gathered, int world, int C, float* __restrict__ mean, float* __restrict__ var,
                     float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float* __restrict__ inv_count) {
  const int rec = 2 * C + 1;
  double n = 0.0;
  for (int r = 0; r < world; ++r) n += gathered[static_cast<size_t>(r) * rec + 2 * C];
  for (int c = threadIdx.x; c < C; c += 64) {
    double m = 0.0;
    for (int r = 0; r < world; ++r) m += static_cast<double>(gathered[static_cast<size_t>(r) * rec + c]) * gathered[static_cast<size_t>(r) * rec + 2 * C];
    m /= n;
    double v = 0.0;
    for (int r = 0; r < world; ++r) {
      const double d = gathered[static_cast<size_t>(r) * rec + c] - m;
      v += (gathered[static_cast<size_t>(r) * rec + C + c] + d * d) * gathered[static_cast<size_t>(r) * rec + 2 * C];
    }
    v /= n;
    mean[c] = static_cast<float>(m);
    var[c] = static_cast<float>(v);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * static_cast<float>(m);
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * static_cast<float>(v * (n / (n > 1.0 ? n - 1.0 : 1.0)));
  }
  if (threadIdx.x == 0 && inv_count) *inv_count = static_cast<float>(1.0 / n);
}

extern "C" int ts_bn_sync_merge(const float* gathered, int world, int C, float* mean, float* var, float* running_mean,
                                float* running_var, float momentum, float* inv_count, void* stream) {
  TS_REQUIRE(world > 0 && C > 0, TS_ERR_SHAPE, "bn_sync_merge: bad size");
  TS_REQUIRE_PTR(gathered); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var);
  hipLaunchKernelGGL(bn_sync_merge_kernel, dim3(1), dim3(64), 0, ts::as_stream(stream), gathered, world, C, mean, var, running_mean,
                     running_var, momentum, inv_count);
  return ts::launched("bn_sync_merge_kernel");
}

// Single-rank training form of the pair (ts_bn_stats_fwd, ts_bn_apply_act_fwd): two launches instead of three.
extern "C" int ts_bn_train_fwd(const float* x, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                               long long* num_batches_tracked, const float* gamma, const float* beta, float* out,
                               void* workspace, int B, int C, long long N, long long x_bstride, long long x_cstride,
                               long long out_bstride, long long out_cstride, float eps, int act, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535, TS_ERR_SHAPE, "bn_train_fwd: bad size");
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_UNSUPPORTED, "bn_train_fwd: activation %d", act);
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(out); TS_REQUIRE_PTR(workspace);
  if (bn_small(B, N)) {             // one launch: a workgroup per channel does the sums and the normalisation
    const BNA a{B, C, N, x_bstride, x_cstride, out_bstride, out_cstride, act, 1, eps, 0.f};
    if (bn_vec(x, out, nullptr, N, x_bstride, x_cstride, out_bstride, out_cstride))
      hipLaunchKernelGGL(bn_train_fwd_small_kernel<true>, dim3(C), dim3(kSmallThreads), 0, ts::as_stream(stream), x, mean, var, running_mean,
                         running_var, momentum, num_batches_tracked, gamma, beta, out, a);
    else
      hipLaunchKernelGGL(bn_train_fwd_small_kernel<false>, dim3(C), dim3(kSmallThreads), 0, ts::as_stream(stream), x, mean, var, running_mean,
                         running_var, momentum, num_batches_tracked, gamma, beta, out, a);
    return ts::launched("bn_train_fwd_small_kernel");
  }
  BN p{B, C, N, x_bstride, x_cstride, 0, 0};
  p.nchunk = chunks_for(N, B, C, p.chunk);
  float2* partial = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(bn_stats_partial, dim3(p.nchunk, C, B), dim3(256), 0, ts::as_stream(stream), x, partial, p,
                     bn_vec(x, nullptr, nullptr, N, x_bstride, x_cstride, 0, 0) ? 1 : 0);
  if (int rc = ts::launched("bn_stats_partial")) return rc;
  const BNA a{B, C, N, x_bstride, x_cstride, out_bstride, out_cstride, act, 1, eps, 0.f};
  if (bn_vec(x, out, nullptr, N, x_bstride, x_cstride, out_bstride, out_cstride))
    hipLaunchKernelGGL(bn_finish_apply_act_kernel<true>, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0,
                       ts::as_stream(stream), x, partial, B * p.nchunk, mean, var, running_mean, running_var, momentum,
                       num_batches_tracked, gamma, beta, out, a);
  else
    hipLaunchKernelGGL(bn_finish_apply_act_kernel<false>, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0,
                       ts::as_stream(stream), x, partial, B * p.nchunk, mean, var, running_mean, running_var, momentum,
                       num_batches_tracked, gamma, beta, out, a);
  return ts::launched("bn_finish_apply_act_kernel");
}

// Single-rank training form of the pair (ts_bn_act_bwd_reduce, ts_bn_act_bwd_apply with train = 1): two launches instead of three.
extern "C" int ts_bn_train_bwd(const float* x, const float* dy, const float* mean, const float* var, const float* gamma,
                               const float* beta, float* sum_dz, float* sum_dz_xhat, float* dx, void* workspace, int B, int C,
                               long long N, long long x_bstride, long long x_cstride, long long dy_bstride, long long dy_cstride,
                               float eps, int act, float count, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && C <= 65535 && B <= 65535 && count > 0.f, TS_ERR_SHAPE, "bn_train_bwd: bad size");
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_UNSUPPORTED, "bn_train_bwd: activation %d", act);
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(dy); TS_REQUIRE_PTR(mean); TS_REQUIRE_PTR(var); TS_REQUIRE_PTR(sum_dz);
  TS_REQUIRE_PTR(sum_dz_xhat); TS_REQUIRE_PTR(dx); TS_REQUIRE_PTR(workspace);
  const BNA p{B, C, N, x_bstride, x_cstride, dy_bstride, dy_cstride, act, 1, eps, 1.f / count};
  if (bn_small(B, N)) {
    if (bn_vec(x, dy, dx, N, x_bstride, x_cstride, dy_bstride, dy_cstride))
      hipLaunchKernelGGL(bn_train_bwd_small_kernel<true>, dim3(C), dim3(kSmallThreads), 0, ts::as_stream(stream), x, dy, mean, var, gamma,
                         beta, sum_dz, sum_dz_xhat, dx, p);
    else
      hipLaunchKernelGGL(bn_train_bwd_small_kernel<false>, dim3(C), dim3(kSmallThreads), 0, ts::as_stream(stream), x, dy, mean, var, gamma,
                         beta, sum_dz, sum_dz_xhat, dx, p);
    return ts::launched("bn_train_bwd_small_kernel");
  }
  long long chunk;
  const int n = chunks_for(N, B, C, chunk);
  float2* partial = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(n, C, B), dim3(256), 0, ts::as_stream(stream), x, dy, mean, var, gamma, beta, partial,
                     p, n, chunk, bn_vec(x, dy, nullptr, N, x_bstride, x_cstride, dy_bstride, dy_cstride) ? 1 : 0);
  if (int rc = ts::launched("bn_bwd_reduce_kernel")) return rc;
  if (bn_vec(x, dy, dx, N, x_bstride, x_cstride, dy_bstride, dy_cstride))
    hipLaunchKernelGGL(bn_bwd_finish_apply_kernel<true>, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0,
                       ts::as_stream(stream), x, dy, mean, var, gamma, beta, partial, B * n, sum_dz, sum_dz_xhat, dx, p);
  else
    hipLaunchKernelGGL(bn_bwd_finish_apply_kernel<false>, dim3(static_cast<unsigned>((N + 1023) / 1024), C, B), dim3(256), 0,
                       ts::as_stream(stream), x, dy, mean, var, gamma, beta, partial, B * n, sum_dz, sum_dz_xhat, dx, p);
  return ts::launched("bn_bwd_finish_apply_kernel");
}
