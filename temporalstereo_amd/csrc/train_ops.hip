// Training-side glue of the aggregation path as kernels (round 3): what was left to the framework's element-wise operators in a
// training step -- ~280 launches of 4-5 us each per T=2 step, 1.3 ms of a 14 ms replayed step -- plus the pieces that put the
// UNet decoder's two ConvTranspose2d(4, stride 2, padding 1) (module.py:453-457) on the convolution kernels in training, and
// the optimizer:
//
//   candidates_in_range   fine.py:82-87 / precise.py:73-78   |high - low| * {0,3,4,5,8}/8 + min(low, high), both ways
//   offset_head           module.py:384-390                  tanh(x / 100).clamp(-1, 1) * delta, both ways
//   space_to_depth2       z[(py,px,c)][y][x] = x[c][2y+py][2x+px]: the gradient of a k4-s2 transposed convolution w.r.t. its
//                         input is a 3x3 stride-1 convolution of this re-arrangement (four parity classes of two taps per axis)
//   deconv4 weight forms  W[ci][co][4][4] -> the [(p,co)][9][pad(ci)] weight of that 3x3 convolution, and the weight gradient
//                         of the same convolution gathered back into [ci][co][4][4]
//   clip + RMSprop        torch.nn.utils.clip_grad_norm_ (dist_train.py:94 gradient_clip_val=0.1) + torch.optim.RMSprop
//                         (sceneflow.yaml:21-24) over a pointer table of all parameters: two launches per step
#include "ts_common.hpp"

namespace {

unsigned grid_for(long long n, int threads, long long cap_blocks = 4096) {
  long long blocks = (n + threads - 1) / threads;
  if (blocks > cap_blocks) blocks = cap_blocks;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

__global__ void __launch_bounds__(256)
candidates_fwd_kernel(const float* __restrict__ low, const float* __restrict__ high, float* __restrict__ cand, int B, int HW,
                      int coff, int ctot) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int px = static_cast<int>(i - static_cast<long long>(b) * HW);
    const float lo = low[i], hi = high[i];
    const float span = fabsf(hi - lo), base = fminf(lo, hi);
    float* c = cand + (static_cast<size_t>(b) * ctot + coff) * HW + px;
    c[0] = span * 0.f + base;
    c[static_cast<size_t>(1) * HW] = span * 0.375f + base;
    c[static_cast<size_t>(2) * HW] = span * 0.5f + base;
    c[static_cast<size_t>(3) * HW] = span * 0.625f + base;
    c[static_cast<size_t>(4) * HW] = span * 1.f + base;
  }
}

// d/d(low, high): abs' = sign (0 at 0), min' = 1 on the smaller argument, 1/2 each at a tie (the framework's conventions)
__global__ void __launch_bounds__(256)
candidates_bwd_kernel(const float* __restrict__ low, const float* __restrict__ high, const float* __restrict__ gcand,
                      float* __restrict__ glow, float* __restrict__ ghigh, int B, int HW, int coff, int ctot) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int px = static_cast<int>(i - static_cast<long long>(b) * HW);
    const float lo = low[i], hi = high[i];
    const float* g = gcand + (static_cast<size_t>(b) * ctot + coff) * HW + px;
    const float g0 = g[0], g1 = g[static_cast<size_t>(1) * HW], g2 = g[static_cast<size_t>(2) * HW],
                g3 = g[static_cast<size_t>(3) * HW], g4 = g[static_cast<size_t>(4) * HW];
    const float gs = g1 * 0.375f + g2 * 0.5f + g3 * 0.625f + g4;       // d/d span
    const float gb = g0 + g1 + g2 + g3 + g4;                           // d/d base
    const float sg = hi > lo ? 1.f : (hi < lo ? -1.f : 0.f);
    const float wl = lo < hi ? 1.f : (lo > hi ? 0.f : 0.5f);
    glow[i] = -sg * gs + wl * gb;
    ghigh[i] = sg * gs + (1.f - wl) * gb;
  }
}

__global__ void __launch_bounds__(256)
offset_head_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float delta) {
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const float t = tanhf(x[i] / 100.f);
    y[i] = fminf(fmaxf(t, -1.f), 1.f) * delta;
  }
}

__global__ void __launch_bounds__(256)
offset_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ gx, long long n, float delta) {
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const float t = tanhf(x[i] / 100.f);          // |t| <= 1: the clamp is the identity and passes the gradient
    gx[i] = g[i] * delta * (1.f - t * t) / 100.f;
  }
}

// z [B][4C][H][W] from x [B][C][2H][2W]; one lane = 4 consecutive z pixels of one (parity, channel) plane <- 8 consecutive x pixels
__global__ void __launch_bounds__(256)
space_to_depth2_kernel(const float* __restrict__ x, float* __restrict__ z, int B, int C, int H, int W, int vec) {
  const int Wq = (W + 3) / 4;
  const long long n = static_cast<long long>(B) * C * 2 * H * Wq;         // (b, c, row of x = 2y+py, quad)
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int q = static_cast<int>(i % Wq);
    long long t = i / Wq;
    const int ry = static_cast<int>(t % (2 * H)); t /= 2 * H;
    const int c = static_cast<int>(t % C), b = static_cast<int>(t / C);
    const int y = ry >> 1, py = ry & 1;
    const float* src = x + ((static_cast<size_t>(b) * C + c) * 2 * H + ry) * 2 * W + 8 * q;
    float v[8];
    if (vec && 8 * q + 8 <= 2 * W) {         // vec: rows of x start on 16 bytes (2W a multiple of 4; odd W takes the scalar form)
      const float4 a = *reinterpret_cast<const float4*>(src), bb = *reinterpret_cast<const float4*>(src + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bb.x; v[5] = bb.y; v[6] = bb.z; v[7] = bb.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (8 * q + k < 2 * W) ? src[k] : 0.f;
    }
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      float* dst = z + ((static_cast<size_t>(b) * 4 * C + static_cast<size_t>(py * 2 + px) * C + c) * H + y) * W + 4 * q;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * q + k < W) dst[k] = v[2 * k + px];
    }
  }
}

// out[(p*Cout + co)][tap][ci] (ci < cpad, zero beyond Cin) = W[ci][co][ky][kx], ky = 2*oy + py + 1, kx = 2*ox + px + 1 with
// (oy, ox) = (tap / 3 - 1, tap % 3 - 1); taps whose (ky, kx) fall outside the 4x4 kernel are zero.
__global__ void __launch_bounds__(256)
deconv4_weight_to_conv3_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int cpad) {
  const long long n = static_cast<long long>(4) * Cout * 9 * cpad;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int ci = static_cast<int>(i % cpad);
    long long t = i / cpad;
    const int tap = static_cast<int>(t % 9); t /= 9;
    const int co = static_cast<int>(t % Cout), p = static_cast<int>(t / Cout);
    const int ky = 2 * (tap / 3 - 1) + (p >> 1) + 1, kx = 2 * (tap % 3 - 1) + (p & 1) + 1;
    float v = 0.f;
    if (ci < Cin && ky >= 0 && ky < 4 && kx >= 0 && kx < 4) v = w[((static_cast<size_t>(ci) * Cout + co) * 4 + ky) * 4 + kx];
    out[i] = v;
  }
}

// dW[ci][co][ky][kx] = dW3[ci][(p*Cout + co)][tap]: every (ky, kx) belongs to exactly one (parity, tap)
__global__ void __launch_bounds__(256)
deconv4_wgrad_from_conv3_kernel(const float* __restrict__ dw3, float* __restrict__ dw, int Cin, int Cout) {
  const long long n = static_cast<long long>(Cin) * Cout * 16;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int kx = static_cast<int>(i & 3), ky = static_cast<int>((i >> 2) & 3);
    const long long t = i >> 4;
    const int co = static_cast<int>(t % Cout), ci = static_cast<int>(t / Cout);
    const int py = (ky + 1) & 1, px = (kx + 1) & 1;                // ky = 2*oy + py + 1
    const int oy = (ky - py - 1) / 2, ox = (kx - px - 1) / 2;      // exact: the numerators are even (-2, 0 or 2)
    const int tap = (oy + 1) * 3 + (ox + 1);
    dw[i] = dw3[(static_cast<size_t>(ci) * 4 * Cout + static_cast<size_t>(py * 2 + px) * Cout + co) * 9 + tap];
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm folding (eval)
// scale[c] = gamma / sqrt(var + eps), shift[c] = beta + (bias - mean) * scale   (c < C; 0 beyond: the padded output channels)
struct FoldEntry {          // == ts_bn_fold_entry of include/ts_hip.h (64 bytes)
  const float* gamma; const float* beta; const float* mean; const float* var; const float* bias; float* scale; float* shift;
  int C, pad;
};
static_assert(sizeof(FoldEntry) == 64, "table entry layout");

__global__ void __launch_bounds__(64)
bn_fold_many_kernel(const FoldEntry* __restrict__ table, float eps) {
  const FoldEntry e = table[blockIdx.x];
  for (int c = threadIdx.x; c < e.pad; c += 64) {
    float s = 0.f, t = 0.f;
    if (c < e.C) {
      // var == NULL: no BatchNorm behind the convolution (scale 1, shift = bias)
      s = e.var ? (e.gamma ? e.gamma[c] : 1.f) / sqrtf(e.var[c] + eps) : 1.f;
      // (separately rounded product and sum, as the framework's `beta + (bias - mean) * s` of aggregation/native.Folded: an in-place
      // re-fold must give the bits a rebuild gives)
      t = __fadd_rn(e.beta ? e.beta[c] : 0.f, __fmul_rn((e.bias ? e.bias[c] : 0.f) - (e.mean ? e.mean[c] : 0.f), s));
    }
    e.scale[c] = s;
    e.shift[c] = t;
  }
}

// ------------------------------------------------------------------------------------------------ clip + RMSprop
struct OptEntry {           // == ts_opt_entry of include/ts_hip.h (32 bytes)
  float* param; const float* grad; float* square_avg; long long n;
};
static_assert(sizeof(OptEntry) == 32, "table entry layout");
constexpr int OPT_BLOCKS = 8;          // workgroups per tensor (grid.x); tensors are grid.y

__global__ void __launch_bounds__(256)
opt_sumsq_kernel(const OptEntry* __restrict__ table, float* __restrict__ partial) {
  const OptEntry e = table[blockIdx.y];
  float acc = 0.f;
  if (e.grad != nullptr)
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < e.n; i += static_cast<long long>(gridDim.x) * blockDim.x) {
      const float g = e.grad[i];
      acc += g * g;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// every workgroup re-derives the total norm from the partials in the same fixed order (deterministic), then updates its slice:
//   g *= min(1, max_norm / (norm + 1e-6))                 clip_grad_norm_
//   v = alpha v + (1 - alpha) g^2 ;  p -= lr g / (sqrt(v) + eps)        RMSprop (no momentum, not centred, no weight decay)
__global__ void __launch_bounds__(256)
opt_clip_rmsprop_kernel(const OptEntry* __restrict__ table, const float* __restrict__ partial, int n_partial, float max_norm, float lr,
                        float alpha, float eps, float* __restrict__ norm_out) {
  __shared__ float s[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  const float norm = sqrtf(s[0]);
  if (norm_out && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *norm_out = norm;
  const float coef = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  const OptEntry e = table[blockIdx.y];
  if (e.grad == nullptr) return;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < e.n; i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const float g = e.grad[i] * coef;
    const float v = alpha * e.square_avg[i] + (1.f - alpha) * g * g;
    e.square_avg[i] = v;
    e.param[i] -= lr * g / (sqrtf(v) + eps);
  }
}

}  // namespace

extern "C" int ts_candidates_in_range_fwd(const float* low, const float* high, float* candidates, int B, int H, int W,
                                          int channel_offset, int channels_total, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && channel_offset >= 0 && channel_offset + 5 <= channels_total, TS_ERR_SHAPE, "candidates_in_range: bad size");
  TS_REQUIRE_PTR(low); TS_REQUIRE_PTR(high); TS_REQUIRE_PTR(candidates);
  hipLaunchKernelGGL(candidates_fwd_kernel, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0, ts::as_stream(stream),
                     low, high, candidates, B, H * W, channel_offset, channels_total);
  return ts::launched("candidates_fwd_kernel");
}

extern "C" int ts_candidates_in_range_bwd(const float* low, const float* high, const float* grad_candidates, float* grad_low,
                                          float* grad_high, int B, int H, int W, int channel_offset, int channels_total, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && channel_offset >= 0 && channel_offset + 5 <= channels_total, TS_ERR_SHAPE, "candidates_in_range_bwd: bad size");
  TS_REQUIRE_PTR(low); TS_REQUIRE_PTR(high); TS_REQUIRE_PTR(grad_candidates); TS_REQUIRE_PTR(grad_low); TS_REQUIRE_PTR(grad_high);
  hipLaunchKernelGGL(candidates_bwd_kernel, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0, ts::as_stream(stream),
                     low, high, grad_candidates, grad_low, grad_high, B, H * W, channel_offset, channels_total);
  return ts::launched("candidates_bwd_kernel");
}

extern "C" int ts_offset_head_fwd(const float* x, float* y, long long n, float delta, void* stream) {
  TS_REQUIRE(n > 0, TS_ERR_SHAPE, "offset_head: empty tensor");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(y);
  hipLaunchKernelGGL(offset_head_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ts::as_stream(stream), x, y, n, delta);
  return ts::launched("offset_head_fwd_kernel");
}

extern "C" int ts_offset_head_bwd(const float* x, const float* grad_y, float* grad_x, long long n, float delta, void* stream) {
  TS_REQUIRE(n > 0, TS_ERR_SHAPE, "offset_head_bwd: empty tensor");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(grad_y); TS_REQUIRE_PTR(grad_x);
  hipLaunchKernelGGL(offset_head_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ts::as_stream(stream), x, grad_y, grad_x, n, delta);
  return ts::launched("offset_head_bwd_kernel");
}

extern "C" int ts_space_to_depth2_fwd(const float* x, float* z, int B, int C, int H, int W, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "space_to_depth2: non-positive size");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(z);
  TS_REQUIRE_ALIGNED(x);
  const long long n = static_cast<long long>(B) * C * 2 * H * ((W + 3) / 4);
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for(n, 256, 1 << 16)), dim3(256), 0, ts::as_stream(stream), x, z, B, C, H, W,
                     (2 * W) % 4 == 0 ? 1 : 0);
  return ts::launched("space_to_depth2_kernel");
}

extern "C" int ts_deconv2d_k4s2_weight_to_conv3(const float* w, float* out, int Cin, int Cout, int cin_pad, void* stream) {
  TS_REQUIRE(Cin > 0 && Cout > 0 && cin_pad >= Cin, TS_ERR_SHAPE, "deconv2d_k4s2_weight_to_conv3: bad size");
  TS_REQUIRE_PTR(w); TS_REQUIRE_PTR(out);
  const long long n = 4ll * Cout * 9 * cin_pad;
  hipLaunchKernelGGL(deconv4_weight_to_conv3_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ts::as_stream(stream), w, out, Cin, Cout, cin_pad);
  return ts::launched("deconv4_weight_to_conv3_kernel");
}

extern "C" int ts_deconv2d_k4s2_wgrad_from_conv3(const float* dw3, float* dw, int Cin, int Cout, void* stream) {
  TS_REQUIRE(Cin > 0 && Cout > 0, TS_ERR_SHAPE, "deconv2d_k4s2_wgrad_from_conv3: bad size");
  TS_REQUIRE_PTR(dw3); TS_REQUIRE_PTR(dw);
  hipLaunchKernelGGL(deconv4_wgrad_from_conv3_kernel, dim3(grid_for(16ll * Cin * Cout, 256)), dim3(256), 0, ts::as_stream(stream), dw3, dw, Cin, Cout);
  return ts::launched("deconv4_wgrad_from_conv3_kernel");
}

// table: n entries of ts_bn_fold_entry in DEVICE memory; one launch folds every eval-mode BatchNorm of a step into the
// per-channel scale / shift the convolution kernels apply in their epilogue (what aggregation/native.py does once per model,
// here once per training step for the frames that run in eval mode: projects/TemporalStereo/TemporalStereo.py:268-274)
extern "C" int ts_bn_fold_many(const void* table, int n, float eps, void* stream) {
  TS_REQUIRE(n > 0 && n <= 65535, TS_ERR_SHAPE, "bn_fold_many: %d entries", n);
  TS_REQUIRE_PTR(table);
  hipLaunchKernelGGL(bn_fold_many_kernel, dim3(n), dim3(64), 0, ts::as_stream(stream), static_cast<const FoldEntry*>(table), eps);
  return ts::launched("bn_fold_many_kernel");
}

extern "C" size_t ts_clip_rmsprop_workspace_bytes(int n_tensors) {
  return n_tensors > 0 ? static_cast<size_t>(n_tensors) * OPT_BLOCKS * sizeof(float) + 16 : 0;
}

// table: n_tensors entries of ts_opt_entry in DEVICE memory (grad NULL = the parameter received no gradient: skipped, as the
// framework's optimizers do).  workspace: ts_clip_rmsprop_workspace_bytes; its last float receives the total gradient norm.
extern "C" int ts_clip_rmsprop_step(const void* table, int n_tensors, float max_norm, float lr, float alpha, float eps,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  TS_REQUIRE(n_tensors > 0 && n_tensors <= 65535, TS_ERR_SHAPE, "clip_rmsprop_step: %d tensors", n_tensors);
  TS_REQUIRE_PTR(table); TS_REQUIRE_PTR(workspace);
  TS_REQUIRE(workspace_bytes >= ts_clip_rmsprop_workspace_bytes(n_tensors), TS_ERR_SHAPE, "clip_rmsprop_step: workspace too small");
  TS_REQUIRE(lr >= 0.f && alpha >= 0.f && alpha <= 1.f && eps >= 0.f, TS_ERR_SHAPE, "clip_rmsprop_step: bad hyper-parameter");
  float* partial = static_cast<float*>(workspace);
  hipStream_t st = ts::as_stream(stream);
  hipLaunchKernelGGL(opt_sumsq_kernel, dim3(OPT_BLOCKS, n_tensors), dim3(256), 0, st, static_cast<const OptEntry*>(table), partial);
  if (int rc = ts::launched("opt_sumsq_kernel")) return rc;
  hipLaunchKernelGGL(opt_clip_rmsprop_kernel, dim3(OPT_BLOCKS, n_tensors), dim3(256), 0, st, static_cast<const OptEntry*>(table), partial,
                     n_tensors * OPT_BLOCKS, max_norm, lr, alpha, eps, partial + n_tensors * OPT_BLOCKS);
  return ts::launched("opt_clip_rmsprop_kernel");
}
