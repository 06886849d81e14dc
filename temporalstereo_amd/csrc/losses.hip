// Loss-side kernels of the hot path's outputs, gfx950.
//
// The untouched losses of the reference consume what the aggregation returns:
//   WarssersteinDistanceLoss.loss_per_level   architecture/modeling/losses/warsserstein_distance_loss.py:52-78
//       mean_{b,y,x} sum_d (softmax_d(cost) + 0.25) * |off_d + sample_d - gt'| * mask,
//       gt' = adaptive_{avg,max}_pool2d(gt / scale, (H, W)), scale = Wg / W, mask = start < gt' < max_disp / scale
//   DispSmoothL1Loss.loss_per_level           architecture/modeling/losses/smooth_l1_loss.py:49-76
//       on disparities that the model wrapper first rescales to full resolution,
//       F.interpolate(d * full_w / dw, (full_h, full_w), bilinear, align_corners=True)   projects/TemporalStereo/TemporalStereo.py:305-309
// As torch ops that is, per level, softmax + 6 element-wise volumes of [B,D,H,W] + pooling + reductions (and a full-resolution
// copy of every disparity).  Here each level is one pass over its inputs per direction:
//   wasserstein_fwd   one lane per pixel: pooled ground truth, two passes over the D costs in L2 (max, then sums), per-workgroup
//                     partial sums; a one-workgroup finish adds them in a fixed order (deterministic mean)
//   wasserstein_bwd   per pixel again: grad cost_d = G p_d (a_d - sum_j p_j a_j), grad off_d = G (p_d + 0.25) sign(.)
//   smooth_l1_fwd     one lane per 4 full-resolution pixels: the bilinear rescale of the low-resolution disparity is evaluated
//                     in registers (never written), smooth-L1 against gt under the validity mask, partial sum + count
//   smooth_l1_bwd     one lane per LOW-resolution pixel gathers the full-resolution pixels whose bilinear footprint contains it
//                     (the adjoint of the rescale without atomics: deterministic)
// All HBM/latency-bound and tiny next to K1/K3; what they remove is ~40 framework launches per training step.
#include "ts_common.hpp"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sums of up to two values -> thread 0 (256 threads = 4 waves)
__device__ __forceinline__ void block_sum2(float& a, float& b) {
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

// adaptive_{avg,max}_pool2d bin of output index i: [floor(i*in/out), ceil((i+1)*in/out))
__device__ __forceinline__ void bin(int i, int in, int out, int& lo, int& hi) {
  lo = static_cast<int>((static_cast<long long>(i) * in) / out);
  hi = static_cast<int>((static_cast<long long>(i + 1) * in + out - 1) / out);
}

// gt' of one pixel (warsserstein_distance_loss.py:56-62): gt / scale pooled to (H, W); identity when the sizes agree
__device__ __forceinline__ float pooled_gt(const float* __restrict__ gt, int y, int x, int H, int W, int Hg, int Wg, float scale,
                                           int sparse) {
  if (Hg == H && Wg == W) return gt[static_cast<size_t>(y) * Wg + x];
  int y0, y1, x0, x1;
  bin(y, Hg, H, y0, y1);
  bin(x, Wg, W, x0, x1);
  float acc = sparse ? -INFINITY : 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float v = gt[static_cast<size_t>(yy) * Wg + xx] / scale;
      acc = sparse ? fmaxf(acc, v) : acc + v;
    }
  return sparse ? acc : acc / static_cast<float>((y1 - y0) * (x1 - x0));
}

__global__ void __launch_bounds__(256)
wasserstein_fwd_kernel(const float* __restrict__ cost, const float* __restrict__ off, const float* __restrict__ samp,
                       const float* __restrict__ gt, float* __restrict__ gt_scaled, float* __restrict__ partial, int B, int D,
                       int H, int W, int Hg, int Wg, float scale, float lo, float hi, int sparse) {
  const long long n = static_cast<long long>(B) * H * W;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  float v = 0.f, unused = 0.f;
  if (i < n) {
    const int x = static_cast<int>(i % W);
    const long long t = i / W;
    const int y = static_cast<int>(t % H), b = static_cast<int>(t / H);
    const float g = pooled_gt(gt + static_cast<size_t>(b) * Hg * Wg, y, x, H, W, Hg, Wg, scale, sparse);
    gt_scaled[i] = g;
    const bool valid = (g > lo) && (g < hi);
    const size_t HW = static_cast<size_t>(H) * W;
    const size_t base = static_cast<size_t>(b) * D * HW + static_cast<size_t>(y) * W + x;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, cost[base + d * HW]);
    float se = 0.f, sea = 0.f, sa = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = __expf(cost[base + d * HW] - m);
      const float a = fabsf(off[base + d * HW] + samp[base + d * HW] - g);
      se += e;
      sea += e * a;
      sa += a;
    }
    v = valid ? (sea / se + 0.25f * sa) : 0.f;
  }
  block_sum2(v, unused);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

// fixed-order sum of the partials (one workgroup): out[0] = sum / denom
__global__ void __launch_bounds__(256)
finish_mean_kernel(const float* __restrict__ partial, int n, float denom, float* __restrict__ out) {
  float a = 0.f, unused = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
  block_sum2(a, unused);
  if (threadIdx.x == 0) out[0] = a / denom;
}

__global__ void __launch_bounds__(256)
wasserstein_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ off, const float* __restrict__ samp,
                       const float* __restrict__ gt_scaled, const float* __restrict__ grad_loss, float* __restrict__ gcost,
                       float* __restrict__ goff, int B, int D, int H, int W, float lo, float hi) {
  const long long n = static_cast<long long>(B) * H * W;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = static_cast<int>(i % W);
  const long long t = i / W;
  const int y = static_cast<int>(t % H), b = static_cast<int>(t / H);
  const float g = gt_scaled[i];
  const bool valid = (g > lo) && (g < hi);
  const float G = valid ? grad_loss[0] / static_cast<float>(n) : 0.f;
  const size_t HW = static_cast<size_t>(H) * W;
  const size_t base = static_cast<size_t>(b) * D * HW + static_cast<size_t>(y) * W + x;
  float m = -INFINITY;
  for (int d = 0; d < D; ++d) m = fmaxf(m, cost[base + d * HW]);
  float se = 0.f, sea = 0.f;
  for (int d = 0; d < D; ++d) {
    const float e = __expf(cost[base + d * HW] - m);
    se += e;
    sea += e * fabsf(off[base + d * HW] + samp[base + d * HW] - g);
  }
  const float inv = 1.f / se, A = sea * inv;
  for (int d = 0; d < D; ++d) {
    const float p = __expf(cost[base + d * HW] - m) * inv;
    const float r = off[base + d * HW] + samp[base + d * HW] - g;
    const float sg = r > 0.f ? 1.f : (r < 0.f ? -1.f : 0.f);
    if (gcost) gcost[base + d * HW] = G * p * (fabsf(r) - A);
    if (goff) goff[base + d * HW] = G * (p + 0.25f) * sg;
  }
}

// align_corners bilinear source of one output index (the arithmetic of ts_resize_bilinear_fwd)
__device__ __forceinline__ void lin_src(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  const float s = scale * static_cast<float>(dst);
  i0 = static_cast<int>(s);
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - static_cast<float>(i0);
}

__device__ __forceinline__ float rescaled(const float* __restrict__ e, int h, int w, float sh, float sw, float vs, int oy, int ox) {
  int y0, y1, x0, x1;
  float ly, lx;
  lin_src(sh, oy, h, y0, y1, ly);
  lin_src(sw, ox, w, x0, x1, lx);
  const float top = (1.f - lx) * e[y0 * w + x0] + lx * e[y0 * w + x1];
  const float bot = (1.f - lx) * e[y1 * w + x0] + lx * e[y1 * w + x1];
  return ((1.f - ly) * top + ly * bot) * vs;
}

__global__ void __launch_bounds__(256)
smooth_l1_fwd_kernel(const float* __restrict__ est, const float* __restrict__ gt, float* __restrict__ partial_sum,
                     float* __restrict__ partial_cnt, int B, int h, int w, int Hg, int Wg, float sh, float sw, float vs, float lo,
                     float hi) {
  const long long n = static_cast<long long>(B) * Hg * Wg;
  float s = 0.f, c = 0.f;
  for (int k = 0; k < 4; ++k) {                       // 4 pixels per lane, a stride of the grid apart (coalesced)
    const long long i = (static_cast<long long>(blockIdx.x) * 4 + k) * blockDim.x + threadIdx.x;
    if (i < n) {
      const int ox = static_cast<int>(i % Wg);
      const long long t = i / Wg;
      const int oy = static_cast<int>(t % Hg), b = static_cast<int>(t / Hg);
      const float g = gt[i];
      if (g > lo && g < hi) {
        const float e = (h == Hg && w == Wg) ? est[i] : rescaled(est + static_cast<size_t>(b) * h * w, h, w, sh, sw, vs, oy, ox);
        const float r = fabsf(e - g);
        s += r < 1.f ? 0.5f * r * r : r - 0.5f;       // F.smooth_l1_loss, beta = 1
        c += 1.f;
      }
    }
  }
  block_sum2(s, c);
  if (threadIdx.x == 0) { partial_sum[blockIdx.x] = s; partial_cnt[blockIdx.x] = c; }
}

// out[0] = sum(partial_sum) / count (0 when nothing is valid: the reference's fallback branch evaluates to 0), out[1] = count
__global__ void __launch_bounds__(256)
finish_ratio_kernel(const float* __restrict__ ps, const float* __restrict__ pc, int n, float* __restrict__ out) {
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += ps[i]; c += pc[i]; }
  block_sum2(a, c);
  if (threadIdx.x == 0) { out[0] = c > 0.f ? a / c : 0.f; out[1] = c; }
}

// d loss / d est (low resolution): sum over the full-resolution pixels whose bilinear footprint contains (y, x)
__global__ void __launch_bounds__(256)
smooth_l1_bwd_kernel(const float* __restrict__ est, const float* __restrict__ gt, const float* __restrict__ grad_loss,
                     const float* __restrict__ loss_count, float* __restrict__ gest, int B, int h, int w, int Hg, int Wg, float sh,
                     float sw, float vs, float lo, float hi, int lpp) {
  // lpp lanes per low-resolution pixel (1 at equal sizes, else 16): its bilinear footprint is up to (2 Hg/h + 1)^2 full-resolution
  // pixels -- 289 at the 1/8 level -- and one lane walking it alone left the 1/8 and 1/16 levels on 8-32 workgroups for ~95 us;
  // the lanes take the footprint's rows in turn and meet in a fixed-order shuffle tree (still deterministic).
  const long long n = static_cast<long long>(B) * h * w;
  const long long gi = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long i = gi / lpp;
  const int sub = static_cast<int>(gi - i * lpp);
  if (i >= n) return;
  const int x = static_cast<int>(i % w);
  const long long t = i / w;
  const int y = static_cast<int>(t % h), b = static_cast<int>(t / h);
  const float cnt = loss_count[1];
  const float G = cnt > 0.f ? grad_loss[0] / cnt : 0.f;
  const float* e = est + static_cast<size_t>(b) * h * w;
  const float* g = gt + static_cast<size_t>(b) * Hg * Wg;
  if (h == Hg && w == Wg) {
    const float gv = g[static_cast<size_t>(y) * w + x];
    const float r = e[static_cast<size_t>(y) * w + x] - gv;
    gest[i] = (gv > lo && gv < hi) ? G * fminf(fmaxf(r, -1.f), 1.f) : 0.f;
    return;
  }
  // full-resolution rows whose source row lies in (y-1, y+1): oy*sh in (y-1, y+1)
  const float ih = sh > 0.f ? 1.f / sh : 0.f, iw = sw > 0.f ? 1.f / sw : 0.f;
  int oy0 = sh > 0.f ? static_cast<int>(floorf((static_cast<float>(y) - 1.f) * ih)) : 0;
  int oy1 = sh > 0.f ? static_cast<int>(ceilf((static_cast<float>(y) + 1.f) * ih)) : Hg - 1;
  int ox0 = sw > 0.f ? static_cast<int>(floorf((static_cast<float>(x) - 1.f) * iw)) : 0;
  int ox1 = sw > 0.f ? static_cast<int>(ceilf((static_cast<float>(x) + 1.f) * iw)) : Wg - 1;
  oy0 = max(oy0, 0); oy1 = min(oy1, Hg - 1); ox0 = max(ox0, 0); ox1 = min(ox1, Wg - 1);
  float acc = 0.f;
  for (int oy = oy0 + sub; oy <= oy1; oy += lpp) {
    int y0, y1;
    float ly;
    lin_src(sh, oy, h, y0, y1, ly);
    const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox0; ox <= ox1; ++ox) {
      int x0, x1;
      float lx;
      lin_src(sw, ox, w, x0, x1, lx);
      const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
      if (wx == 0.f) continue;
      const float gv = g[static_cast<size_t>(oy) * Wg + ox];
      if (!(gv > lo && gv < hi)) continue;
      const float top = (1.f - lx) * e[y0 * w + x0] + lx * e[y0 * w + x1];
      const float bot = (1.f - lx) * e[y1 * w + x0] + lx * e[y1 * w + x1];
      const float r = ((1.f - ly) * top + ly * bot) * vs - gv;
      acc += fminf(fmaxf(r, -1.f), 1.f) * wy * wx;
    }
  }
  for (int o = lpp >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (sub == 0) gest[i] = G * vs * acc;
}

inline float ac_scale(int in_size, int out_size) {
  return out_size > 1 ? static_cast<float>(in_size - 1) / static_cast<float>(out_size - 1) : 0.f;
}

inline int blocks_for(long long n, int per_block) { return static_cast<int>((n + per_block - 1) / per_block); }

}  // namespace

extern "C" size_t ts_wasserstein_loss_workspace_bytes(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return ts::round_up(static_cast<size_t>(blocks_for(static_cast<long long>(B) * H * W, 256)) * sizeof(float), 256);
}

extern "C" int ts_wasserstein_loss_fwd(const float* cost, const float* offset, const float* sample, const float* gt, float* loss,
                                       float* gt_scaled, void* workspace, int B, int D, int H, int W, int Hg, int Wg,
                                       float max_disp, float start_disp, int sparse, void* stream) {
  TS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Hg >= H && Wg >= W, TS_ERR_SHAPE, "wasserstein_loss: bad size");
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(offset); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(gt); TS_REQUIRE_PTR(loss);
  TS_REQUIRE_PTR(gt_scaled); TS_REQUIRE_PTR(workspace);
  const bool same = (Hg == H && Wg == W);
  const float scale = same ? 1.f : static_cast<float>(Wg) / (static_cast<float>(W) * 1.0f);
  const long long n = static_cast<long long>(B) * H * W;
  const int nb = blocks_for(n, 256);
  float* partial = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(wasserstein_fwd_kernel, dim3(nb), dim3(256), 0, ts::as_stream(stream), cost, offset, sample, gt, gt_scaled,
                     partial, B, D, H, W, Hg, Wg, scale, start_disp, max_disp / scale, sparse);
  if (int rc = ts::launched("wasserstein_fwd_kernel")) return rc;
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(256), 0, ts::as_stream(stream), partial, nb, static_cast<float>(n), loss);
  return ts::launched("finish_mean_kernel");
}

extern "C" int ts_wasserstein_loss_bwd(const float* cost, const float* offset, const float* sample, const float* gt_scaled,
                                       const float* grad_loss, float* grad_cost, float* grad_offset, int B, int D, int H, int W,
                                       int Wg, float max_disp, float start_disp, void* stream) {
  TS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Wg >= W, TS_ERR_SHAPE, "wasserstein_loss: bad size");
  TS_REQUIRE_PTR(cost); TS_REQUIRE_PTR(offset); TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(gt_scaled); TS_REQUIRE_PTR(grad_loss);
  const float scale = (Wg == W) ? 1.f : static_cast<float>(Wg) / (static_cast<float>(W) * 1.0f);
  const long long n = static_cast<long long>(B) * H * W;
  hipLaunchKernelGGL(wasserstein_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, ts::as_stream(stream), cost, offset, sample,
                     gt_scaled, grad_loss, grad_cost, grad_offset, B, D, H, W, start_disp, max_disp / scale);
  return ts::launched("wasserstein_bwd_kernel");
}

extern "C" size_t ts_disp_smooth_l1_workspace_bytes(int B, int Hg, int Wg) {
  if (B <= 0 || Hg <= 0 || Wg <= 0) return 0;
  return ts::round_up(2 * static_cast<size_t>(blocks_for(static_cast<long long>(B) * Hg * Wg, 1024)) * sizeof(float), 256);
}

extern "C" int ts_disp_smooth_l1_fwd(const float* est, const float* gt, float* loss_count, void* workspace, int B, int h, int w,
                                     int Hg, int Wg, float max_disp, float start_disp, void* stream) {
  TS_REQUIRE(B > 0 && h > 0 && w > 0 && Hg >= h && Wg >= w, TS_ERR_SHAPE, "disp_smooth_l1: bad size");
  TS_REQUIRE_PTR(est); TS_REQUIRE_PTR(gt); TS_REQUIRE_PTR(loss_count); TS_REQUIRE_PTR(workspace);
  const long long n = static_cast<long long>(B) * Hg * Wg;
  const int nb = blocks_for(n, 1024);
  float* ps = reinterpret_cast<float*>(workspace);
  float* pc = ps + nb;
  const float vs = static_cast<float>(Wg) / static_cast<float>(w);          // d * full_w / dw (TemporalStereo.py:307)
  hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(nb), dim3(256), 0, ts::as_stream(stream), est, gt, ps, pc, B, h, w, Hg, Wg,
                     ac_scale(h, Hg), ac_scale(w, Wg), vs, start_disp, max_disp);
  if (int rc = ts::launched("smooth_l1_fwd_kernel")) return rc;
  hipLaunchKernelGGL(finish_ratio_kernel, dim3(1), dim3(256), 0, ts::as_stream(stream), ps, pc, nb, loss_count);
  return ts::launched("finish_ratio_kernel");
}

extern "C" int ts_disp_smooth_l1_bwd(const float* est, const float* gt, const float* grad_loss, const float* loss_count,
                                     float* grad_est, int B, int h, int w, int Hg, int Wg, float max_disp, float start_disp,
                                     void* stream) {
  TS_REQUIRE(B > 0 && h > 0 && w > 0 && Hg >= h && Wg >= w, TS_ERR_SHAPE, "disp_smooth_l1: bad size");
  TS_REQUIRE_PTR(est); TS_REQUIRE_PTR(gt); TS_REQUIRE_PTR(grad_loss); TS_REQUIRE_PTR(loss_count); TS_REQUIRE_PTR(grad_est);
  const long long n = static_cast<long long>(B) * h * w;
  const float vs = static_cast<float>(Wg) / static_cast<float>(w);
  const int lpp = (h == Hg && w == Wg) ? 1 : 16;
  hipLaunchKernelGGL(smooth_l1_bwd_kernel, dim3(blocks_for(n * lpp, 256)), dim3(256), 0, ts::as_stream(stream), est, gt, grad_loss,
                     loss_count, grad_est, B, h, w, Hg, Wg, ac_scale(h, Hg), ac_scale(w, Wg), vs, start_disp, max_disp, lpp);
  return ts::launched("smooth_l1_bwd_kernel");
}
