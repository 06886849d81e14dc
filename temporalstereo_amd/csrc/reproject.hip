// K2c -- the whole temporal state update of one frame in three launches (gfx950 / MI355X).
//
//   ts_reproject_memory_fwd  replaces the closures of update_map,
//        projects/TemporalStereo/TemporalStereo.py:326-461 (update_local_map :340-384, update_past_cost :386-426):
//        previous disparity -> 1/8 grid (bilinear, align_corners, value scale)      :357-359 / :404-406
//        intrinsics of the 1/8 grid, their inverse, pose composition                :333-338, :349-355
//        disparity -> depth -> rigid motion -> depth -> disparity for every plane   :361-371 / :408-414
//        soft-max splat of [moved candidates | their costs | moved local maps]     :373-379 / :415-419
//
// The reference issues ~60 framework ops for this (two project_to_3d, two softsplats, inverse, bmm,
// resizes, cats); every tensor is B x <=9 x 68 x 120, so the device work is a few microseconds and the
// cost is the launches.  Here: `prepare` (resize + partial sums of the mean + zeroing of the
// accumulator), `scatter` (mean, matrices, re-projection and atomically accumulated splat of all
// planes, which share one flow and one metric), `normalize`.  The global mean of the metric
// (TemporalStereo.py:374: prev_disp.mean() over batch and pixels) is what forces the first barrier.
#include "ts_common.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxParts = 1024;

struct Reproj {
  int B, Hf, Wf, h, w, k, nl_in, nl_out, kdim, nparts;
  long long disp_bstride;
  float sh, sw;                 // align_corners source scales
  float factor;                 // full_w / w
  float baseline;               // used when baseline_ptr is null
};

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < kThreads / 64; ++i) t += red[i];      // same order in every thread
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(kThreads)
reproject_prepare(const float* __restrict__ disp, float* __restrict__ pd, float* __restrict__ parts,
                  float* __restrict__ accum, long long accum_elems, Reproj g) {
  __shared__ float red[kThreads / 64];
  const int hw = g.h * g.w;
  const long long n = static_cast<long long>(g.B) * hw;
  float local = 0.f;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * kThreads) {
    const int b = static_cast<int>(i / hw);
    const int p = static_cast<int>(i - static_cast<long long>(b) * hw);
    const int oy = p / g.w, ox = p - oy * g.w;
    const float sy = g.sh * static_cast<float>(oy), sx = g.sw * static_cast<float>(ox);
    const int y0 = static_cast<int>(sy), x0 = static_cast<int>(sx);
    const int y1 = y0 + (y0 < g.Hf - 1), x1 = x0 + (x0 < g.Wf - 1);
    const float ly = sy - static_cast<float>(y0), lx = sx - static_cast<float>(x0);
    const float* s = disp + static_cast<size_t>(b) * g.disp_bstride;
    const float vw = static_cast<float>(g.w), vf = static_cast<float>(g.Wf);
    auto at = [&](int y, int x) { return s[static_cast<size_t>(y) * g.Wf + x] * vw / vf; };   // disp * w / W, :357
    const float top = (1.f - lx) * at(y0, x0) + lx * at(y0, x1);
    const float bot = (1.f - lx) * at(y1, x0) + lx * at(y1, x1);
    const float v = (1.f - ly) * top + ly * bot;
    pd[i] = v;
    local += v;
  }
  const float t = block_sum(local, red);
  if (threadIdx.x == 0) parts[blockIdx.x] = t;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < accum_elems;
       i += static_cast<long long>(gridDim.x) * kThreads)
    accum[i] = 0.f;
}

struct Cam {
  float P[3][4];       // (K4 * T)[:3]
  float iK[3][3];      // inverse of the scaled intrinsics
  float bf;            // baseline * focal length of the 1/8 grid
};

// thread 0 of a block: matrices of batch b (inverse_warp.py:138-146 for P; TemporalStereo.py:333-338,349-355)
__device__ void make_cam(Cam& cam, const float* __restrict__ K, const float* __restrict__ Ta, const float* __restrict__ Tb,
                         const float* __restrict__ baseline_ptr, int b, const Reproj& g) {
  float K4[4][4];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      float v = (r == c) ? 1.f : 0.f;
      if (r < g.kdim && c < g.kdim) v = K[(static_cast<size_t>(b) * g.kdim + r) * g.kdim + c];
      if (r < 2) v = v / g.factor;
      K4[r][c] = v;
    }
  float T[4][4];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      if (Tb == nullptr) { T[r][c] = Ta[(static_cast<size_t>(b) * 4 + r) * 4 + c]; continue; }
      float acc = 0.f;
      for (int j = 0; j < 4; ++j) acc += Ta[(static_cast<size_t>(b) * 4 + r) * 4 + j] * Tb[(static_cast<size_t>(b) * 4 + j) * 4 + c];
      T[r][c] = acc;
    }
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float acc = 0.f;
      for (int j = 0; j < 4; ++j) acc += K4[r][j] * T[j][c];
      cam.P[r][c] = acc;
    }
  // adjugate inverse of the upper-left 3x3 (the [:3,:3] block of inverse(K4) when the last row is 0 0 0 1)
  const double a = K4[0][0], bb = K4[0][1], c = K4[0][2], d = K4[1][0], e = K4[1][1], f = K4[1][2];
  const double gg = K4[2][0], hh = K4[2][1], ii = K4[2][2];
  const double A = e * ii - f * hh, Bq = -(d * ii - f * gg), Cq = d * hh - e * gg;
  const double det = a * A + bb * Bq + c * Cq;
  const double inv = 1.0 / det;
  cam.iK[0][0] = static_cast<float>(A * inv);
  cam.iK[0][1] = static_cast<float>(-(bb * ii - c * hh) * inv);
  cam.iK[0][2] = static_cast<float>((bb * f - c * e) * inv);
  cam.iK[1][0] = static_cast<float>(Bq * inv);
  cam.iK[1][1] = static_cast<float>((a * ii - c * gg) * inv);
  cam.iK[1][2] = static_cast<float>(-(a * f - c * d) * inv);
  cam.iK[2][0] = static_cast<float>(Cq * inv);
  cam.iK[2][1] = static_cast<float>(-(a * hh - bb * gg) * inv);
  cam.iK[2][2] = static_cast<float>((a * e - bb * d) * inv);
  const float base = baseline_ptr ? baseline_ptr[b] : g.baseline;
  cam.bf = base * K4[0][0];
}

__global__ void __launch_bounds__(kThreads)
reproject_scatter(const float* __restrict__ pd, const float* __restrict__ parts, const float* __restrict__ mem_ds,
                  const float* __restrict__ mem_cv, const float* __restrict__ local_map, const float* __restrict__ K,
                  const float* __restrict__ Ta, const float* __restrict__ Tb, const float* __restrict__ baseline_ptr,
                  float* __restrict__ accum, Reproj g) {
  __shared__ float red[kThreads / 64];
  __shared__ Cam cam;
  const int b = blockIdx.y;
  const int hw = g.h * g.w, H = g.h, W = g.w;
  float s = 0.f;
  for (int j = threadIdx.x; j < g.nparts; j += kThreads) s += parts[j];
  if (threadIdx.x == 0) make_cam(cam, K, Ta, Tb, baseline_ptr, b, g);
  const float mean = block_sum(s, red) / static_cast<float>(static_cast<long long>(g.B) * hw);   // syncs: cam is visible
  const int CO = 2 * g.k + g.nl_out + 1;
  const float bf = cam.bf;
  float* ob = accum + static_cast<size_t>(b) * CO * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const int y = p / W, x = p - y * W;
    const float u = static_cast<float>(x), v = static_cast<float>(y);
    const float rx = cam.iK[0][0] * u + cam.iK[0][1] * v + cam.iK[0][2];
    const float ry = cam.iK[1][0] * u + cam.iK[1][1] * v + cam.iK[1][2];
    const float rz = cam.iK[2][0] * u + cam.iK[2][1] * v + cam.iK[2][2];
    const float d0 = pd[static_cast<size_t>(b) * hw + p];
    // depth after the rigid motion of a plane of disparity d (inverse_warp.py:132,148 then the z row)
    auto moved_z = [&](float d) {
      const float z = bf / (d + 1e-5f);
      return cam.P[2][0] * (rx * z) + cam.P[2][1] * (ry * z) + cam.P[2][2] * (rz * z) + cam.P[2][3];
    };
    const float z0 = bf / (d0 + 1e-5f);
    const float X = rx * z0, Y = ry * z0, Z = rz * z0;
    const float cx = cam.P[0][0] * X + cam.P[0][1] * Y + cam.P[0][2] * Z + cam.P[0][3];
    const float cy = cam.P[1][0] * X + cam.P[1][1] * Y + cam.P[1][2] * Z + cam.P[1][3];
    const float cz = cam.P[2][0] * X + cam.P[2][1] * Y + cam.P[2][2] * Z + cam.P[2][3];
    const float fx = cx / (cz + 1e-7f) - u, fy = cy / (cz + 1e-7f) - v;        // optical flow, :154,:170
    const float ox = u + fx, oy = v + fy;                                    // softsplat.py:19-20
    const float fx0 = floorf(ox), fy0 = floorf(oy);
    const int x0 = static_cast<int>(fminf(fmaxf(fx0, -2.f), 1.0e9f));
    const int y0 = static_cast<int>(fminf(fmaxf(fy0, -2.f), 1.0e9f));
    const bool xa = (x0 >= 0) & (x0 < W), xb = (x0 + 1 >= 0) & (x0 + 1 < W);
    const bool ya = (y0 >= 0) & (y0 < H), yb = (y0 + 1 >= 0) & (y0 + 1 < H);
    const bool inw = xa & ya, ine = xb & ya, isw = xa & yb, ise = xb & yb;
    if (!(inw | ine | isw | ise)) continue;
    const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
    const float nw = (x1 - ox) * (y1 - oy), ne = (ox - fx0) * (y1 - oy);
    const float sw = (x1 - ox) * (oy - fy0), se = (ox - fx0) * (oy - fy0);
    const float e = expf(fminf(fmaxf(d0 - mean, -50.f), 50.f));              // TemporalStereo.py:374-375
    float* o = ob + y0 * W + x0;
    auto put = [&](float val) {
      if (inw) unsafeAtomicAdd(o, val * nw);
      if (ine) unsafeAtomicAdd(o + 1, val * ne);
      if (isw) unsafeAtomicAdd(o + W, val * sw);
      if (ise) unsafeAtomicAdd(o + W + 1, val * se);
      o += hw;
    };
    for (int c = 0; c < g.k; ++c)
      put(bf / (moved_z(mem_ds[(static_cast<size_t>(b) * g.k + c) * hw + p]) + 1e-5f) * e);
    for (int c = 0; c < g.k; ++c) put(mem_cv[(static_cast<size_t>(b) * g.k + c) * hw + p] * e);
    for (int c = 0; c < g.nl_out; ++c) {
      const float d = (c == 0) ? d0 : local_map[(static_cast<size_t>(b) * g.nl_in + (c - 1)) * hw + p];
      put(bf / (moved_z(d) + 1e-5f) * e);
    }
    put(e);
  }
}

// out = accum / (accum[last] + 1e-22), routed to the three state tensors (softsplat.py:352-357)
__global__ void __launch_bounds__(kThreads)
reproject_normalize(const float* __restrict__ accum, float* __restrict__ out_ds, float* __restrict__ out_cv,
                    float* __restrict__ out_local, Reproj g) {
  const int hw = g.h * g.w;
  const int CO = 2 * g.k + g.nl_out + 1;
  const long long n = static_cast<long long>(g.B) * hw;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * kThreads) {
    const int b = static_cast<int>(i / hw);
    const int p = static_cast<int>(i - static_cast<long long>(b) * hw);
    const float* a = accum + static_cast<size_t>(b) * CO * hw + p;
    const float den = a[static_cast<size_t>(CO - 1) * hw] + 1e-22f;
    for (int c = 0; c < g.k; ++c) {
      out_ds[(static_cast<size_t>(b) * g.k + c) * hw + p] = a[static_cast<size_t>(c) * hw] / den;
      out_cv[(static_cast<size_t>(b) * g.k + c) * hw + p] = a[static_cast<size_t>(g.k + c) * hw] / den;
    }
    for (int c = 0; c < g.nl_out; ++c)
      out_local[(static_cast<size_t>(b) * g.nl_out + c) * hw + p] = a[static_cast<size_t>(2 * g.k + c) * hw] / den;
  }
}

int parts_for(long long n) {
  long long blocks = (n + kThreads - 1) / kThreads;
  if (blocks > kMaxParts) blocks = kMaxParts;
  return static_cast<int>(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" size_t ts_reproject_memory_workspace_bytes(int B, int h, int w, int k, int n_local_out) {
  if (B <= 0 || h <= 0 || w <= 0 || k < 0 || n_local_out < 0) return 0;
  const size_t hw = static_cast<size_t>(h) * w;
  return ts::round_up((static_cast<size_t>(B) * hw * (2 + 2 * k + n_local_out) + kMaxParts) * sizeof(float), 256);
}

extern "C" int ts_reproject_memory_fwd(const float* prev_disp, long long disp_bstride, int full_h, int full_w,
                                       const float* mem_disp, const float* mem_cost, int k,
                                       const float* local_map, int n_local_in, int n_local_out,
                                       const float* K, int k_dim, const float* T_a, const float* T_b,
                                       const float* baseline_ptr, float baseline, float factor,
                                       float* out_disp, float* out_cost, float* out_local, void* workspace,
                                       int B, int h, int w, void* stream) {
  TS_REQUIRE(B > 0 && h > 0 && w > 0 && full_h > 0 && full_w > 0, TS_ERR_SHAPE, "reproject_memory: non-positive size");
  TS_REQUIRE(B <= 65535, TS_ERR_UNSUPPORTED, "reproject_memory: batch too large");
  TS_REQUIRE(k >= 0 && n_local_in >= 0 && n_local_out >= 0 && k + n_local_out > 0, TS_ERR_SHAPE,
             "reproject_memory: nothing to re-project (k=%d, local=%d)", k, n_local_out);
  TS_REQUIRE(n_local_out <= n_local_in + 1, TS_ERR_SHAPE, "reproject_memory: %d local maps out of %d + previous disparity",
             n_local_out, n_local_in);
  TS_REQUIRE(k_dim == 3 || k_dim == 4, TS_ERR_SHAPE, "reproject_memory: K must be 3x3 or 4x4");
  TS_REQUIRE(factor > 0.f, TS_ERR_SHAPE, "reproject_memory: factor must be positive");
  TS_REQUIRE(static_cast<long long>(2 * k + n_local_out + 1) * h * w < (1ll << 31), TS_ERR_UNSUPPORTED,
             "reproject_memory: tensor too large");
  TS_REQUIRE_PTR(prev_disp); TS_REQUIRE_PTR(K); TS_REQUIRE_PTR(T_a); TS_REQUIRE_PTR(workspace);
  if (k > 0) { TS_REQUIRE_PTR(mem_disp); TS_REQUIRE_PTR(mem_cost); TS_REQUIRE_PTR(out_disp); TS_REQUIRE_PTR(out_cost); }
  if (n_local_out > 0) TS_REQUIRE_PTR(out_local);
  if (n_local_out > 1) TS_REQUIRE_PTR(local_map);
  hipStream_t st = ts::as_stream(stream);
  const long long n = static_cast<long long>(B) * h * w;
  Reproj g;
  g.B = B; g.Hf = full_h; g.Wf = full_w; g.h = h; g.w = w; g.k = k; g.nl_in = n_local_in; g.nl_out = n_local_out;
  g.kdim = k_dim; g.nparts = parts_for(n); g.disp_bstride = disp_bstride;
  g.sh = h > 1 ? static_cast<float>(full_h - 1) / static_cast<float>(h - 1) : 0.f;
  g.sw = w > 1 ? static_cast<float>(full_w - 1) / static_cast<float>(w - 1) : 0.f;
  g.factor = factor; g.baseline = baseline;
  float* pd = reinterpret_cast<float*>(workspace);
  float* parts = pd + n;
  float* accum = parts + kMaxParts;
  const long long accum_elems = n * (2 * k + n_local_out + 1);
  hipLaunchKernelGGL(reproject_prepare, dim3(g.nparts), dim3(kThreads), 0, st, prev_disp, pd, parts, accum, accum_elems, g);
  if (int rc = ts::launched("reproject_prepare")) return rc;
  const unsigned gx = static_cast<unsigned>((static_cast<long long>(h) * w + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(reproject_scatter, dim3(gx, B), dim3(kThreads), 0, st, pd, parts, mem_disp, mem_cost, local_map, K,
                     T_a, T_b, baseline_ptr, accum, g);
  if (int rc = ts::launched("reproject_scatter")) return rc;
  hipLaunchKernelGGL(reproject_normalize, dim3(g.nparts), dim3(kThreads), 0, st, accum, out_disp, out_cost, out_local, g);
  return ts::launched("reproject_normalize");
}
