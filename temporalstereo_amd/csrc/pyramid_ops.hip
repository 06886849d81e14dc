// K3/K5 -- the non-convolution stages of the aggregation pyramid for gfx950 (MI355X), inference form.
//
//   resize_add_act   F.interpolate(trilinear, align_corners) to the skip's size + add + SiLU
//                    (ResidualBlock3D.forward, module.py:285-295)
//   pool5_avgmax     avg_pool3d / max_pool3d 5^3, stride 1, padding 2 (PyramidFusion, module.py:415-417)
//   merge_candidates past_conv + cat + sort + gather along D (coarse.py:84-105, fine.py:105-122)
//   convex_upsample  softmax-over-9 weighted x2 upsampling (ConvexUpsample.forward, module.py:336-353)
//   unet_upsample    softmax-over-9 weighted x4 upsampling (UNet.upsample, module.py:468-482)
//   resize_bilinear  F.interpolate(bilinear, align_corners) with a value scale (memory resizes,
//                    coarse.py:91-96, precise.py:100-103, projects/TemporalStereo/TemporalStereo.py:305-309)
//
// All are bandwidth/latency-bound element kernels: W is the contiguous axis, consecutive lanes own
// consecutive pixels, every tensor is read once and written once (the 5^3 pooling goes through an
// LDS plane tile and a 5-deep register ring along D instead of 125 global taps per output).
#include "ts_common.hpp"

namespace {

// (the inference forms; v_exp_f32 / v_rcp_f32 as conv_common.hpp's silu_fast)
__device__ __forceinline__ float silu(float v) {
#ifdef TS_EXACT_SILU
  return v / (1.f + expf(-v));
#endif
  const float d = 1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  const float r = __builtin_amdgcn_rcpf(d);
  const float rn = fmaf(fmaf(-d, r, 1.f), r, r);
  return v * (d < 3.0e38f ? rn : r);          // d = inf: the Newton step would be inf * 0
}

unsigned grid_for(long long n, int threads) {
  long long blocks = (n + threads - 1) / threads;
  const long long cap = static_cast<long long>(ts::kNumCU) * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return static_cast<unsigned>(blocks);
}

// one 256-thread workgroup per 256 elements, uncapped (kernels without a grid-stride loop)
unsigned exact_grid(long long n) { return static_cast<unsigned>((n + 255) / 256); }

// source index / weight of torch's align_corners=True linear interpolation
__device__ __forceinline__ void lin_src(float scale, int dst, int in_size, int& i0, int& i1, float& l1) {
  const float s = scale * static_cast<float>(dst);
  i0 = static_cast<int>(s);
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - static_cast<float>(i0);
}

inline float ac_scale(int in_size, int out_size) {
  return out_size > 1 ? static_cast<float>(in_size - 1) / static_cast<float>(out_size - 1) : 0.f;
}

// ------------------------------------------------------------------------------- resize + add + act
struct Resize3 {
  int C, Da, Ha, Wa, D, H, W;
  float sd, sh, sw;
  int act;   // 0 none, 1 SiLU
  long long a_bstride, a_cstride, b_bstride, b_cstride, o_bstride, o_cstride;
};

__global__ void __launch_bounds__(256)
resize_add_act_kernel(const float* __restrict__ a, const float* __restrict__ bsrc, float* __restrict__ out, const Resize3 p) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int n = p.D * p.H * p.W;
  const float* ap = a + static_cast<size_t>(b) * p.a_bstride + static_cast<size_t>(c) * p.a_cstride;
  const float* bp = bsrc ? bsrc + static_cast<size_t>(b) * p.b_bstride + static_cast<size_t>(c) * p.b_cstride : nullptr;
  float* op = out + static_cast<size_t>(b) * p.o_bstride + static_cast<size_t>(c) * p.o_cstride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = i % p.W;
    const int t = i / p.W;
    const int y = t % p.H, d = t / p.H;
    int d0, d1, y0, y1, x0, x1;
    float ld, ly, lx;
    lin_src(p.sd, d, p.Da, d0, d1, ld);
    lin_src(p.sh, y, p.Ha, y0, y1, ly);
    lin_src(p.sw, x, p.Wa, x0, x1, lx);
    const size_t HWa = static_cast<size_t>(p.Ha) * p.Wa;
    const float* p0 = ap + d0 * HWa;
    const float* p1 = ap + d1 * HWa;
    const float v00 = (1.f - lx) * p0[y0 * p.Wa + x0] + lx * p0[y0 * p.Wa + x1];
    const float v01 = (1.f - lx) * p0[y1 * p.Wa + x0] + lx * p0[y1 * p.Wa + x1];
    const float v10 = (1.f - lx) * p1[y0 * p.Wa + x0] + lx * p1[y0 * p.Wa + x1];
    const float v11 = (1.f - lx) * p1[y1 * p.Wa + x0] + lx * p1[y1 * p.Wa + x1];
    float v = (1.f - ld) * ((1.f - ly) * v00 + ly * v01) + ld * ((1.f - ly) * v10 + ly * v11);
    if (bp) v += bp[i];
    op[i] = p.act == 1 ? silu(v) : v;
  }
}

// ------------------------------------------------------------------------------- 5^3 avg + max pool
constexpr int PT_Y = 8, PT_X = 32;
struct Pool5 {
  int C, D, H, W;
  long long x_bstride, x_cstride, avg_bstride, avg_cstride, max_bstride, max_cstride;
};

__global__ void __launch_bounds__(256)
pool5_avgmax_kernel(const float* __restrict__ x, float* __restrict__ oavg, float* __restrict__ omax, const Pool5 p) {
  // the haloed plane tile TWICE: padding as 0 for the average (count_include_pad) and as -inf for the maximum, so that a tap is one
  // add and one max.  (Rounds 2-5 kept ONE tile with padding marked NaN and tested every tap: 5 VALU instructions per tap, 125 per
  // plane and lane -- the kernel was bound by exactly that: 15.5 us for the coarse level's 14 planes, tools/layer_table.py.)
  __shared__ float tile_s[PT_Y + 4][PT_X + 4 + 1];
  __shared__ float tile_m[PT_Y + 4][PT_X + 4 + 1];
  const int tiles_x = (p.W + PT_X - 1) / PT_X;
  const int ty0 = (blockIdx.x / tiles_x) * PT_Y, tx0 = (blockIdx.x % tiles_x) * PT_X;
  const int c = blockIdx.y, b = blockIdx.z;
  const int tx = threadIdx.x & (PT_X - 1), ty = threadIdx.x / PT_X;
  const int y = ty0 + ty, xx = tx0 + tx;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* xp = x + static_cast<size_t>(b) * p.x_bstride + static_cast<size_t>(c) * p.x_cstride;
  float* ap = oavg + static_cast<size_t>(b) * p.avg_bstride + static_cast<size_t>(c) * p.avg_cstride;
  float* mp = omax + static_cast<size_t>(b) * p.max_bstride + static_cast<size_t>(c) * p.max_cstride;
  // this thread's (up to two) elements of the haloed plane tile: clamped source offset + padding flag
  constexpr int TW = PT_X + 4, TN = (PT_Y + 4) * TW;
  int soff[2], lidx[2];
  bool pad[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = min(static_cast<int>(threadIdx.x) + 256 * e, TN - 1);       // the surplus threads of e == 1 repeat the last element
    const int cx = i % TW, cy = i / TW;
    const int gy = ty0 + cy - 2, gx = tx0 + cx - 2;
    pad[e] = !(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W);
    soff[e] = min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1);
    lidx[e] = cy * (TW + 1) + cx;
  }
  float* tls = &tile_s[0][0];
  float* tlm = &tile_m[0][0];
  float nxt[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) nxt[e] = xp[soff[e]];                            // plane 0
  float rs[5], rm[5];       // 2-D pooled planes d-4 .. d (ring kept in order by shifting)
#pragma unroll
  for (int i = 0; i < 5; ++i) { rs[i] = 0.f; rm[i] = -INFINITY; }
  for (int d = 0; d < p.D + 2; ++d) {
    float s2 = 0.f, m2 = -INFINITY;
    if (d < p.D) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        tls[lidx[e]] = pad[e] ? 0.f : nxt[e];
        tlm[lidx[e]] = pad[e] ? -INFINITY : nxt[e];
      }
      __syncthreads();
      const int dn = min(d + 1, p.D - 1);                                      // next plane in flight under the 25 taps
#pragma unroll
      for (int e = 0; e < 2; ++e) nxt[e] = xp[dn * HW + soff[e]];
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          s2 += tile_s[ty + ky][tx + kx];
          m2 = fmaxf(m2, tile_m[ty + ky][tx + kx]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { rs[i] = rs[i + 1]; rm[i] = rm[i + 1]; }
    rs[4] = s2; rm[4] = m2;                     // planes beyond D contribute 0 / -inf
    const int od = d - 2;
    if (od >= 0 && y < p.H && xx < p.W) {
      const float s = rs[0] + rs[1] + rs[2] + rs[3] + rs[4];
      const float m = fmaxf(fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3])), rm[4]);
      const size_t o = od * HW + static_cast<size_t>(y) * p.W + xx;
      if (oavg) ap[o] = s * (1.f / 125.f);
      if (omax) mp[o] = m;
    }
  }
}

// ------------------------------------------------------------------------------- merge candidates
constexpr int MERGE_DMAX = 20;
struct Merge {
  int B, C, D0, K, HW;            // D0 computed candidates + K memory candidates
  int implicit_samples;            // 1: candidate i has disparity i (coarse level)
  long long vol_bstride, vol_cstride, out_bstride, out_cstride;
};

// One lane per pixel: rank every candidate (stable: ties keep their original order, SURVEY.md
// Appendix B.2), write the sorted disparities, and move each candidate's C-channel column of the
// volume to its sorted slot.  Memory candidates get their volume from past_conv:
// Conv3d(1->C, 1x1x1, no bias) + BatchNorm + SiLU of the remembered cost (coarse.py:42,98).
constexpr int MERGE_CPB = 2;    // channels moved per workgroup row (grid.y walks channel groups); 4 / 2 / 1: 11.5 / 9.3 / 8.5 us at the coarse level, 22.3 / 18.4 / 18.8 at batch 4

// DM: compile-time bound on the candidate count.  Every load is unconditional (indices clamped, results
// selected afterwards) and issued before anything waits, so a lane pays ONE memory round trip for its
// samples and one for its volume columns instead of one per candidate.
template <int DM>
__global__ void __launch_bounds__(256)
merge_candidates_kernel(const float* __restrict__ vol, const float* __restrict__ samp, const float* __restrict__ mem_samp,
                        const float* __restrict__ mem_cost, const float* __restrict__ pw, const float* __restrict__ pscale,
                        const float* __restrict__ pshift, float* __restrict__ out_samp, float* __restrict__ out_vol,
                        const Merge p) {
  const int DT = p.D0 + p.K;
  const int c0 = blockIdx.y * MERGE_CPB;
  const long long n = static_cast<long long>(p.B) * p.HW;
  const bool has_mem = mem_samp != nullptr && mem_cost != nullptr && p.K > 0;
  const float* ms = has_mem ? mem_samp : vol;          // any readable address; the value is discarded
  const float* mc = has_mem ? mem_cost : vol;
  const int Kc = max(p.K, 1);
  float w_[MERGE_CPB], sc_[MERGE_CPB], sh_[MERGE_CPB];
#pragma unroll
  for (int cc = 0; cc < MERGE_CPB; ++cc) {
    const int c = min(c0 + cc, p.C - 1);
    w_[cc] = p.K > 0 ? pw[c] : 0.f; sc_[cc] = p.K > 0 ? pscale[c] : 0.f; sh_[cc] = p.K > 0 ? pshift[c] : 0.f;
  }
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / p.HW);
    const int px = static_cast<int>(i - static_cast<long long>(b) * p.HW);
    float s[DM], m[DM], v[DM][MERGE_CPB];
    const float* vb = vol + static_cast<size_t>(b) * p.vol_bstride + px;
#pragma unroll
    for (int j = 0; j < DM; ++j) {
      const int jv = min(j, p.D0 - 1), jm = min(max(j - p.D0, 0), Kc - 1);
      const size_t mo = has_mem ? (static_cast<size_t>(b) * p.K + jm) * p.HW + px : 0;
      s[j] = p.implicit_samples ? static_cast<float>(j) : samp[(static_cast<size_t>(b) * p.D0 + jv) * p.HW + px];
      const float a = ms[mo];
      m[j] = mc[mo];
      if (j >= p.D0) s[j] = (j < DT) ? (has_mem ? a : 0.f) : INFINITY;
      if (!has_mem) m[j] = 0.f;
#pragma unroll
      for (int cc = 0; cc < MERGE_CPB; ++cc)
        v[j][cc] = vb[static_cast<size_t>(jv) * p.HW + static_cast<size_t>(min(c0 + cc, p.C - 1)) * p.vol_cstride];
    }
#pragma unroll
    for (int j = 0; j < DM; ++j) {
      int rank = 0;
#pragma unroll
      for (int q = 0; q < DM; ++q) rank += (s[q] < s[j]) || (s[q] == s[j] && q < j);     // padding (inf) ranks last
      if (j < DT) {
        if (c0 == 0) out_samp[(static_cast<size_t>(b) * DT + rank) * p.HW + px] = s[j];
        float* ov = out_vol + static_cast<size_t>(b) * p.out_bstride + static_cast<size_t>(rank) * p.HW + px;
#pragma unroll
        for (int cc = 0; cc < MERGE_CPB; ++cc) {
          const float val = (j < p.D0) ? v[j][cc] : silu(w_[cc] * m[j] * sc_[cc] + sh_[cc]);
          if (c0 + cc < p.C) ov[static_cast<size_t>(c0 + cc) * p.out_cstride] = val;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------- 9-tap upsamplers
// ConvexUpsample x r (r = 2): mask [B, 9*r*r, H, W] viewed (9, r, r); out [B,1,rH,rW]
__global__ void __launch_bounds__(256)
convex_upsample_kernel(const float* __restrict__ mask, const float* __restrict__ disp, float* __restrict__ out,
                       int B, int H, int W, int r, float disp_scale,
                       float* __restrict__ low, float* __restrict__ high, float* __restrict__ cand, float range, int coff, int ctot) {
  const int HW = H * W, Ho = H * r, Wo = W * r;
  const long long n = static_cast<long long>(B) * Ho * Wo;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int ox = static_cast<int>(i % Wo);
    const long long t = i / Wo;
    const int oy = static_cast<int>(t % Ho), b = static_cast<int>(t / Ho);
    const int y = oy / r, ry = oy - y * r, x = ox / r, rx = ox - x * r;
    const float* mp = mask + (static_cast<size_t>(b) * 9 * r * r + ry * r + rx) * HW + static_cast<size_t>(y) * W + x;
    const float* dp = disp + static_cast<size_t>(b) * HW;
    float m[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = mp[static_cast<size_t>(k) * r * r * HW]; mx = fmaxf(mx, m[k]); }
    float den = 0.f, acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float e = expf(m[k] - mx);
      den += e;
      const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
      const float dv = dp[min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];        // unconditional; zero padding by select
      const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? dv * disp_scale : 0.f;
      acc += e * v;
    }
    const float dup = acc / den;
    out[i] = dup;
    if (low) {      // the next level's search range and candidates in the same pass (see range_candidates_kernel)
      const float lo = dup - range, hi = dup + range;
      low[i] = lo; high[i] = hi;
      const float span = fabsf(hi - lo), base = fminf(lo, hi);
      const size_t HWo = static_cast<size_t>(Ho) * Wo;
      float* c = cand + (static_cast<size_t>(b) * ctot + coff) * HWo + static_cast<size_t>(oy) * Wo + ox;
      c[0] = span * 0.f + base;
      c[HWo] = span * 0.375f + base;
      c[2 * HWo] = span * 0.5f + base;
      c[3 * HWo] = span * 0.625f + base;
      c[4 * HWo] = span * 1.f + base;
    }
  }
}

// UNet.upsample: mask [B,9,Ho,Wo] softmax over 9; out = sum_k bilinear(unfold(disp)_k * Wo/w)(oy,ox) * p_k
// V consecutive output pixels of a row per thread: the 9 logit planes (the bulk of the traffic: 18.8 MB at
// 544x960) are read as 16-byte vectors when V == 4.
template <int V>
__global__ void __launch_bounds__(256)
unet_upsample_kernel(const float* __restrict__ mask, const float* __restrict__ disp, float* __restrict__ out,
                     int B, int h, int w, int Ho, int Wo, float sh, float sw) {
  const int Wv = Wo / V;
  const long long n = static_cast<long long>(B) * Ho * Wv;
  const size_t HWo = static_cast<size_t>(Ho) * Wo;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int oxv = static_cast<int>(i % Wv) * V;
    const long long t = i / Wv;
    const int oy = static_cast<int>(t % Ho), b = static_cast<int>(t / Ho);
    const float* mp = mask + static_cast<size_t>(b) * 9 * HWo + static_cast<size_t>(oy) * Wo + oxv;
    const float* dp = disp + static_cast<size_t>(b) * h * w;
    float m[9][V];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if constexpr (V == 4) {
        const float4 q = *reinterpret_cast<const float4*>(mp + k * HWo);
        m[k][0] = q.x; m[k][1] = q.y; m[k][2] = q.z; m[k][3] = q.w;
      } else {
        m[k][0] = mp[k * HWo];
      }
    }
    int y0, y1;
    float ly;
    lin_src(sh, oy, h, y0, y1, ly);
    const bool iy = y1 > y0;
    // The 36 (tap, bilinear corner) reads of the reference for one output pixel all land in the 4x4 patch around
    // (y0, x0); the V pixels of a thread span at most V/ceil(1/sw)+1 source columns, so ONE 4 x (4 + XS) patch serves
    // them all (XS = 1 when V == 4 and the upsampling factor is >= 4: 20 loads instead of 64).
    constexpr int XS = (V == 4) ? 1 : 0;
    int xb, xb1;
    float lxb;
    lin_src(sw, oxv, w, xb, xb1, lxb);
    int xe = xb, xe1;
    float lxe;
    if (V > 1) lin_src(sw, oxv + V - 1, w, xe, xe1, lxe);
    const bool shared = (xe - xb) <= XS;
    float P[4][4 + XS];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4 + XS; ++c) {
        const int yy = y0 - 1 + r, xx = xb - 1 + c;
        const float dv = dp[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
        // disp * w_out / w_in in the reference's evaluation order (module.py:478); zero padding of unfold
        P[r][c] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dv * static_cast<float>(Wo) / static_cast<float>(w) : 0.f;
      }
    float res[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int ox = oxv + v;
      int x0, x1;
      float lx;
      lin_src(sw, ox, w, x0, x1, lx);
      float pd[4][4];
      if (shared) {
        const bool sh = XS && (x0 != xb);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) pd[r][c] = sh ? P[r][c + XS] : P[r][c];
      } else {                                  // small upsampling factors: each pixel reads its own patch
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int yy = y0 - 1 + r, xx = x0 - 1 + c;
            const float dv = dp[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
            pd[r][c] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dv * static_cast<float>(Wo) / static_cast<float>(w) : 0.f;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 9; ++k) mx = fmaxf(mx, m[k][v]);
      const bool ix = x1 > x0;
      float den = 0.f, acc = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const float e = expf(m[k][v] - mx);
        den += e;
        const int ry = k / 3, rx = k % 3;            // patch row / column of (y0 + dy, x0 + dx)
        const float a00 = pd[ry][rx];
        const float a01 = ix ? pd[ry][rx + 1] : a00;
        const float a10 = iy ? pd[ry + 1][rx] : a00;
        const float a11 = iy ? (ix ? pd[ry + 1][rx + 1] : pd[ry + 1][rx]) : a01;
        const float top = (1.f - lx) * a00 + lx * a01;
        const float bot = (1.f - lx) * a10 + lx * a11;
        acc += e * ((1.f - ly) * top + ly * bot);
      }
      res[v] = acc / den;
    }
    float* op = out + static_cast<size_t>(b) * HWo + static_cast<size_t>(oy) * Wo + oxv;
    if constexpr (V == 4) *reinterpret_cast<float4*>(op) = make_float4(res[0], res[1], res[2], res[3]);
    else op[0] = res[0];
  }
}

// ------------------------------------------------------------------------------- bilinear resize
__global__ void __launch_bounds__(256)
resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, int BC, int C, int h, int w, int Ho, int Wo,
                       float sh, float sw, float vscale, long long out_bstride) {
  const long long n = static_cast<long long>(BC) * Ho * Wo;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int ox = static_cast<int>(i % Wo);
    const long long t = i / Wo;
    const int oy = static_cast<int>(t % Ho), bc = static_cast<int>(t / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    lin_src(sh, oy, h, y0, y1, ly);
    lin_src(sw, ox, w, x0, x1, lx);
    const float* p = x + static_cast<size_t>(bc) * h * w;
    const float top = (1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1];
    const float bot = (1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1];
    const int b = bc / C, c = bc - b * C;
    out[static_cast<size_t>(b) * out_bstride + (static_cast<size_t>(c) * Ho + oy) * Wo + ox] = ((1.f - ly) * top + ly * bot) * vscale;
  }
}

// two same-shaped maps in one launch (the top-k memory's candidates and costs, precise.py:100-103 / coarse.py:91-96)
__global__ void __launch_bounds__(256)
resize_bilinear_pair_kernel(const float* __restrict__ x0, const float* __restrict__ x1, float* __restrict__ out0,
                            float* __restrict__ out1, int BC, int h, int w, int Ho, int Wo, float sh, float sw, float vs0,
                            float vs1) {
  const float* x = blockIdx.y ? x1 : x0;
  float* out = blockIdx.y ? out1 : out0;
  const float vscale = blockIdx.y ? vs1 : vs0;
  const long long n = static_cast<long long>(BC) * Ho * Wo;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int ox = static_cast<int>(i % Wo);
    const long long t = i / Wo;
    const int oy = static_cast<int>(t % Ho), bc = static_cast<int>(t / Ho);
    int y0, y1, xa, xb;
    float ly, lx;
    lin_src(sh, oy, h, y0, y1, ly);
    lin_src(sw, ox, w, xa, xb, lx);
    const float* p = x + static_cast<size_t>(bc) * h * w;
    const float top = (1.f - lx) * p[y0 * w + xa] + lx * p[y0 * w + xb];
    const float bot = (1.f - lx) * p[y1 * w + xa] + lx * p[y1 * w + xb];
    out[i] = ((1.f - ly) * top + ly * bot) * vscale;
  }
}

// search range and its five candidates from an upsampled disparity (TemporalStereo.py:110,119 and
// fine.py:82-87 / precise.py:73-78): low = d - r, high = d + r, cand_i = |high-low| * {0,3,4,5,8}/8 + min(low,high)
__global__ void __launch_bounds__(256)
range_candidates_kernel(const float* __restrict__ disp, float* __restrict__ low, float* __restrict__ high,
                        float* __restrict__ cand, int B, int HW, float range, int coff, int ctot) {
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int px = static_cast<int>(i - static_cast<long long>(b) * HW);
    const float d = disp[i];
    const float lo = d - range, hi = d + range;
    low[i] = lo; high[i] = hi;
    const float span = fabsf(hi - lo), base = fminf(lo, hi);
    float* c = cand + (static_cast<size_t>(b) * ctot + coff) * HW + px;
    c[0] = span * 0.f + base;
    c[static_cast<size_t>(1) * HW] = span * 0.375f + base;
    c[static_cast<size_t>(2) * HW] = span * 0.5f + base;
    c[static_cast<size_t>(3) * HW] = span * 0.625f + base;
    c[static_cast<size_t>(4) * HW] = span * 1.f + base;
  }
}

// ------------------------------------------------------------------------------- backward (training)
// resize_add_act: out = act(trilinear(a) + add).  One lane per output element: recompute the
// pre-activation, g = dOut * act'(pre); dAdd = g; dA += the 8 interpolation weights * g (fp32 atomics,
// dA zero-filled by the entry -- the same scatter torch's upsample_trilinear3d_backward performs).
struct Resize3B {
  Resize3 f;
  long long g_bstride, g_cstride, ga_bstride, ga_cstride, gb_bstride, gb_cstride;
};

__global__ void __launch_bounds__(256)
resize_add_act_bwd_kernel(const float* __restrict__ a, const float* __restrict__ bsrc, const float* __restrict__ gout,
                          float* __restrict__ ga, float* __restrict__ gadd, const Resize3B q) {
  const Resize3& p = q.f;
  const int c = blockIdx.y, b = blockIdx.z;
  const int n = p.D * p.H * p.W;
  const float* ap = a + static_cast<size_t>(b) * p.a_bstride + static_cast<size_t>(c) * p.a_cstride;
  const float* bp = bsrc ? bsrc + static_cast<size_t>(b) * p.b_bstride + static_cast<size_t>(c) * p.b_cstride : nullptr;
  const float* gp = gout + static_cast<size_t>(b) * q.g_bstride + static_cast<size_t>(c) * q.g_cstride;
  float* gap = ga + static_cast<size_t>(b) * q.ga_bstride + static_cast<size_t>(c) * q.ga_cstride;
  float* gbp = gadd ? gadd + static_cast<size_t>(b) * q.gb_bstride + static_cast<size_t>(c) * q.gb_cstride : nullptr;
  const size_t HWa = static_cast<size_t>(p.Ha) * p.Wa;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = i % p.W;
    const int t = i / p.W;
    const int y = t % p.H, d = t / p.H;
    int d0, d1, y0, y1, x0, x1;
    float ld, ly, lx;
    lin_src(p.sd, d, p.Da, d0, d1, ld);
    lin_src(p.sh, y, p.Ha, y0, y1, ly);
    lin_src(p.sw, x, p.Wa, x0, x1, lx);
    float g = gp[i];
    if (p.act == 1) {
      const float* p0 = ap + d0 * HWa;
      const float* p1 = ap + d1 * HWa;
      const float v00 = (1.f - lx) * p0[y0 * p.Wa + x0] + lx * p0[y0 * p.Wa + x1];
      const float v01 = (1.f - lx) * p0[y1 * p.Wa + x0] + lx * p0[y1 * p.Wa + x1];
      const float v10 = (1.f - lx) * p1[y0 * p.Wa + x0] + lx * p1[y0 * p.Wa + x1];
      const float v11 = (1.f - lx) * p1[y1 * p.Wa + x0] + lx * p1[y1 * p.Wa + x1];
      float v = (1.f - ld) * ((1.f - ly) * v00 + ly * v01) + ld * ((1.f - ly) * v10 + ly * v11);
      if (bp) v += bp[i];
      const float sg = 1.f / (1.f + expf(-v));
      g *= sg * (1.f + v * (1.f - sg));                    // d/dv [v * sigmoid(v)]
    }
    if (gbp) gbp[i] = g;
    const float wd[2] = {1.f - ld, ld}, wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
    const int dd[2] = {d0, d1}, yy[2] = {y0, y1}, xx[2] = {x0, x1};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kd = k >> 2, ky = (k >> 1) & 1, kx = k & 1;
      const float w = wd[kd] * wy[ky] * wx[kx];
      if (w != 0.f) atomicAdd(gap + dd[kd] * HWa + static_cast<size_t>(yy[ky]) * p.Wa + xx[kx], g * w);
    }
  }
}

// max half of the 5^3 pooling backward: every output re-finds its arg-max (first occurrence in (d,y,x)
// order, strict '>', as the framework's max_pool3d_with_indices) through the same plane tile + 5-deep
// ring as the forward and adds its gradient there.  The avg half is the forward's own box filter applied
// to dAvg (a symmetric kernel is its own adjoint), written first by the entry.
__global__ void __launch_bounds__(256)
pool5_max_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gmax, float* __restrict__ gx, const Pool5 p) {
  __shared__ float tile[PT_Y + 4][PT_X + 4 + 1];
  const int tiles_x = (p.W + PT_X - 1) / PT_X;
  const int ty0 = (blockIdx.x / tiles_x) * PT_Y, tx0 = (blockIdx.x % tiles_x) * PT_X;
  const int c = blockIdx.y, b = blockIdx.z;
  const int tx = threadIdx.x & (PT_X - 1), ty = threadIdx.x / PT_X;
  const int y = ty0 + ty, xx = tx0 + tx;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* xp = x + static_cast<size_t>(b) * p.x_bstride + static_cast<size_t>(c) * p.x_cstride;
  const float* gp = gmax + static_cast<size_t>(b) * p.max_bstride + static_cast<size_t>(c) * p.max_cstride;
  float* op = gx + static_cast<size_t>(b) * p.avg_bstride + static_cast<size_t>(c) * p.avg_cstride;
  // the haloed plane tile as in pool5_avgmax_kernel: this thread's (up to two) elements, their clamped source offsets and padding
  // flags computed once, the NEXT plane's values in flight under the 25 taps (round 5: the loads of a plane used to be issued
  // after the barrier that waits for them, one exposed round trip per depth plane: 66-73 us per call against the forward's 14)
  constexpr int TW = PT_X + 4, TN = (PT_Y + 4) * TW;
  int soff[2], lidx[2];
  bool pad[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = min(static_cast<int>(threadIdx.x) + 256 * e, TN - 1);
    const int cx = i % TW, cy = i / TW;
    const int gy = ty0 + cy - 2, gx = tx0 + cx - 2;
    pad[e] = !(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W);
    soff[e] = min(max(gy, 0), p.H - 1) * p.W + min(max(gx, 0), p.W - 1);
    lidx[e] = cy * (TW + 1) + cx;
  }
  float* tl = &tile[0][0];
  float nxt[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) nxt[e] = xp[soff[e]];
  // Neighbouring windows share their maximum (a local maximum wins up to 125 of them): one global atomic per OUTPUT piled them onto
  // a few addresses.  The gradient is gathered in LDS first -- one double accumulator per element of the haloed tile and plane of the
  // five-plane ring, ds_add_f64 (tools/exp/lds_atomic_rate.hip) -- and a plane leaves with one global atomic per touched element once
  // no later output plane can reach it.
  __shared__ double acc[5][TN];
  for (int i = threadIdx.x; i < 5 * TN; i += 256) (&acc[0][0])[i] = 0.0;
  float rm[5];
  int ri[5];                  // winner of the plane as a position in the haloed tile (row * TW + column), -1 = none
#pragma unroll
  for (int i = 0; i < 5; ++i) { rm[i] = -INFINITY; ri[i] = -1; }
  auto flush = [&](int plane) {                                  // plane >= 0: its accumulators are final
    double* a = acc[plane % 5];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = static_cast<int>(threadIdx.x) + 256 * e;
      if (i < TN) {
        const double v = a[i];
        if (v != 0.0) {                                          // (padding never wins: its accumulators stay zero)
          atomicAdd(op + static_cast<size_t>(plane) * HW + soff[e], static_cast<float>(v));
          a[i] = 0.0;
        }
      }
    }
  };
  for (int d = 0; d < p.D + 2; ++d) {
    float m2 = -INFINITY;
    int i2 = -1;
    if (d < p.D) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 2; ++e) tl[lidx[e]] = pad[e] ? NAN : nxt[e];
      __syncthreads();
      const int dn = min(d + 1, p.D - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e) nxt[e] = xp[dn * HW + soff[e]];
      int best = -1;                                               // tap number ky * 5 + kx of the plane's maximum
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float v = tile[ty + ky][tx + kx];
          if (v > m2) { m2 = v; best = ky * 5 + kx; }            // NaN (padding) never wins; the first of equals stays
        }
      if (best >= 0) i2 = (ty + best / 5) * TW + (tx + best % 5);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { rm[i] = rm[i + 1]; ri[i] = ri[i + 1]; }
    rm[4] = m2; ri[4] = i2;
    const int od = d - 2;
    if (od >= 0 && y < p.H && xx < p.W) {
      float m = rm[0];
      int mi = ri[0], mp = 0;                                     // ring slot i holds plane od - 2 + i
#pragma unroll
      for (int i = 1; i < 5; ++i)
        if (rm[i] > m) { m = rm[i]; mi = ri[i]; mp = i; }
      if (mi >= 0)
        __hip_atomic_fetch_add(&acc[(od - 2 + mp + 5) % 5][mi], static_cast<double>(gp[od * HW + static_cast<size_t>(y) * p.W + xx]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (od >= 2) {                // plane od - 2 is out of reach of every later output plane
      __syncthreads();
      flush(od - 2);
    }
  }
  __syncthreads();
  for (int plane = max(p.D - 2, 0); plane < p.D; ++plane) flush(plane);
}

// sort + gather backward (coarse.py:103-105 / fine.py:120-122 under autograd): rank of candidate j among its
// pixel's DT candidates (stable), dVol[:, j] = dOutVol[:, rank_j], dSample[j] = dOutSample[rank_j].
template <int DM>
__global__ void __launch_bounds__(256)
merge_bwd_kernel(const float* __restrict__ samp, const float* __restrict__ g_out_vol, const float* __restrict__ g_out_samp,
                 float* __restrict__ g_vol, float* __restrict__ g_samp, int B, int C, int DT, int HW) {
  const int c0 = blockIdx.y * MERGE_CPB;
  const long long n = static_cast<long long>(B) * HW;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i / HW);
    const int px = static_cast<int>(i - static_cast<long long>(b) * HW);
    float s[DM];
#pragma unroll
    for (int j = 0; j < DM; ++j) s[j] = (j < DT) ? samp[(static_cast<size_t>(b) * DT + min(j, DT - 1)) * HW + px] : INFINITY;
#pragma unroll
    for (int j = 0; j < DM; ++j) {
      if (j >= DT) continue;
      int rank = 0;
#pragma unroll
      for (int q = 0; q < DM; ++q) rank += (s[q] < s[j]) || (s[q] == s[j] && q < j);
      if (c0 == 0 && g_samp) g_samp[(static_cast<size_t>(b) * DT + j) * HW + px] = g_out_samp ? g_out_samp[(static_cast<size_t>(b) * DT + rank) * HW + px] : 0.f;
#pragma unroll
      for (int cc = 0; cc < MERGE_CPB; ++cc) {
        const int c = c0 + cc;
        if (c < C) g_vol[((static_cast<size_t>(b) * C + c) * DT + j) * HW + px] = g_out_vol[((static_cast<size_t>(b) * C + c) * DT + rank) * HW + px];
      }
    }
  }
}

}  // namespace

extern "C" int ts_range_candidates_fwd(const float* disp, float* low, float* high, float* candidates, int B, int H, int W,
                                       float range, int channel_offset, int channels_total, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && channel_offset >= 0 && channel_offset + 5 <= channels_total, TS_ERR_SHAPE,
             "range_candidates: bad size");
  TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(low); TS_REQUIRE_PTR(high); TS_REQUIRE_PTR(candidates);
  hipLaunchKernelGGL(range_candidates_kernel, dim3(grid_for(static_cast<long long>(B) * H * W, 256)), dim3(256), 0,
                     ts::as_stream(stream), disp, low, high, candidates, B, H * W, range, channel_offset, channels_total);
  return ts::launched("range_candidates_kernel");
}

extern "C" int ts_resize3d_add_act_fwd(const float* a, const float* add, float* out, int B, int C, int Da, int Ha, int Wa,
                                       int D, int H, int W, int act, long long a_bstride, long long a_cstride,
                                       long long add_bstride, long long add_cstride, long long out_bstride,
                                       long long out_cstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && Da > 0 && Ha > 0 && Wa > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "resize3d: non-positive size");
  TS_REQUIRE(B <= 65535 && C <= 65535, TS_ERR_UNSUPPORTED, "resize3d: grid too large");
  TS_REQUIRE_PTR(a); TS_REQUIRE_PTR(out);
  Resize3 p;
  p.C = C; p.Da = Da; p.Ha = Ha; p.Wa = Wa; p.D = D; p.H = H; p.W = W;
  p.sd = ac_scale(Da, D); p.sh = ac_scale(Ha, H); p.sw = ac_scale(Wa, W);
  p.act = act;
  p.a_bstride = a_bstride; p.a_cstride = a_cstride; p.b_bstride = add_bstride; p.b_cstride = add_cstride;
  p.o_bstride = out_bstride; p.o_cstride = out_cstride;
  const int n = D * H * W;
  int blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(resize_add_act_kernel, dim3(blocks, C, B), dim3(256), 0, ts::as_stream(stream), a, add, out, p);
  return ts::launched("resize_add_act_kernel");
}

extern "C" int ts_pool3d5_avgmax_fwd(const float* x, float* out_avg, float* out_max, int B, int C, int D, int H, int W,
                                     long long x_bstride, long long x_cstride, long long avg_bstride, long long avg_cstride,
                                     long long max_bstride, long long max_cstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "pool3d5: non-positive size");
  TS_REQUIRE(B <= 65535 && C <= 65535, TS_ERR_UNSUPPORTED, "pool3d5: grid too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(out_avg); TS_REQUIRE_PTR(out_max);
  Pool5 p;
  p.C = C; p.D = D; p.H = H; p.W = W;
  p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.avg_bstride = avg_bstride; p.avg_cstride = avg_cstride;
  p.max_bstride = max_bstride; p.max_cstride = max_cstride;
  const int tiles = ((H + PT_Y - 1) / PT_Y) * ((W + PT_X - 1) / PT_X);
  hipLaunchKernelGGL(pool5_avgmax_kernel, dim3(tiles, C, B), dim3(256), 0, ts::as_stream(stream), x, out_avg, out_max, p);
  return ts::launched("pool5_avgmax_kernel");
}

extern "C" int ts_merge_candidates_fwd(const float* volume, const float* sample, const float* mem_sample, const float* mem_cost,
                                       const float* past_w, const float* past_scale, const float* past_shift,
                                       float* out_sample, float* out_volume, int B, int C, int D0, int K, int H, int W,
                                       long long vol_bstride, long long vol_cstride, long long out_bstride,
                                       long long out_cstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && D0 > 0 && K >= 0 && H > 0 && W > 0, TS_ERR_SHAPE, "merge_candidates: non-positive size");
  TS_REQUIRE(D0 + K <= MERGE_DMAX, TS_ERR_UNSUPPORTED, "merge_candidates: more than %d candidates", MERGE_DMAX);
  TS_REQUIRE_PTR(volume); TS_REQUIRE_PTR(out_sample); TS_REQUIRE_PTR(out_volume);
  if (K > 0) { TS_REQUIRE_PTR(past_w); TS_REQUIRE_PTR(past_scale); TS_REQUIRE_PTR(past_shift); }
  Merge p;
  p.B = B; p.C = C; p.D0 = D0; p.K = K; p.HW = H * W; p.implicit_samples = sample ? 0 : 1;
  p.vol_bstride = vol_bstride; p.vol_cstride = vol_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  const dim3 grid(grid_for(static_cast<long long>(B) * H * W, 256), (C + MERGE_CPB - 1) / MERGE_CPB);
#define TS_MERGE(DMV)                                                                                              \
  hipLaunchKernelGGL(merge_candidates_kernel<DMV>, grid, dim3(256), 0, ts::as_stream(stream), volume, sample,      \
                     mem_sample, mem_cost, past_w, past_scale, past_shift, out_sample, out_volume, p)
  if (D0 + K <= 8) TS_MERGE(8);
  else if (D0 + K <= 14) TS_MERGE(14);
  else TS_MERGE(MERGE_DMAX);
#undef TS_MERGE
  return ts::launched("merge_candidates_kernel");
}

// ---- backward of the two softmax-weighted upsamplers (training form) -------------------------------------------------
// ConvexUpsample: out = sum_k p_k v_k, p = softmax_k(logits), v_k = disp(y + k/3 - 1, x + k%3 - 1) * scale (0 outside).
//   d logit_k = g p_k (v_k - out): every logit belongs to exactly one output pixel -> one lane per output pixel, plain stores
//   d disp    = scale * sum over the 9 r^2 (tap, output pixel) pairs that read the pixel, softmax recomputed -> one lane per
//               low-resolution pixel gathers (deterministic, no atomics)
namespace {
__global__ void __launch_bounds__(256)
convex_upsample_bwd_mask_kernel(const float* __restrict__ mask, const float* __restrict__ disp, const float* __restrict__ g,
                                float* __restrict__ gmask, int B, int H, int W, int r, float disp_scale) {
  const int HW = H * W, Ho = H * r, Wo = W * r;
  const long long n = static_cast<long long>(B) * Ho * Wo;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ox = static_cast<int>(i % Wo);
  const long long t = i / Wo;
  const int oy = static_cast<int>(t % Ho), b = static_cast<int>(t / Ho);
  const int y = oy / r, ry = oy - y * r, x = ox / r, rx = ox - x * r;
  const size_t base = (static_cast<size_t>(b) * 9 * r * r + ry * r + rx) * HW + static_cast<size_t>(y) * W + x;
  const float* dp = disp + static_cast<size_t>(b) * HW;
  float m[9], v[9], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = mask[base + static_cast<size_t>(k) * r * r * HW]; mx = fmaxf(mx, m[k]); }
  float den = 0.f, acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    m[k] = expf(m[k] - mx);
    den += m[k];
    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
    const float dv = dp[min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
    v[k] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? dv * disp_scale : 0.f;
    acc += m[k] * v[k];
  }
  const float inv = 1.f / den, out = acc * inv, gv = g[i];
#pragma unroll
  for (int k = 0; k < 9; ++k) gmask[base + static_cast<size_t>(k) * r * r * HW] = gv * m[k] * inv * (v[k] - out);
}

__global__ void __launch_bounds__(256)
convex_upsample_bwd_disp_kernel(const float* __restrict__ mask, const float* __restrict__ g, float* __restrict__ gdisp, int B, int H,
                                int W, int r, float disp_scale) {
  const int HW = H * W, Ho = H * r, Wo = W * r;
  const long long n = static_cast<long long>(B) * HW;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xx = static_cast<int>(i % W);
  const long long t = i / W;
  const int yy = static_cast<int>(t % H), b = static_cast<int>(t / H);
  float acc = 0.f;
  for (int k = 0; k < 9; ++k) {
    const int y = yy - (k / 3 - 1), x = xx - (k % 3 - 1);           // the pixel whose tap k reads (yy, xx)
    if (y < 0 || y >= H || x < 0 || x >= W) continue;
    for (int q = 0; q < r * r; ++q) {
      const size_t base = (static_cast<size_t>(b) * 9 * r * r + q) * HW + static_cast<size_t>(y) * W + x;
      float mx = -INFINITY, mk = 0.f;
      float e[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) { e[j] = mask[base + static_cast<size_t>(j) * r * r * HW]; mx = fmaxf(mx, e[j]); }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < 9; ++j) { e[j] = expf(e[j] - mx); den += e[j]; mk = (j == k) ? e[j] : mk; }
      const int oy = y * r + q / r, ox = x * r + q % r;
      acc += g[(static_cast<size_t>(b) * Ho + oy) * Wo + ox] * mk / den;
    }
  }
  gdisp[i] = acc * disp_scale;
}

// UNet.upsample: out = sum_k p_k bil_k, p = softmax over the 9 logit planes at the output pixel, bil_k = bilinear (align_corners)
// sample of nb_k(y, x) = disp(y + k/3 - 1, x + k%3 - 1) * Wo / w.
//   pass 1 (one lane per output pixel): d logit_k = g p_k (bil_k - out), and gp_k = g p_k into the workspace
//   pass 2 (one lane per low-resolution pixel): d disp = Wo/w * sum_k sum_{output pixels in the bilinear footprint of
//           (yy - dy_k, xx - dx_k)} gp_k * weight   (gather, deterministic)
__global__ void __launch_bounds__(256)
unet_upsample_bwd_mask_kernel(const float* __restrict__ mask, const float* __restrict__ disp, const float* __restrict__ g,
                              float* __restrict__ gmask, float* __restrict__ gp, int B, int h, int w, int Ho, int Wo, float sh,
                              float sw) {
  const long long n = static_cast<long long>(B) * Ho * Wo;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t HWo = static_cast<size_t>(Ho) * Wo;
  const int ox = static_cast<int>(i % Wo);
  const long long t = i / Wo;
  const int oy = static_cast<int>(t % Ho), b = static_cast<int>(t / Ho);
  const size_t mbase = static_cast<size_t>(b) * 9 * HWo + static_cast<size_t>(oy) * Wo + ox;
  const float* dp = disp + static_cast<size_t>(b) * h * w;
  int y0, y1, x0, x1;
  float ly, lx;
  lin_src(sh, oy, h, y0, y1, ly);
  lin_src(sw, ox, w, x0, x1, lx);
  const float vs = static_cast<float>(Wo) / static_cast<float>(w);
  float m[9], bil[9], mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 9; ++k) { m[k] = mask[mbase + k * HWo]; mx = fmaxf(mx, m[k]); }
  float den = 0.f, acc = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int dy = k / 3 - 1, dx = k % 3 - 1;
    auto nb = [&](int y, int x) {
      const int yy = y + dy, xx = x + dx;
      const float dv = dp[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)];
      return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? dv * vs : 0.f;
    };
    const float top = (1.f - lx) * nb(y0, x0) + lx * nb(y0, x1);
    const float bot = (1.f - lx) * nb(y1, x0) + lx * nb(y1, x1);
    bil[k] = (1.f - ly) * top + ly * bot;
    m[k] = expf(m[k] - mx);
    den += m[k];
    acc += m[k] * bil[k];
  }
  const float inv = 1.f / den, out = acc * inv, gv = g[i];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float pk = m[k] * inv;
    gmask[mbase + k * HWo] = gv * pk * (bil[k] - out);
    gp[mbase + k * HWo] = gv * pk;
  }
}

__global__ void __launch_bounds__(256)
unet_upsample_bwd_disp_kernel(const float* __restrict__ gp, float* __restrict__ gdisp, int B, int h, int w, int Ho, int Wo, float sh,
                              float sw) {
  // 16 lanes per low-resolution pixel, lane k < 9 = tap k (one lane walking all nine footprints of ~9x9 full-resolution pixels left
  // this kernel at 163 us on 128 workgroups); fixed-order shuffle tree: deterministic
  const long long n = static_cast<long long>(B) * h * w;
  const long long gi = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long i = gi >> 4;
  const int k0 = static_cast<int>(gi & 15);
  if (i >= n) return;
  const size_t HWo = static_cast<size_t>(Ho) * Wo;
  const int xx = static_cast<int>(i % w);
  const long long t = i / w;
  const int yy = static_cast<int>(t % h), b = static_cast<int>(t / h);
  const float ih = sh > 0.f ? 1.f / sh : 0.f, iw = sw > 0.f ? 1.f / sw : 0.f;
  float acc = 0.f;
  for (int k = k0; k < 9; k += 16) {
    const int y = yy - (k / 3 - 1), x = xx - (k % 3 - 1);           // the nb_k pixel that holds disp(yy, xx)
    if (y < 0 || y >= h || x < 0 || x >= w) continue;
    int oy0 = sh > 0.f ? static_cast<int>(floorf((static_cast<float>(y) - 1.f) * ih)) : 0;
    int oy1 = sh > 0.f ? static_cast<int>(ceilf((static_cast<float>(y) + 1.f) * ih)) : Ho - 1;
    int ox0 = sw > 0.f ? static_cast<int>(floorf((static_cast<float>(x) - 1.f) * iw)) : 0;
    int ox1 = sw > 0.f ? static_cast<int>(ceilf((static_cast<float>(x) + 1.f) * iw)) : Wo - 1;
    oy0 = max(oy0, 0); oy1 = min(oy1, Ho - 1); ox0 = max(ox0, 0); ox1 = min(ox1, Wo - 1);
    const float* gk = gp + (static_cast<size_t>(b) * 9 + k) * HWo;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int a0, a1;
      float ly;
      lin_src(sh, oy, h, a0, a1, ly);
      const float wy = (a0 == y ? 1.f - ly : 0.f) + (a1 == y ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int c0, c1;
        float lx;
        lin_src(sw, ox, w, c0, c1, lx);
        const float wx = (c0 == x ? 1.f - lx : 0.f) + (c1 == x ? lx : 0.f);
        if (wx != 0.f) acc += gk[static_cast<size_t>(oy) * Wo + ox] * wy * wx;
      }
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (k0 == 0) gdisp[i] = acc * static_cast<float>(Wo) / static_cast<float>(w);
}
}  // namespace

extern "C" int ts_convex_upsample_bwd(const float* mask, const float* disp, const float* grad_out, float* grad_mask,
                                      float* grad_disp, int B, int H, int W, int factor, float disp_scale, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && factor >= 1, TS_ERR_SHAPE, "convex_upsample_bwd: bad size");
  TS_REQUIRE_PTR(mask); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(grad_out);
  if (grad_mask) {
    hipLaunchKernelGGL(convex_upsample_bwd_mask_kernel, dim3(exact_grid(static_cast<long long>(B) * H * W * factor * factor)),
                       dim3(256), 0, ts::as_stream(stream), mask, disp, grad_out, grad_mask, B, H, W, factor, disp_scale);
    if (int rc = ts::launched("convex_upsample_bwd_mask_kernel")) return rc;
  }
  if (grad_disp) {
    hipLaunchKernelGGL(convex_upsample_bwd_disp_kernel, dim3(exact_grid(static_cast<long long>(B) * H * W)), dim3(256), 0,
                       ts::as_stream(stream), mask, grad_out, grad_disp, B, H, W, factor, disp_scale);
    if (int rc = ts::launched("convex_upsample_bwd_disp_kernel")) return rc;
  }
  return TS_OK;
}

// workspace: B * 9 * Ho * Wo floats (the softmax-weighted output gradient per tap)
extern "C" int ts_unet_upsample_bwd(const float* mask, const float* disp, const float* grad_out, float* grad_mask, float* grad_disp,
                                    void* workspace, int B, int h, int w, int Ho, int Wo, void* stream) {
  TS_REQUIRE(B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, TS_ERR_SHAPE, "unet_upsample_bwd: bad size");
  TS_REQUIRE_PTR(mask); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(grad_out); TS_REQUIRE_PTR(grad_mask); TS_REQUIRE_PTR(workspace);
  float* gp = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(unet_upsample_bwd_mask_kernel, dim3(exact_grid(static_cast<long long>(B) * Ho * Wo)), dim3(256), 0,
                     ts::as_stream(stream), mask, disp, grad_out, grad_mask, gp, B, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo));
  if (int rc = ts::launched("unet_upsample_bwd_mask_kernel")) return rc;
  if (grad_disp) {
    hipLaunchKernelGGL(unet_upsample_bwd_disp_kernel, dim3(exact_grid(static_cast<long long>(B) * h * w * 16)), dim3(256), 0,
                       ts::as_stream(stream), gp, grad_disp, B, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo));
    if (int rc = ts::launched("unet_upsample_bwd_disp_kernel")) return rc;
  }
  return TS_OK;
}

extern "C" int ts_convex_upsample_fwd(const float* mask, const float* disp, float* out, int B, int H, int W, int factor,
                                      float disp_scale, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && factor >= 1, TS_ERR_SHAPE, "convex_upsample: bad size");
  TS_REQUIRE_PTR(mask); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(out);
  hipLaunchKernelGGL(convex_upsample_kernel, dim3(grid_for(static_cast<long long>(B) * H * W * factor * factor, 256)), dim3(256), 0,
                     ts::as_stream(stream), mask, disp, out, B, H, W, factor, disp_scale,
                     static_cast<float*>(nullptr), static_cast<float*>(nullptr), static_cast<float*>(nullptr), 0.f, 0, 0);
  return ts::launched("convex_upsample_kernel");
}

// ts_convex_upsample_fwd followed by ts_range_candidates_fwd on its output, as one launch
extern "C" int ts_convex_upsample_candidates_fwd(const float* mask, const float* disp, float* out, float* low, float* high,
                                                 float* candidates, int B, int H, int W, int factor, float disp_scale,
                                                 float range, int channel_offset, int channels_total, void* stream) {
  TS_REQUIRE(B > 0 && H > 0 && W > 0 && factor >= 1, TS_ERR_SHAPE, "convex_upsample_candidates: bad size");
  TS_REQUIRE(channel_offset >= 0 && channel_offset + 5 <= channels_total, TS_ERR_SHAPE, "convex_upsample_candidates: bad channel slice");
  TS_REQUIRE_PTR(mask); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(out); TS_REQUIRE_PTR(low); TS_REQUIRE_PTR(high); TS_REQUIRE_PTR(candidates);
  hipLaunchKernelGGL(convex_upsample_kernel, dim3(grid_for(static_cast<long long>(B) * H * W * factor * factor, 256)), dim3(256), 0,
                     ts::as_stream(stream), mask, disp, out, B, H, W, factor, disp_scale, low, high, candidates, range,
                     channel_offset, channels_total);
  return ts::launched("convex_upsample_kernel");
}

extern "C" int ts_unet_upsample_fwd(const float* mask, const float* disp, float* out, int B, int h, int w, int Ho, int Wo,
                                    void* stream) {
  TS_REQUIRE(B > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, TS_ERR_SHAPE, "unet_upsample: bad size");
  TS_REQUIRE_PTR(mask); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(out);
  const bool vec = (Wo % 4 == 0) && ts::aligned16(mask) && ts::aligned16(out);
  if (vec)
    hipLaunchKernelGGL(unet_upsample_kernel<4>, dim3(grid_for(static_cast<long long>(B) * Ho * (Wo / 4), 256)), dim3(256), 0,
                       ts::as_stream(stream), mask, disp, out, B, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo));
  else
    hipLaunchKernelGGL(unet_upsample_kernel<1>, dim3(grid_for(static_cast<long long>(B) * Ho * Wo, 256)), dim3(256), 0,
                       ts::as_stream(stream), mask, disp, out, B, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo));
  return ts::launched("unet_upsample_kernel");
}

extern "C" int ts_resize_bilinear_fwd(const float* x, float* out, int B, int C, int h, int w, int Ho, int Wo, float value_scale,
                                      long long out_bstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, TS_ERR_SHAPE, "resize_bilinear: bad size");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(out);
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for(static_cast<long long>(B) * C * Ho * Wo, 256)), dim3(256), 0,
                     ts::as_stream(stream), x, out, B * C, C, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo), value_scale,
                     out_bstride);
  return ts::launched("resize_bilinear_kernel");
}

extern "C" int ts_resize_bilinear_pair_fwd(const float* x0, const float* x1, float* out0, float* out1, int B, int C, int h,
                                           int w, int Ho, int Wo, float value_scale0, float value_scale1, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0 && Ho > 0 && Wo > 0, TS_ERR_SHAPE, "resize_bilinear_pair: bad size");
  TS_REQUIRE_PTR(x0); TS_REQUIRE_PTR(x1); TS_REQUIRE_PTR(out0); TS_REQUIRE_PTR(out1);
  hipLaunchKernelGGL(resize_bilinear_pair_kernel, dim3(grid_for(static_cast<long long>(B) * C * Ho * Wo, 256), 2), dim3(256), 0,
                     ts::as_stream(stream), x0, x1, out0, out1, B * C, h, w, Ho, Wo, ac_scale(h, Ho), ac_scale(w, Wo),
                     value_scale0, value_scale1);
  return ts::launched("resize_bilinear_pair_kernel");
}

// ---- backward entries (training) ------------------------------------------------------------------
// d(out)/d(a, add) of ts_resize3d_add_act_fwd.  grad_a [B,C,Da,Ha,Wa] is OVERWRITTEN (zero-filled, then fp32
// atomics); grad_add (may be NULL) [B,C,D,H,W].  All five tensors dense NCDHW.
extern "C" int ts_resize3d_add_act_bwd(const float* a, const float* add, const float* grad_out, float* grad_a, float* grad_add,
                                       int B, int C, int Da, int Ha, int Wa, int D, int H, int W, int act, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && Da > 0 && Ha > 0 && Wa > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "resize3d_bwd: non-positive size");
  TS_REQUIRE(B <= 65535 && C <= 65535, TS_ERR_UNSUPPORTED, "resize3d_bwd: grid too large");
  TS_REQUIRE_PTR(a); TS_REQUIRE_PTR(grad_out); TS_REQUIRE_PTR(grad_a);
  Resize3B q;
  Resize3& p = q.f;
  p.C = C; p.Da = Da; p.Ha = Ha; p.Wa = Wa; p.D = D; p.H = H; p.W = W;
  p.sd = ac_scale(Da, D); p.sh = ac_scale(Ha, H); p.sw = ac_scale(Wa, W);
  p.act = act;
  const long long na = static_cast<long long>(Da) * Ha * Wa, n = static_cast<long long>(D) * H * W;
  p.a_bstride = C * na; p.a_cstride = na; p.b_bstride = C * n; p.b_cstride = n; p.o_bstride = C * n; p.o_cstride = n;
  q.g_bstride = C * n; q.g_cstride = n; q.ga_bstride = C * na; q.ga_cstride = na; q.gb_bstride = C * n; q.gb_cstride = n;
  hipStream_t st = ts::as_stream(stream);
  hipError_t e = hipMemsetAsync(grad_a, 0, static_cast<size_t>(B) * C * na * sizeof(float), st);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "resize3d_bwd: %s", hipGetErrorString(e));
  int blocks = static_cast<int>((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(resize_add_act_bwd_kernel, dim3(blocks, C, B), dim3(256), 0, st, a, add, grad_out, grad_a, grad_add, q);
  return ts::launched("resize_add_act_bwd_kernel");
}

// d/dx of ts_pool3d5_avgmax_fwd: grad_x = box5^3(grad_avg) / 125 + scatter of grad_max to each window's arg-max.
// Dense [B,C,D,H,W] tensors; grad_x OVERWRITTEN.
extern "C" int ts_pool3d5_avgmax_bwd(const float* x, const float* grad_avg, const float* grad_max, float* grad_x,
                                     int B, int C, int D, int H, int W, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "pool3d5_bwd: non-positive size");
  TS_REQUIRE(B <= 65535 && C <= 65535, TS_ERR_UNSUPPORTED, "pool3d5_bwd: grid too large");
  TS_REQUIRE(static_cast<long long>(D) * H * W < (1ll << 31), TS_ERR_UNSUPPORTED, "pool3d5_bwd: plane set too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(grad_avg); TS_REQUIRE_PTR(grad_max); TS_REQUIRE_PTR(grad_x);
  Pool5 p;
  p.C = C; p.D = D; p.H = H; p.W = W;
  const long long n = static_cast<long long>(D) * H * W;
  p.x_bstride = C * n; p.x_cstride = n; p.avg_bstride = C * n; p.avg_cstride = n; p.max_bstride = C * n; p.max_cstride = n;
  const int tiles = ((H + PT_Y - 1) / PT_Y) * ((W + PT_X - 1) / PT_X);
  hipStream_t st = ts::as_stream(stream);
  hipLaunchKernelGGL(pool5_avgmax_kernel, dim3(tiles, C, B), dim3(256), 0, st, grad_avg, grad_x, static_cast<float*>(nullptr), p);
  if (int rc = ts::launched("pool5_avgmax_kernel")) return rc;
  hipLaunchKernelGGL(pool5_max_bwd_kernel, dim3(tiles, C, B), dim3(256), 0, st, x, grad_max, grad_x, p);
  return ts::launched("pool5_max_bwd_kernel");
}

// backward of the sort + gather half of ts_merge_candidates_fwd (K = 0 form: sample [B,DT,H,W], volume
// [B,C,DT,H,W] dense): grad_volume / grad_sample (may be NULL) OVERWRITTEN; grad_out_sample may be NULL.
extern "C" int ts_merge_candidates_bwd(const float* sample, const float* grad_out_volume, const float* grad_out_sample,
                                       float* grad_volume, float* grad_sample, int B, int C, int DT, int H, int W, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && DT > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "merge_candidates_bwd: non-positive size");
  TS_REQUIRE(DT <= MERGE_DMAX, TS_ERR_UNSUPPORTED, "merge_candidates_bwd: more than %d candidates", MERGE_DMAX);
  TS_REQUIRE_PTR(sample); TS_REQUIRE_PTR(grad_out_volume); TS_REQUIRE_PTR(grad_volume);
  const dim3 grid(grid_for(static_cast<long long>(B) * H * W, 256), (C + MERGE_CPB - 1) / MERGE_CPB);
#define TS_MERGE_B(DMV)                                                                                       \
  hipLaunchKernelGGL(merge_bwd_kernel<DMV>, grid, dim3(256), 0, ts::as_stream(stream), sample, grad_out_volume, \
                     grad_out_sample, grad_volume, grad_sample, B, C, DT, H * W)
  if (DT <= 8) TS_MERGE_B(8);
  else if (DT <= 14) TS_MERGE_B(14);
  else TS_MERGE_B(MERGE_DMAX);
#undef TS_MERGE_B
  return ts::launched("merge_bwd_kernel");
}
