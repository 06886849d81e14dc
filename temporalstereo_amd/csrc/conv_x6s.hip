// K3 -- the bf16-split ("x6") form of the STRIDED and TRANSPOSED convolutions of the pyramid for gfx950 (MI355X):
//   XS_S2 : Conv3d (1,3,3), stride 2, padding 1                                  module.py:111-147 (DepthwiseConv3D, stride 2), basic_layers.py:194-235
//   XS_T3 : ConvTranspose3d (1,3,3), stride 2, padding 1, output_padding 1       module.py:149-184 (DepthwiseConvTranspose3D)
//   XS_T4 : ConvTranspose2d 4x4, stride 2, padding 1 (UNet deconv4 / deconv2)    module.py:453-457
// conv3d.hip runs these on the f32-input MFMA (1/16 of the bf16 rate on gfx950) -- VERDICT round 3, item 4 / "missing" 4.  Every fp32
// product is formed as in ig_conv_x6_kernel (conv3d.hip): a = a0 + a1 + a2 in bf16 parts, a b ~= six bf16 products accumulated in fp32
// by v_mfma_f32_16x16x32_bf16, a chunk's products summed apart from the running sum.
//
// What differs from the stride-1 kernel is the K layout: a chunk is EIGHT input channels and an MFMA's K = 32 is (four taps) x (8
// channels) -- lane group kq holds tap 4u + kq.  A stride-2 / transposed layer stages four input pixels per output pixel (or produces four
// outputs per input pixel), so its LDS tile per unit of matrix work is four times the stride-1 layer's; with 16-channel chunks one
// workgroup would fill a CU's LDS alone, and a lone workgroup's chunks are a chain of exposed load -> split -> commit -> multiply
// sequences (conv3d.hip, x6 split-K note).  Eight channels keep 2-3 workgroups co-resident (32-50 KB each).
//   XS_S2: workgroup = 4 x 32 OUTPUT pixels x (CB 16) channels; 9 taps = 3 steps, slots 9-11 multiply zero weights.
//          Staged tile: input rows 2 oy0 - 1 .. + 7, columns 2 ox0 - 4 .. + 67 as 18 aligned quads per row.
//   XS_T*: workgroup = 8 x 32 INPUT pixels x 16 channels, all four output parity classes (2 iy + pa, 2 ix + pb) from ONE staging
//          (conv3d.hip's MODE_HWT stages the tile once per class); a class is one step: k4 -> its 2 x 2 taps exactly,
//          k3 -> 1 | 2 | 2 | 4 taps, the rest zero weights.  Output leaves through LDS as whole 256-byte rows (the classes interleave
//          along x: stored directly they would be 4-byte stores at an 8-byte pitch).
// Needs W % 4 == 0 and Wo % 4 == 0 (aligned quads on both sides).
#include "conv_common.hpp"

namespace {

enum { XS_S2 = 0, XS_T3 = 1, XS_T4 = 2 };

template <int MODE, int TR = 8>       // TR: input rows of a transposed form's tile (8: two per wave; 4: one per wave)
struct XG {
  static constexpr bool T = MODE != XS_S2;
  static constexpr int in_rows = T ? TR + 2 : 9;
  static constexpr int QPR = T ? 10 : 18;              // aligned quads per staged row
  static constexpr int LCOLS = 4 * QPR;
  static constexpr int NPIX = in_rows * LCOLS, NPIXP = NPIX + 1;
  static constexpr int SLOTS = in_rows * QPR;          // (row, quad) staging slots
  static constexpr int NSLOT = T ? 16 : 12;            // tap slots per chunk (4 per MFMA step)
  static constexpr int NU = T ? 4 : 3;                 // MFMA steps per chunk (T: one per parity class)
  static constexpr int NPB = T ? TR / 2 : 2;           // 16-pixel blocks per wave
  static constexpr int EPI = T ? TR * 64 + 4 : 132;    // LDS pitch of a channel of the output staging
};
constexpr int kEpiS2 = 132;               // LDS pitches of the output staging (see the epilogues)

// which weight tap (index into w_t's tap dimension) slot s multiplies, -1: none (zero weights)
//   transposed forms, per axis and output parity: even -> option 0 = (k 1, d 0), option 1 = (k 3, d -1) [k4 only];
//                                                 odd  -> option 0 = (k 2, d 0), option 1 = (k 0, d +1)
inline int slot_tap(int mode, int s) {
  if (mode == XS_S2) return s < 9 ? s : -1;
  const int ks = mode == XS_T4 ? 4 : 3;
  const int cls = s >> 2, a = (s >> 1) & 1, b = s & 1, pa = cls >> 1, pb = cls & 1;
  auto k = [&](int par, int opt) { return par ? (opt ? 0 : 2) : (opt ? (ks == 4 ? 3 : -1) : 1); };
  const int ky = k(pa, a), kx = k(pb, b);
  return (ky < 0 || kx < 0) ? -1 : ky * ks + kx;
}

struct SlotMap { int tap[16]; };

// w_t fp32 [Cin][ntaps][wpad] -> w6 [chunk of 8][part 3][slot][coutp] x 8 bf16 (channels past Cin, empty slots: zero)
__global__ void __launch_bounds__(256)
weight_split6_g8_kernel(const float* __restrict__ w_t, u32x4* __restrict__ w6, int Cin, int ntaps, int wpad, int coutp, int nchunk,
                        int nslot, const SlotMap m) {
  const int n = nchunk * nslot * coutp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int co = i % coutp;
    const int r = i / coutp;
    const int slot = r % nslot, chunk = r / nslot;
    const int tap = m.tap[slot];
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = chunk * 8 + e;
      v[e] = (tap >= 0 && ci < Cin && co < wpad) ? w_t[(static_cast<size_t>(ci) * ntaps + tap) * wpad + co] : 0.f;
    }
    unsigned part[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split6(v[2 * e], v[2 * e + 1], part[0][e], part[1][e], part[2][e]);
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
      w6[((static_cast<size_t>(chunk) * 3 + pt) * nslot + slot) * coutp + co] = u32x4{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
  }
}

template <int MODE, int CB, int TR = 8>
__global__ void __launch_bounds__(256, TR == 4 ? 3 : 2)
ig_conv_x6s_kernel(const float* __restrict__ x, const u32x4* __restrict__ w6, const float* __restrict__ scale,
                   const float* __restrict__ shift, float* __restrict__ y, const IG p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds6[];
  using G = XG<MODE, TR>;
  constexpr bool T = G::T;
  static_assert(!T || CB == 1, "transposed forms: 16 output channels per workgroup (four classes of accumulators)");
  constexpr int LCOLS = G::LCOLS, NPIXP = G::NPIXP, NSLOT = G::NSLOT, NU = G::NU, NPB = G::NPB, SLOTS = G::SLOTS;
  constexpr int COB = CB * 16;
  constexpr int WV = 3 * NSLOT * COB, RWN = (WV + 255) / 256;
  constexpr int NCH = T ? 4 : 8;                        // channels one staging thread handles
  u32x4* in6 = lds6;                                    // [part][NPIXP]: 8 channels of one part per pixel
  u32x4* w6s = lds6 + 3 * NPIXP;                        // [part][slot][COB] (+ dump)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {   // XCD-banded order (ig_conv_x6_kernel): an XCD's L2 serves the halo rows its tiles share
    const unsigned total = gridDim.x * gridDim.y * gridDim.z, per = total / 8;
    if (p.xcd && lin < per * 8) lin = (lin % 8) * per + lin / 8;
  }
  const int tile = lin % gridDim.x;
  const int od = (lin / gridDim.x) % gridDim.y;
  const int bz = lin / (gridDim.x * gridDim.y);
  const int cog = bz % p.co_groups, b = bz / p.co_groups;
  const int co0 = cog * COB;
  // tile origin: T -> input pixels (8 x 32), S2 -> output pixels (4 x 32)
  const int ty0 = (tile / p.tiles_x) * (T ? TR : 4), tx0 = (tile % p.tiles_x) * 32;
  const int row0 = T ? ty0 - 1 : 2 * ty0 - 1, col0 = T ? tx0 - 4 : 2 * tx0 - 4;     // first staged input row / column
  const unsigned HW = static_cast<unsigned>(p.H) * p.W;
  const unsigned cstride_b = static_cast<unsigned>(p.in_cstride) * 4u;

  // ---- staging: T -> threads [0,128) channels 0-3, [128,256) channels 4-7 of a (row, quad) slot; S2 -> 162 threads, 8 channels each
  const int half = T ? static_cast<int>(threadIdx.x >> 7) : 0;
  const int sslot = T ? static_cast<int>(threadIdx.x & 127) : static_cast<int>(threadIdx.x);
  const bool stager = sslot < SLOTS;
  const int srow = sslot / G::QPR, squad = sslot - srow * G::QPR;
  unsigned goff = kOOB;
  {
    const int gy = row0 + srow, gx = col0 + 4 * squad;
    if (stager && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
      goff = (static_cast<unsigned>(od) * HW + static_cast<unsigned>(gy) * p.W + gx) * 4u;
  }
  const int lpix = srow * LCOLS + 4 * squad;

  unsigned woff[RWN];
  int wl[RWN];
#pragma unroll
  for (int q = 0; q < RWN; ++q) {
    const int v = threadIdx.x + 256 * q;
    const int col = v % COB, r = v / COB;               // r = part * NSLOT + slot
    const bool ok = v < WV && co0 + col < p.coutp;
    woff[q] = ok ? static_cast<unsigned>(r * p.coutp + co0 + col) * 16u : kOOB;
    wl[q] = v < WV ? v : WV;
  }
  const __amdgpu_buffer_rsrc_t xr = ig_rsrc(x + static_cast<long long>(b) * p.in_bstride, p.in_bytes);
  const __amdgpu_buffer_rsrc_t wr = ig_rsrc(w6, p.w_bytes);
  const unsigned wchunk_b = static_cast<unsigned>(3 * NSLOT * p.coutp) * 16u;

  // ---- fragment bases (u32x4 units).  B: this lane's pixel of each 16-pixel block + the tap of its lane group, per step
  int boff[NPB], toff[NU];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    if (T) boff[pb] = ((TR == 8 ? wave * 2 + (pb >> 1) : wave) + 1) * LCOLS + 4 + (pb & 1) * 16 + j;       // input pixel (ty0 + 2w + (pb>>1), tx0 + ...)
    else boff[pb] = (2 * wave) * LCOLS + 3 + 2 * (pb * 16 + j);                         // input pixel (2 oy - 1, 2 ox - 1): tap (0, 0)
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    if (T) {
      const int pa = u >> 1, pbit = u & 1, a = kq >> 1, bq = kq & 1;
      const int dy = a ? (pa ? 1 : (MODE == XS_T4 ? -1 : 0)) : 0, dx = bq ? (pbit ? 1 : (MODE == XS_T4 ? -1 : 0)) : 0;
      toff[u] = dy * LCOLS + dx;
    } else {
      const int s = 4 * u + kq;
      toff[u] = s < 9 ? (s / 3) * LCOLS + (s % 3) : 0;
    }
  }
  const int aoff = kq * COB + j;

  v4f acc[T ? 4 : 1][CB][NPB];
#pragma unroll
  for (int c = 0; c < (T ? 4 : 1); ++c)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) acc[c][cb][pb] = v4f{0.f, 0.f, 0.f, 0.f};
  float esc[CB][4], esh[CB][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = min(co0 + cb * 16 + kq * 4 + r, p.Cout - 1);
      esc[cb][r] = scale ? scale[co] : 1.f;
      esh[cb][r] = shift ? shift[co] : 0.f;
    }

  v4f rin[NCH];
  u32x4 rw[RWN];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // channels past Cin re-read the last real one; their weights are zero
      const unsigned co = static_cast<unsigned>(min(c0 + half * 4 + c, p.Cin - 1)) * cstride_b;
      rin[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, goff == kOOB ? kOOB : goff + co, 0, 0));
    }
    const unsigned wso = static_cast<unsigned>(c0 / 8) * wchunk_b;
#pragma unroll
    for (int q = 0; q < RWN; ++q) rw[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff[q], wso, 0);
  };
  auto commit = [&]() {
    if (stager) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned part[3][NCH / 2];
#pragma unroll
        for (int e = 0; e < NCH / 2; ++e) split6(rin[2 * e][i], rin[2 * e + 1][i], part[0][e], part[1][e], part[2][e]);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          if constexpr (T) reinterpret_cast<u32x2*>(in6 + pt * NPIXP + lpix + i)[half] = u32x2{part[pt][0], part[pt][1]};
          else in6[pt * NPIXP + lpix + i] = u32x4{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RWN; ++q) w6s[wl[q]] = rw[q];
  };

  fetch(0);
  for (int c0 = 0; c0 < p.Cin; c0 += 8) {
    __syncthreads();
    commit();
    __syncthreads();
    if (c0 + 8 < p.Cin) fetch(c0 + 8);
    // a chunk's products are summed apart and added to the running sum in fp32 (ig_conv_x6_kernel: the matrix core aligns an
    // instruction's products AND its accumulator to the largest addend)
    v4f part[CB][NPB];
    auto zero_part = [&]() {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) part[cb][pb] = v4f{0.f, 0.f, 0.f, 0.f};
    };
    zero_part();
    bf16x8 a[2][3][CB], bv[2][3][NPB];
    auto load_frag = [&](int u, int buf) {
#pragma unroll
      for (int pt = 0; pt < 3; ++pt) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) a[buf][pt][cb] = __builtin_bit_cast(bf16x8, w6s[(pt * NSLOT + 4 * u) * COB + aoff + cb * 16]);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) bv[buf][pt][pb] = __builtin_bit_cast(bf16x8, in6[pt * NPIXP + boff[pb] + toff[u]]);
      }
    };
    load_frag(0, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u + 1 < NU) load_frag(u + 1, (u + 1) & 1);
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb)
            part[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u & 1][PA[t]][cb], bv[u & 1][PB[t]][pb], part[cb][pb], 0, 0, 0);
      if (T) {           // a step is a parity class
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) acc[u][cb][pb] += part[cb][pb];
        zero_part();
      }
    }
    if (!T) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[0][cb][pb] += part[cb][pb];
    }
  }

  // ---- epilogue through LDS: whole quads of one channel row per store instruction ----
  const unsigned hw_o = static_cast<unsigned>(p.Ho) * p.Wo;
  const __amdgpu_buffer_rsrc_t yr = ig_rsrc(y + static_cast<long long>(b) * p.out_bstride, p.out_bytes);
  const unsigned ocs = static_cast<unsigned>(p.out_cstride) * 4u;
  const unsigned obase = static_cast<unsigned>(od) * hw_o;
  float* epi = reinterpret_cast<float*>(lds6);
  if constexpr (!T) {
    __syncthreads();                                    // the last chunk's fragments are consumed
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + cb * 16 + kq * 4 + r;
          epi[(cb * 16 + kq * 4 + r) * kEpiS2 + wave * 32 + pb * 16 + j] =
              apply_act(acc[0][cb][pb][r] * esc[cb][r] + esh[cb][r], p.act, p.act_param, co);
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CB * 2; ++it) {
      const int idx = static_cast<int>(threadIdx.x) + 256 * it;
      const int col = idx >> 5, q = idx & 31;
      const int oy = ty0 + (q >> 3), ox = tx0 + (q & 7) * 4;
      const u32x4 val = *reinterpret_cast<const u32x4*>(epi + col * kEpiS2 + q * 4);
      const int co = co0 + col;
      const unsigned off = (oy < p.Ho && ox < p.Wo && co < p.Cout)
          ? (obase + static_cast<unsigned>(oy) * p.Wo + ox) * 4u + static_cast<unsigned>(co) * ocs : kOOB;
      __builtin_amdgcn_raw_buffer_store_b128(val, yr, off, 0, 0);
    }
  } else {
#pragma unroll
    for (int pa = 0; pa < 2; ++pa) {
      __syncthreads();                                  // fragments / the previous half's rows are consumed
#pragma unroll
      for (int pbit = 0; pbit < 2; ++pbit)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + kq * 4 + r;
            epi[(kq * 4 + r) * G::EPI + (TR == 8 ? wave * 2 + (pb >> 1) : wave) * 64 + 2 * ((pb & 1) * 16 + j) + pbit] =
                apply_act(acc[pa * 2 + pbit][0][pb][r] * esc[0][r] + esh[0][r], p.act, p.act_param, co);
          }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < TR; ++it) {
        const int idx = static_cast<int>(threadIdx.x) + 256 * it;
        const int col = idx / (TR * 16), rem = idx % (TR * 16);
        const int row = rem >> 4, q = rem & 15;
        const int iy = ty0 + row, oy = 2 * iy + pa, ox = 2 * tx0 + 4 * q;
        const u32x4 val = *reinterpret_cast<const u32x4*>(epi + col * G::EPI + row * 64 + q * 4);
        const int co = co0 + col;
        const unsigned off = (iy < p.H && oy < p.Ho && ox < p.Wo && co < p.Cout)
            ? (obase + static_cast<unsigned>(oy) * p.Wo + ox) * 4u + static_cast<unsigned>(co) * ocs : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(val, yr, off, 0, 0);
      }
    }
  }
}

template <int MODE, int CB, int TR = 8>
int launch_x6s(const float* x, const void* w6, const float* scale, const float* shift, float* y, const IG& p, dim3 grid, hipStream_t st) {
  using G = XG<MODE, TR>;
  constexpr size_t main_b = (static_cast<size_t>(3) * G::NPIXP + 3 * G::NSLOT * CB * 16 + 1) * 16;
  constexpr size_t epi_b = G::T ? static_cast<size_t>(16) * G::EPI * 4 : static_cast<size_t>(CB) * 16 * kEpiS2 * 4;
  constexpr size_t lds = main_b > epi_b ? main_b : epi_b;
  static_assert(lds <= 64 * 1024, "ig_conv_x6s_kernel: LDS tile");
  hipLaunchKernelGGL((ig_conv_x6s_kernel<MODE, CB, TR>), grid, dim3(256), lds, st, x, static_cast<const u32x4*>(w6), scale, shift, y, p);
  return ts::launched("ig_conv_x6s_kernel");
}

inline int coutp6(int Cout) { return (Cout + 15) / 16 * 16; }
inline int mode_taps(int mode) { return mode == XS_T4 ? 16 : 9; }
inline int mode_slots(int mode) { return mode == XS_S2 ? 12 : 16; }
}  // namespace

extern "C" int ts_conv3d_hw_x6s_supported(int Cin, int Cout, int H, int W, int mode) {
  if (mode < 0 || mode > 2 || Cin < 16 || Cout <= 0 || Cout > 512 || H <= 0 || W <= 0 || W % 4) return 0;
  if (mode == XS_S2) return (((W - 1) / 2 + 1) % 4) == 0;
  return 1;
}

extern "C" size_t ts_conv3d_hw_x6s_weight_bytes(int Cin, int Cout, int mode) {
  if (Cin <= 0 || Cout <= 0 || Cout > 512 || mode < 0 || mode > 2) return 0;
  return static_cast<size_t>((Cin + 7) / 8) * 3 * mode_slots(mode) * coutp6(Cout) * 16;
}

extern "C" int ts_conv3d_hw_x6s_weight_split(const float* w_t, void* w6, int Cin, int Cout, int w_pad, int mode, void* stream) {
  TS_REQUIRE(Cin > 0 && Cout > 0 && Cout <= 512 && w_pad >= Cout, TS_ERR_SHAPE, "conv3d_hw_x6s_weight_split: bad channel counts");
  TS_REQUIRE(mode >= 0 && mode <= 2, TS_ERR_SHAPE, "conv3d_hw_x6s_weight_split: unknown mode %d", mode);
  TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(w6);
  const int nchunk = (Cin + 7) / 8, nslot = mode_slots(mode), cp = coutp6(Cout);
  SlotMap m;
  for (int s = 0; s < 16; ++s) m.tap[s] = s < nslot ? slot_tap(mode, s) : -1;
  const int n = nchunk * nslot * cp;
  hipLaunchKernelGGL(weight_split6_g8_kernel, dim3((n + 255) / 256), dim3(256), 0, ts::as_stream(stream), w_t,
                     static_cast<u32x4*>(w6), Cin, mode_taps(mode), w_pad, cp, nchunk, nslot, m);
  return ts::launched("weight_split6_g8_kernel");
}

// x [B,Cin,D,H,W] -> y [B,Cout,D,Ho,Wo]: mode 0 -> Ho = (H-1)/2+1 (stride-2 convolution), modes 1, 2 -> Ho = 2H (transposed forms; mode 2 is the
// 2-D 4x4 deconvolution, D = 1).  w6 from ts_conv3d_hw_x6s_weight_split; scale / shift [>= Cout] or null.  Strides in elements.
extern "C" int ts_conv3d_hw_x6s_fwd(const float* x, const void* w6, const float* scale, const float* shift, float* y,
                                    int B, int Cin, int Cout, int D, int H, int W, int mode, int act, float act_param,
                                    long long in_bstride, long long in_cstride, long long out_bstride, long long out_cstride,
                                    void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw_x6s: non-positive size");
  TS_REQUIRE(ts_conv3d_hw_x6s_supported(Cin, Cout, H, W, mode), TS_ERR_UNSUPPORTED,
             "conv3d_hw_x6s: needs Cin >= 16, Cout <= 512, W %% 4 == 0 and Wo %% 4 == 0 (Cin=%d Cout=%d W=%d mode=%d)", Cin, Cout, W, mode);
  TS_REQUIRE(act >= 0 && act <= 4, TS_ERR_SHAPE, "conv3d_hw_x6s: unknown activation");
  TS_REQUIRE(D <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw_x6s: grid too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w6); TS_REQUIRE_PTR(y);
  IG p;
  p.Cin = Cin; p.Cout = Cout; p.coutp = coutp6(Cout); p.D = D; p.H = H; p.W = W; p.Do = D;
  if (mode == XS_S2) { p.Ho = (H - 1) / 2 + 1; p.Wo = (W - 1) / 2 + 1; }
  else { p.Ho = 2 * H; p.Wo = 2 * W; }
  p.stride = 2; p.dil = 1; p.pad = 1; p.k = mode == XS_T4 ? 4 : 3; p.transposed = mode != XS_S2;
  p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  p.ksplit = 1; p.kspan = Cin; p.partial = nullptr; p.B = B; p.part_bytes = 0;
  p.addend = nullptr; p.add_bstride = p.add_cstride = p.add_dstride = 0;
  static const int xcd = env_not_zero("TS_X6_XCD") ? 1 : 0;
  p.xcd = xcd;
  {
    const unsigned long long in_b = (static_cast<unsigned long long>(Cin - 1) * in_cstride + static_cast<unsigned long long>(D) * H * W) * 4ull;
    const unsigned long long out_b = (static_cast<unsigned long long>(Cout - 1) * out_cstride + static_cast<unsigned long long>(D) * p.Ho * p.Wo) * 4ull;
    const unsigned long long w_b = ts_conv3d_hw_x6s_weight_bytes(Cin, Cout, mode);
    TS_REQUIRE(in_b < 0x7fffffffull && out_b < 0x7fffffffull && w_b < 0x7fffffffull && in_cstride >= 0 && out_cstride >= 0,
               TS_ERR_UNSUPPORTED, "conv3d_hw_x6s: a batch element spans 2 GiB or more");
    p.in_bytes = static_cast<unsigned>(in_b); p.out_bytes = static_cast<unsigned>(out_b); p.w_bytes = static_cast<unsigned>(w_b);
  }
  hipStream_t st = ts::as_stream(stream);
  TS_REQUIRE(static_cast<long long>(B) * ((p.coutp + 15) / 16) <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw_x6s: batch x channel groups beyond the grid's z extent");
  if (mode == XS_S2) {
    p.tiles_x = (p.Wo + 31) / 32;
    const int tiles = ((p.Ho + 3) / 4) * p.tiles_x;
    const int cb = Cout > 16 ? 2 : 1;
    p.co_groups = (p.coutp / 16 + cb - 1) / cb;
    const dim3 grid(tiles, D, B * p.co_groups);
    return cb == 2 ? launch_x6s<XS_S2, 2>(x, w6, scale, shift, y, p, grid, st) : launch_x6s<XS_S2, 1>(x, w6, scale, shift, y, p, grid, st);
  }
  p.tiles_x = (W + 31) / 32;
  p.co_groups = p.coutp / 16;
  // Tile rows.  A workgroup of the 8-row form is a long serial chain (barrier - commit - barrier - fragment reads - MFMAs per chunk,
  // then the two-half epilogue: ~15 us alone) at 226-236 VGPRs, two per CU; the 4-row form (one input row per wave: 150-160 VGPRs,
  // 24 KB of LDS, three per CU) halves the chain.  Measured (tools/x6s_bench.py, 8 -> 4 rows): deconv4 32 -> 32 on 136 x 240 (272
  // workgroups of 8 rows) 21.5 -> 17.8 us, the hourglasses' transposed layers 15.5 -> 10.8 and 19.4 -> 12.1; deconv2 32 -> 9 on 272 x 480
  // (510 workgroups: a full round of the 8-row form) 23.2 -> 24.9, batch 4 flat.  So: 4 rows below 3/4 of a round.  TS_X6S_TR=4 | 8 forces one.
  static const long long tr_env = env_ll("TS_X6S_TR", 0);
  const long long wgs8 = static_cast<long long>((H + 7) / 8) * p.tiles_x * D * B * p.co_groups;
  const long long tr = tr_env ? tr_env : (wgs8 < 3 * ts::kNumCU / 2 ? 4 : 8);
  if (tr == 4) {
    const dim3 grid(((H + 3) / 4) * p.tiles_x, D, B * p.co_groups);
    return mode == XS_T4 ? launch_x6s<XS_T4, 1, 4>(x, w6, scale, shift, y, p, grid, st) : launch_x6s<XS_T3, 1, 4>(x, w6, scale, shift, y, p, grid, st);
  }
  const int tiles = ((H + 7) / 8) * p.tiles_x;
  const dim3 grid(tiles, D, B * p.co_groups);
  return mode == XS_T4 ? launch_x6s<XS_T4, 1>(x, w6, scale, shift, y, p, grid, st) : launch_x6s<XS_T3, 1>(x, w6, scale, shift, y, p, grid, st);
}
