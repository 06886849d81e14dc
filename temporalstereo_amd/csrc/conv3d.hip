// K3 -- separable 3-D convolution building blocks of the aggregation pyramid for gfx950 (MI355X),
// inference form: BatchNorm folded to a per-channel (scale, shift), activation fused.
//
// Replaces, layer by layer, what the reference reaches through torch/cuDNN for
//   Conv3d / ConvTranspose3d wrappers   architecture/modeling/layers/basic_layers.py:194-235,340-388
//   DepthwiseConv3D (separable pair)    architecture/modeling/aggregation/TemporalStereo/module.py:111-147
//   DepthwiseConvTranspose3D            module.py:149-184
// Every 3-D convolution of the model has a kernel that is 1 along D or 1 along H,W, so two kernel
// families cover them all:
//   conv_hw : (1,3,3) taps, stride/dilation in H,W; the (B,D) planes are independent images.
//   conv_d  : (k,1,1) taps (k = 1,3,5) along D; every pixel column is independent.
// plus their stride-2 transposed forms.
//
// fp32 everywhere (the parity bar |dEPE| < 1e-3 px is an fp32 bar).  A first version ran these as
// direct convolutions on the vector ALU with the weights in SGPRs (s_load_dwordx16 -> v_fmac with a
// scalar operand); it stalled on lgkmcnt(0) -- SMEM returns out of order, so every tap waited for
// its scalar loads AND its LDS read -- and reached 3-15 TFLOP/s.  The implicit-GEMM MFMA kernel
// below replaced it (6-70 TFLOP/s on the same layers).
// Weights are pre-laid out [Cin][taps][CoutPad] (output channel contiguous) by the host.
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "conv_common.hpp"
#include "conv_x6p.hpp"

namespace {


// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on the matrix cores, all four conv families in one kernel:
//     out[co][px] = sum_{tap, ci} W[ci][tap][co] * X[ci][px shifted by tap]
// with v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation: parity is unchanged).  The point
// is not FLOPs (f32 MFMA == the VALU rate on gfx950) but operand delivery: one MFMA takes ONE LDS
// dword per lane for each operand and retires 1024 MACs, where the VALU form needs a weight per FMA.
//
//   workgroup = 256 output pixels x (CB*16) output channels; wave w owns 64 pixels as four 16-pixel
//   blocks and keeps 4*CB accumulator tiles.  Per (tap, 4 input channels): CB weight fragments + 4
//   input fragments from LDS, 4*CB MFMAs.
//   K loop = chunks of 8 input channels.  While a chunk is being multiplied, the next chunk's input
//   tile and weights are already in flight from global memory into registers (issued before the
//   MFMAs), and are written to LDS after them -- global latency hides under the matrix work, which is
//   what makes the many small layers of the hourglass cheap.
//   MODE_HW : (1,3,3) taps, stride 1|2, dilation 1|2.  Tile = 8 x 32 pixels of one depth plane.
//   MODE_HWT: ConvTranspose (1,3,3), stride 2, padding 1, output_padding 1, as four parity classes,
//             each an ordinary 1/2/2/4-tap convolution over the input grid.
//   MODE_D  : (k,1,1) taps, k = 1|3|5, stride/dilation along D, or its stride-2 transposed form.
//             Tile = 256 consecutive pixels of one output depth.
// ------------------------------------------------------------------------------------------------


// TP (MODE_D): pixels per workgroup, 256 or 64 (one 16-pixel block per wave: layers that run on a few dozen workgroups last as long as
// ONE workgroup's K loop, so the tile is cut instead of the grid being filled)
template <int MODE, int KT, int ST, int DL, int TP = 256>
struct Geom {
  static constexpr int NTR = (MODE == MODE_D) ? KT : 1;                 // planes staged per channel (MODE_D)
  static constexpr int TRW = (TP == 64) ? 4 : 8, TCW = (TP == 64) ? 16 : 32;   // MODE_HW output tile (rows x columns)
  static constexpr int in_rows = (MODE == MODE_HW) ? (TRW - 1) * ST + 2 * DL + 1 : ((MODE == MODE_HWT) ? ((KT == 16) ? TRW + 2 : TRW + 1) : 1);
  static constexpr int in_cols = (MODE == MODE_HW) ? (TCW - 1) * ST + 2 * DL + 1 : ((MODE == MODE_HWT) ? ((KT == 16) ? TCW + 2 : TCW + 1) : TP);
  static constexpr int pitch = (MODE == MODE_D) ? TP : (in_cols | 1);
  // staged elements per thread and channel: the tile is dealt linearly to the 256 threads (a (row, column-of-64) deal
  // wasted over half of the load slots on a 10 x 34 tile)
  static constexpr int RQ_HW = (in_rows * in_cols + 255) / 256;
  static constexpr int chan_raw = (MODE == MODE_D) ? NTR * TP : in_rows * pitch;
  static constexpr int pad0 = (16 - (chan_raw & 31)) & 31;
  // == 16 (mod 32): the four k-slots of a B fragment sit on disjoint banks; >= 1 spare word (dump slot)
  static constexpr int chan_elems = chan_raw + (pad0 ? pad0 : 32);
};


// ST / DL: stride and dilation of MODE_HW as compile-time constants, so that every LDS fragment read is
// `base register + immediate` (no address arithmetic between MFMAs).  NC: input channels per K chunk.
// PR (row pairing, Cout <= 8 on a stride-1 (1,3,3) layer): half of a 16-wide MFMA would multiply zero-padded channels
// 8..15.  Instead the 16 rows of the A operand are (output row, channel): rows 0-7 = channels 0-7 of output row R+DL, rows
// 8-15 = channels 0-7 of output row R.  The two output rows share input rows (R+DL and R+2DL of the tile), so one B fragment
// feeds both: per kx, input row R+rho*DL (rho = 0..3) is multiplied by [ W[ky=rho-1] | W[ky=rho] ] (zero where ky is
// outside 0..2) -- 12 "virtual taps" with 18 useful (tap, row) products in 24 half-tiles instead of 18 in 36, i.e. a third
// fewer MFMAs and a third fewer B-fragment reads.  The paired weight rows are assembled while the weights are staged.
template <int CB, int MODE, int KT, int ST, int DL, int NC, int PR = 0, int PF = 0, int TP = 256>
__global__ void __launch_bounds__(256)
ig_conv_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
               const float* __restrict__ shift, float* __restrict__ y, const IG p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  static_assert(!PR || (MODE == MODE_HW && KT == 9 && ST == 1 && CB == 1 && NC >= 8), "row pairing: stride-1 (1,3,3), Cout <= 8");
  static_assert(TP == 256 || (TP == 64 && (MODE == MODE_D || MODE == MODE_HWT || (MODE == MODE_HW && !PR))), "64-pixel tiles: not the row-paired form");
  using G = Geom<MODE, KT, ST, DL, TP>;
  constexpr int WP = (CB * 16) | 16;                  // weight row pitch (k-slots on disjoint banks)
  constexpr int NTR = G::NTR, RQ = (MODE == MODE_D) ? NTR : G::RQ_HW;
  constexpr int in_rows = G::in_rows, in_cols = G::in_cols, pitch = G::pitch, chan_elems = G::chan_elems;
  constexpr int KTW = PR ? 12 : KT;                   // taps as staged in LDS (virtual taps when pairing)
  constexpr int NPB = PR ? 2 : TP / 64;               // 16-pixel blocks per wave
  constexpr int WV = KTW * NC * CB * 4;               // 16-byte weight vectors per chunk
  constexpr int RWN = (WV + 255) / 256;

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  // XCD-aware placement (p.xcd): consecutive workgroup ids go round the eight XCDs, each with its own L2; workgroup L takes slot
  // (L % 8) * (total / 8) + L / 8 of the (z, plane, tile) order, so that an XCD works on a contiguous band and its L2 serves the
  // halo rows / neighbouring depth planes its workgroups share.
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (p.xcd) {
    const unsigned per = (gridDim.x * gridDim.y * gridDim.z) / 8;
    if (lin < per * 8) lin = (lin % 8) * per + lin / 8;
  }
  const int bx = lin % gridDim.x, by = (lin / gridDim.x) % gridDim.y, bz = lin / (gridDim.x * gridDim.y);
  const int cog = bz % p.co_groups;
  const int ks = (bz / p.co_groups) % p.ksplit, b = bz / (p.co_groups * p.ksplit);
  const int kbeg = ks * p.kspan, kend = min(p.Cin, kbeg + p.kspan);
  const int co0 = cog * CB * 16;
  const int od = by;

  // ---- geometry of this workgroup -------------------------------------------------------------
  int tile = bx, pa = 0, pbit = 0;
  if (MODE == MODE_HWT) { pa = (tile & 3) >> 1; pbit = tile & 1; tile >>= 2; }
  int iy0 = 0, ix0 = 0, ty0 = 0, tx0 = 0;
  if (MODE == MODE_HW) {
    ty0 = (tile / p.tiles_x) * G::TRW; tx0 = (tile % p.tiles_x) * G::TCW;
    iy0 = ty0 * ST - DL; ix0 = tx0 * ST - DL;                              // padding == dilation
  } else if (MODE == MODE_HWT) {
    // KT == 9 : k3 s2 p1 op1, taps reach rows/cols {0,+1};  KT == 16: k4 s2 p1, taps reach {-1,0,+1}
    ty0 = (tile / p.tiles_x) * G::TRW; tx0 = (tile % p.tiles_x) * G::TCW;
    iy0 = ty0 - ((KT == 16) ? 1 : 0); ix0 = tx0 - ((KT == 16) ? 1 : 0);
  }
  float* in_tile = lds;
  float* w_tile = lds + NC * chan_elems;               // [KT][NC][WP] (+ one dump vector)
  const unsigned HW = static_cast<unsigned>(p.H) * p.W;

  // MODE_D: which input plane and which weight tap each staged plane is (uniform)
  int plane_id[NTR], plane_wt[NTR], ntaps = KT;
  if (MODE == MODE_D) {
    if (p.transposed) {
      ntaps = 0;
      if ((od & 1) == 0) { plane_id[0] = od >> 1; plane_wt[0] = 1; ntaps = 1; }
      else {
        plane_id[0] = (od - 1) >> 1; plane_wt[0] = 2; ntaps = 1;
        if (NTR > 1 && ((od + 1) >> 1) < p.D) { plane_id[1] = (od + 1) >> 1; plane_wt[1] = 0; ntaps = 2; }
      }
#pragma unroll
      for (int t = 0; t < NTR; ++t) if (t >= ntaps) { plane_id[t] = -1; plane_wt[t] = 0; }
    } else {
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        const int id = od * p.stride + t * p.dil - p.pad;
        plane_id[t] = (id >= 0 && id < p.D) ? id : -1;
        plane_wt[t] = t;
      }
    }
  }
  const int px0 = (MODE == MODE_D) ? bx * TP : 0;

  // ---- staging geometry of this thread: where each of its RQ elements of a channel comes from (byte
  // offset inside the batch element, kOOB = zero padding) and where it goes in the LDS channel tile ----
  unsigned goff[RQ];
  int loff[RQ];
  if (MODE == MODE_D) {
    // MODE_D stages 16 bytes per lane: the 256 pixels of a (channel, plane) row are 64 quads, so one instruction of the workgroup
    // moves four rows -- wave w takes the channels c = w (mod 4) of the chunk, all NTR planes of each.  (Round 3: the kernel staged
    // one dword per lane, NC * NTR loads per thread and chunk at ~29 cycles of the CU's address unit each -- 1.15 us per 8-channel
    // chunk whatever the grid, and that, not the matrix work, was the duration of every (k,1,1) layer: 7 / 9.5 / 14 us at 16 / 32 /
    // 64 channels, tools/exp/conv_d_bench.py.  The x6 kernel had learnt the same lesson in round 2.)  A quad that straddles the end
    // of the plane carries foreign elements into pixels >= H W, which no output reads (the reduction is per pixel column).
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      const unsigned px = px0 + 4u * (threadIdx.x & 63u);
      goff[t] = (plane_id[t] >= 0 && px < HW) ? (static_cast<unsigned>(plane_id[t]) * HW + px) * 4u : kOOB;
      loff[t] = t * 256 + 4 * static_cast<int>(threadIdx.x & 63u);
    }
  } else {
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int e = static_cast<int>(threadIdx.x) + 256 * q;
      const bool slot = e < in_rows * in_cols;
      const int cy = e / in_cols, cx = e - cy * in_cols;
      const int gy = iy0 + cy, gx = ix0 + cx;
      const bool live = slot && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      goff[q] = live ? (static_cast<unsigned>(od) * HW + static_cast<unsigned>(gy) * p.W + gx) * 4u : kOOB;
      loff[q] = slot ? cy * pitch + cx : G::chan_raw;                        // dump slot in the channel padding
    }
  }
  // TP == 64: a (channel, plane) row is 16 quads, one instruction of the workgroup moves 16 rows; row r = c * NTR + t of the chunk
  constexpr int NI64 = (MODE == MODE_D && TP == 64) ? (NC * NTR + 15) / 16 : 1;
  unsigned go64[NI64];
  int lo64[NI64], ch64[NI64];
  if constexpr (MODE == MODE_D && TP == 64) {
#pragma unroll
    for (int i = 0; i < NI64; ++i) {
      const int r = static_cast<int>(threadIdx.x >> 4) + 16 * i;
      const bool ok = r < NC * NTR;
      const int c = r / NTR, t = r - c * NTR;
      int pid = -1;
#pragma unroll
      for (int u = 0; u < NTR; ++u) pid = (u == t) ? plane_id[u] : pid;
      const unsigned px = px0 + 4u * (threadIdx.x & 15u);
      go64[i] = (ok && pid >= 0 && px < HW) ? (static_cast<unsigned>(pid) * HW + px) * 4u : kOOB;
      ch64[i] = ok ? c : 0;
      lo64[i] = ok ? c * chan_elems + t * TP + 4 * static_cast<int>(threadIdx.x & 15u) : G::chan_raw;     // no row: the spare quad of channel 0
    }
  }
  // weights: vector v of a chunk = 4 consecutive output channels of (tap, ci)
  unsigned woff[RWN];
  int wci[RWN], wl[RWN];
#pragma unroll
  for (int q = 0; q < RWN; ++q) {
    const int v = threadIdx.x + 256 * q;
    const int co4 = v % (CB * 4), r = v / (CB * 4);
    const int ci = r % NC, tap = r / NC;
    if constexpr (PR) {
      // virtual tap (rho, kx); vectors 0,1 = rows 0-7 (output row R+DL, ky = rho-1), vectors 2,3 = rows 8-15 (row R, ky = rho)
      const int rho = tap / 3, kx = tap % 3, ky = rho - ((co4 >> 1) ? 0 : 1);
      const bool ok = v < WV && ky >= 0 && ky <= 2;
      woff[q] = ok ? static_cast<unsigned>((ci * KT + ky * 3 + kx) * p.coutp + (co4 & 1) * 4) * 4u : kOOB;
    } else {
      const bool ok = v < WV && co0 + co4 * 4 < p.coutp;
      woff[q] = ok ? static_cast<unsigned>((ci * KT + tap) * p.coutp + co0 + co4 * 4) * 4u : kOOB;
    }
    wci[q] = ci;
    wl[q] = v < WV ? r * WP + co4 * 4 : KTW * NC * WP;
  }
  const __amdgpu_buffer_rsrc_t xr = ig_rsrc(x + static_cast<long long>(b) * p.in_bstride, p.in_bytes);   // signed: the batch stride may be the distance between two allocations
  const __amdgpu_buffer_rsrc_t wr = ig_rsrc(w, p.w_bytes);
  const unsigned cstride_b = static_cast<unsigned>(p.in_cstride) * 4u;
  const unsigned wstride_b = static_cast<unsigned>(KT * p.coutp) * 4u;

  // ---- per-lane fragment bases ----------------------------------------------------------------
  int boff[4];
#pragma unroll
  for (int pb = 0; pb < 4; ++pb) {
    if (MODE == MODE_D) boff[pb] = kq * chan_elems + wave * (TP / 4) + pb * 16 + j;
    else if (PR) {
      // output rows (base, base + DL) of the 8-row tile: DL 1 -> (2w, 2w+1); DL 2 -> (w&1) + 4(w>>1) + {0, 2}
      const int base = (DL == 1) ? wave * 2 : (wave & 1) + 4 * (wave >> 1);
      boff[pb] = kq * chan_elems + base * pitch + (pb & 1) * 16 + j;
    } else {
      // 8 x 32 tile: a wave owns two rows as 2 x 2 blocks of 16 pixels; 4 x 16 tile (TP == 64): one row, one block
      const int row = (TP == 64) ? wave : wave * 2 + (pb >> 1), col = (TP == 64) ? j : (pb & 1) * 16 + j;
      constexpr int st = (MODE == MODE_HW) ? ST : 1;
      boff[pb] = kq * chan_elems + row * st * pitch + col * st;
    }
  }
  const int aoff = kq * WP + j;

  v4f acc[CB][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = v4f{0.f, 0.f, 0.f, 0.f};
  // epilogue constants of this lane's channels, requested now so that their round trip is over long
  // before the epilogue (on the small layers it would otherwise sit on the critical path)
  float esc[CB][4], esh[CB][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = min(PR ? (kq & 1) * 4 + r : co0 + cb * 16 + kq * 4 + r, p.coutp - 1);     // clamped: unconditional loads, no branches
      esc[cb][r] = scale ? scale[co] : 1.f;              // null: raw convolution (training / backward-data)
      esh[cb][r] = shift ? shift[co] : 0.f;
    }

  // ---- register prefetch of one K chunk: branch-free buffer loads (zero padding, ragged channel
  // counts and partial tiles all resolve to out-of-range offsets or zero weights) ------------------
  // PF: two chunks in flight.  For grids that leave most of the chip idle (a few dozen workgroups: the small layers of the
  // coarse / fine levels) a short chunk's MFMA phase is far shorter than a global round trip, so with one chunk in flight the
  // kernel is a chain of exposed latencies; the second register set costs occupancy, which such a grid does not use anyway
  // (on full grids it loses: 1113 -> 1085 pairs/s with three passes in flight).
  constexpr int NSET = PF ? 2 : 1;
  constexpr bool VD = (MODE == MODE_D);                // 16-byte staging (see above)
  constexpr int NCQ = VD ? NC / 4 : 1;
  float rin[NSET][VD ? 1 : NC][VD ? 1 : RQ];
  constexpr bool VD64 = VD && TP == 64;
  u32x4 rin4[NSET][VD64 ? 1 : NCQ][(VD && !VD64) ? NTR : 1];
  u32x4 rin64[NSET][NI64];
  u32x4 rw[NSET][RWN];
  const int wave_u = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // uniform: the channel offsets stay scalar
  auto fetch = [&](auto set, int c0) {
    constexpr int S = decltype(set)::value;
    if constexpr (VD64) {
#pragma unroll
      for (int i = 0; i < NI64; ++i) {
        const unsigned chb = static_cast<unsigned>(min(c0 + ch64[i], p.Cin - 1)) * cstride_b;
        rin64[S][i] = __builtin_amdgcn_raw_buffer_load_b128(xr, go64[i] == kOOB ? kOOB : go64[i] + chb, 0, 0);
      }
    } else if constexpr (VD) {
#pragma unroll
      for (int ic = 0; ic < NCQ; ++ic) {
        const unsigned so = static_cast<unsigned>(min(c0 + wave_u + 4 * ic, p.Cin - 1)) * cstride_b;
#pragma unroll
        for (int t = 0; t < NTR; ++t) rin4[S][ic][t] = __builtin_amdgcn_raw_buffer_load_b128(xr, goff[t], so, 0);
      }
    } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      // channels past the slice re-read the last real one; their weights are zero
      const unsigned so = static_cast<unsigned>(min(c0 + c, p.Cin - 1)) * cstride_b;
#pragma unroll
      for (int i = 0; i < RQ; ++i) rin[S][c][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, goff[i], so, 0));
    }
    }
    const unsigned wso = static_cast<unsigned>(c0) * wstride_b;
#pragma unroll
    for (int q = 0; q < RWN; ++q)
      rw[S][q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (wci[q] < kend - c0) ? woff[q] + wso : kOOB, 0, 0);
  };
  auto commit = [&](auto set) {
    constexpr int S = decltype(set)::value;
    if constexpr (VD64) {
#pragma unroll
      for (int i = 0; i < NI64; ++i) *reinterpret_cast<u32x4*>(in_tile + lo64[i]) = rin64[S][i];
    } else if constexpr (VD) {
#pragma unroll
      for (int ic = 0; ic < NCQ; ++ic)
#pragma unroll
        for (int t = 0; t < NTR; ++t)
          *reinterpret_cast<u32x4*>(in_tile + (wave_u + 4 * ic) * chan_elems + loff[t]) = rin4[S][ic][t];
    } else {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int i = 0; i < RQ; ++i) in_tile[c * chan_elems + loff[i]] = rin[S][c][i];
    }
#pragma unroll
    for (int q = 0; q < RWN; ++q) *reinterpret_cast<u32x4*>(w_tile + wl[q]) = rw[S][q];
  };
  auto mma = [&]() {
#pragma unroll 1
  for (int c8 = 0; c8 < (NC >= 8 ? NC / 8 : 1); ++c8) {   // eight channels at a time (rolled: code size); NC == 4: one quad
    const float* it = in_tile + c8 * 8 * chan_elems;
    const float* wt0 = w_tile + c8 * 8 * WP + aoff;
    if (MODE == MODE_HW) {
      // 18 steps (tap, 4 channels); the fragments of step s+1 are read from LDS before the MFMAs of
      // step s issue, so the matrix pipe never waits on an LDS round trip
      constexpr int NS = (NC == 4) ? 9 : 2 * KTW;
      float a[2][CB], bv[2][4];
      auto frag = [&](int s, int slot) {
        const int tap = (NC == 4) ? s : (s >> 1), cq = (NC == 4) ? 0 : (s & 1);
        const int toff = (tap / 3) * DL * pitch + (tap % 3) * DL;          // PR: tap = (rho, kx), the same arithmetic
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) a[slot][cb] = wt0[(tap * NC + cq * 4) * WP + cb * 16];
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) bv[slot][pb] = it[boff[pb] + cq * 4 * chan_elems + toff];
      };
      frag(0, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) frag(s + 1, (s + 1) & 1);
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb)
            acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][cb], bv[s & 1][pb], acc[cb][pb], 0, 0, 0);
      }
    } else if (MODE == MODE_HWT) {
      // output parity (pa, pbit) selects which kernel taps land on input pixels:
      //   k3 s2 p1 (KT 9):  even -> (k=1, d=0);            odd -> (k=2, d=0), (k=0, d=+1)
      //   k4 s2 p1 (KT 16): even -> (k=1, d=0), (k=3, d=-1); odd -> (k=2, d=0), (k=0, d=+1)
      constexpr int KS = (KT == 16) ? 4 : 3, ORG = (KT == 16) ? 1 : 0;
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        int ky, dy;
        if (pa) { ky = ta ? 0 : 2; dy = ta ? 1 : 0; }
        else { if (ta && KT != 16) continue; ky = ta ? 3 : 1; dy = ta ? -1 : 0; }
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          int kx, dx;
          if (pbit) { kx = tb ? 0 : 2; dx = tb ? 1 : 0; }
          else { if (tb && KT != 16) continue; kx = tb ? 3 : 1; dx = tb ? -1 : 0; }
          const int toff = (dy + ORG) * pitch + dx + ORG;
          const float* wt = wt0 + ((ky * KS + kx) * NC) * WP;
#pragma unroll
          for (int cq = 0; cq < 2; ++cq) {
            float a[CB], bv[4];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) a[cb] = wt[cq * 4 * WP + cb * 16];
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) bv[pb] = it[boff[pb] + cq * 4 * chan_elems + toff];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
              for (int pb = 0; pb < NPB; ++pb)
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], bv[pb], acc[cb][pb], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        if (t >= ntaps) continue;
        const float* wt = wt0 + (plane_wt[t] * NC) * WP;
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) {
          float a[CB], bv[4];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) a[cb] = wt[cq * 4 * WP + cb * 16];
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) bv[pb] = it[boff[pb] + cq * 4 * chan_elems + t * TP];
#pragma unroll
          for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb)
              acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], bv[pb], acc[cb][pb], 0, 0, 0);
        }
      }
    }
  }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, NSET - 1>;
  fetch(S0{}, kbeg);
  if constexpr (NSET == 2) {
    if (kbeg + NC < kend) fetch(S1{}, kbeg + NC);
    for (int c0 = kbeg; c0 < kend; c0 += 2 * NC) {
      __syncthreads();                  // everyone is done reading the previous chunk
      commit(S0{});
      __syncthreads();
      if (c0 + 2 * NC < kend) fetch(S0{}, c0 + 2 * NC);
      mma();
      if (c0 + NC < kend) {
        __syncthreads();
        commit(S1{});
        __syncthreads();
        if (c0 + 3 * NC < kend) fetch(S1{}, c0 + 3 * NC);
        mma();
      }
    }
  } else {
    for (int c0 = kbeg; c0 < kend; c0 += NC) {
      __syncthreads();                  // everyone is done reading the previous chunk
      commit(S0{});
      __syncthreads();
      if (c0 + NC < kend) fetch(S0{}, c0 + NC);                 // next chunk in flight under the MFMAs below
      mma();
    }
  }

  // ---- epilogue: lane holds channels kq*4 + r of pixel j of each 16x16 tile.  Stores go through a
  // buffer descriptor: channels past Cout and pixels outside the image get an out-of-range offset and
  // are dropped by the hardware, so there is not a single branch around a memory operation here. ----
  const size_t hw_o = static_cast<size_t>(p.Ho) * p.Wo;
  const unsigned oplane = static_cast<unsigned>(p.Do) * static_cast<unsigned>(hw_o);          // one output channel
  const bool split = p.ksplit > 1;
  const __amdgpu_buffer_rsrc_t yr = split
      ? ig_rsrc(p.partial + (static_cast<size_t>(ks) * p.B + b) * p.Cout * oplane, p.part_bytes)
      : ig_rsrc(y + static_cast<long long>(b) * p.out_bstride, p.out_bytes);
  const unsigned ocs = split ? oplane * 4u : static_cast<unsigned>(p.out_cstride) * 4u;
  const float* ab = (MODE == MODE_HW && p.addend) ? p.addend + static_cast<size_t>(b) * p.add_bstride : nullptr;
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    unsigned opix, ppix = 0;          // element index inside a channel of y / inside one depth plane
    bool inside;
    if (PR) {
      // rows 0-7 of the accumulator tile (kq 0,1) = output row base + DL, rows 8-15 (kq 2,3) = output row base
      const int base = (DL == 1) ? wave * 2 : (wave & 1) + 4 * (wave >> 1);
      const int oy = ty0 + base + ((kq < 2) ? DL : 0), ox = tx0 + (pb & 1) * 16 + j;
      inside = oy < p.Ho && ox < p.Wo;
      ppix = static_cast<unsigned>(oy) * p.Wo + ox;
      opix = static_cast<unsigned>(od) * static_cast<unsigned>(hw_o) + ppix;
    } else if (MODE == MODE_D) {
      const unsigned px = px0 + wave * (TP / 4) + pb * 16 + j;
      inside = px < HW;
      opix = static_cast<unsigned>(od) * HW + px;
    } else {
      int oy = (TP == 64) ? ty0 + wave : ty0 + wave * 2 + (pb >> 1), ox = (TP == 64) ? tx0 + j : tx0 + (pb & 1) * 16 + j;
      if (MODE == MODE_HWT) {
        inside = oy < p.H && ox < p.W;
        oy = 2 * oy + pa; ox = 2 * ox + pbit;
        inside = inside && oy < p.Ho && ox < p.Wo;        // output_padding 0: the last row / column does not exist
      } else inside = oy < p.Ho && ox < p.Wo;
      ppix = static_cast<unsigned>(oy) * p.Wo + ox;
      opix = static_cast<unsigned>(od) * static_cast<unsigned>(hw_o) + ppix;
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = PR ? (kq & 1) * 4 + r : co0 + cb * 16 + kq * 4 + r;
        const unsigned off = (inside && co < p.Cout) ? opix * 4u + static_cast<unsigned>(co) * ocs : kOOB;   // per lane: VGPR
        float v = acc[cb][pb][r];
        if (!split) {
          if (MODE == MODE_HW && ab) v += ab[static_cast<size_t>(min(co, p.Cout - 1)) * p.add_cstride + static_cast<size_t>(od) * p.add_dstride + (inside ? ppix : 0u)];
          v = apply_act(v * esc[cb][r] + esh[cb][r], p.act, p.act_param, co);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, off, 0, 0);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// The same convolution with every fp32 product formed from bf16 pieces on the bf16 matrix pipe ("x6"):
//     a = a0 + a1 + a2 (+ O(2^-24 a)),  each part a bf16 (round-to-nearest of the running remainder)
//     a b ~= a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0          (dropped terms <= 2^-24 |a b|)
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The f32-input MFMA runs at 1/16 of the bf16 rate on gfx950
// (MI355X_MICROARCH.md), so six bf16 MFMAs of K=32 replace eight f32 MFMAs of K=4 at 3/8 of the matrix time.  With exact fp32
// accumulation the six-product form is as good as a plain fp32 dot product (tools/exp/bf16_split_eval.py: rms error 1.2e-7 of
// the output rms against 3.0e-7; the three-product split everybody quotes is 4.4e-6, which would not hold the parity bar); the
// matrix core aligns an instruction's 32 products and its accumulator to the largest addend, so a chunk's products are summed
// apart and added to the running sum in fp32 (see the K loop): measured max error 0.15e-6 ... 0.26e-6 of the output's magnitude
// for Cin 16 ... 512, the f32 kernel 0.3e-6 ... 0.7e-6 (tools/exp/x6_accuracy_sweep.py, tests/test_conv_x6_gpu.py).  Stride-1 (1,3,3) layers with >= 16 input channels.
//   K chunk = 16 input channels = two groups of 8; an MFMA's K = 32 is (two taps) x (16 channels): lane group kq holds
//   tap 2s + (kq >> 1), channel group kq & 1.  Nine taps = five steps, the tenth slot multiplies zero weights.
//   LDS: inputs pixel-major, [part][group][pixel] x 16 bytes (8 channels of one part): a B fragment is one ds_read_b128
//   and 16 consecutive pixels are 256 contiguous bytes; the fp32 tile is split while it is committed.  Weights arrive
//   pre-split from the host pass (weight_split6_kernel) as [chunk][part][tap slot][group][co][8]: an A fragment is one
//   ds_read_b128 as well.
// ------------------------------------------------------------------------------------------------


// Epilogue through LDS: the accumulator layout (lane = pixel j of a 16-pixel block, registers = 4 channels) stores 64-byte
// runs of four different channel planes per instruction -- 32 dword stores per thread for a 32-channel tile, ~11,000 cycles per
// workgroup.  Staged as [channel][256 tile pixels] (pitch 272: the four channel rows a wave writes per instruction sit on
// disjoint banks) a wave then stores 64 float4 of ONE channel: eight full 128-byte rows per instruction, a quarter of the
// instructions.  Needs whole quads (Wo % 4 == 0 for the 8 x 32 pixel tiles, H W % 4 == 0 for the 256-pixel runs of MODE_D).
constexpr int kEpiPitch = 272;
template <int NCB, int NPB = 4>       // NPB: 16-pixel blocks per wave (a wave's pixels are NPB * 16 consecutive tile pixels)
__device__ __forceinline__ void epilogue_stage(float* epi, const float (&v)[NCB][NPB][4], int wave, int kq, int j) {
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int r = 0; r < 4; ++r) epi[(cb * 16 + kq * 4 + r) * kEpiPitch + wave * (NPB * 16) + pb * 16 + j] = v[cb][pb][r];
}
// tile pixel quad q (0..63) of channel co: byte offset of its first pixel inside the channel plane, or kOOB
template <int NCB, int NT = 256, int QPC = 64, class QuadOffset>       // QPC: pixel quads per channel of the tile
__device__ __forceinline__ void epilogue_flush(const float* epi, __amdgpu_buffer_rsrc_t yr, unsigned ocs, int co_base, int Cout,
                                               QuadOffset quad_offset) {
#pragma unroll
  for (int it = 0; it < NCB * 16 * QPC / NT; ++it) {
    const int idx = static_cast<int>(threadIdx.x) + NT * it;
    const int col = idx / QPC, q = idx % QPC;
    const u32x4 val = *reinterpret_cast<const u32x4*>(epi + col * kEpiPitch + q * 4);
    const unsigned po = quad_offset(q);
    const int co = co_base + col;
    const unsigned off = (po != kOOB && co < Cout) ? po + static_cast<unsigned>(co) * ocs : kOOB;
    __builtin_amdgcn_raw_buffer_store_b128(val, yr, off, 0, 0);
  }
}

constexpr int X6_NC = 16, X6_SLOTS = 10;

// w_t fp32 [Cin][9][coutp] -> w6 [chunk][part 3][slot 10][group 2][coutp][8] bf16 (slot 9 and channels past Cin: zero)
__global__ void __launch_bounds__(256)
weight_split6_kernel(const float* __restrict__ w_t, u32x4* __restrict__ w6, int Cin, int coutp, int nchunk, int Cout, long long sci,
                     long long sco, long long st, int flip) {
  // Cout > 0: `w_t` is the FRAMEWORK's weight, element (ci, co, tap) at ci * sci + co * sco + tap * st (taps reversed when flip):
  // layout and split in one launch (ts_conv3d_hw_x6_weight_split_from)
  const int n = nchunk * X6_SLOTS * 2 * coutp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int co = i % coutp;
    int r = i / coutp;
    const int g = r & 1; r >>= 1;
    const int slot = r % X6_SLOTS, chunk = r / X6_SLOTS;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = chunk * X6_NC + g * 8 + e;
      if (Cout > 0) v[e] = (slot < 9 && ci < Cin && co < Cout) ? w_t[ci * sci + co * sco + (flip ? 8 - slot : slot) * st] : 0.f;
      else v[e] = (slot < 9 && ci < Cin) ? w_t[(static_cast<size_t>(ci) * 9 + slot) * coutp + co] : 0.f;
    }
    unsigned part[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split6(v[2 * e], v[2 * e + 1], part[0][e], part[1][e], part[2][e]);
#pragma unroll
    for (int pt = 0; pt < 3; ++pt)
      w6[(((static_cast<size_t>(chunk) * 3 + pt) * X6_SLOTS + slot) * 2 + g) * coutp + co] =
          u32x4{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
  }
}

// TR: output rows of the tile.  8 (a wave owns two rows: the form of full grids) or 4 (one row per wave, round 4): on grids under a round
// of workgroups the kernel's duration is ONE workgroup's serial chain (barrier - commit - barrier - fragment reads - MFMAs per chunk),
// and half a tile is half a chain at 3 workgroups per CU instead of 2 -- what the transposed x6s forms gained from the same cut
// (conv_x6s.hip).  Dilation 1 only (the staging deal below).
template <int CB, int DL, int TR = 8>
__global__ void __launch_bounds__(256, TR == 8 ? 2 : 3)
ig_conv_x6_kernel(const float* __restrict__ x, const u32x4* __restrict__ w6, const float* __restrict__ scale,
                  const float* __restrict__ shift, float* __restrict__ y, const IG p) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds6[];
  // Staged tile: rows ty0-DL .. ty0+7+DL, columns tx0-4 .. tx0+35 as ten ALIGNED quads per row (W % 4 == 0: a quad is inside
  // the image or outside it, never across its border).  One dwordx4 per (row, quad, channel): the first version staged single
  // pixels -- 32 dword gathers per thread and chunk, ~29 cycles of the texture addresser each, 60 % of the kernel's time.
  static_assert(TR == 8 || DL == 1, "4-row tiles: dilation 1");
  constexpr int NPB = TR / 2;                          // 16-pixel blocks per wave
  constexpr int SPC = TR == 8 ? 2 : 4;                  // staging threads per (row, quad): each takes 16 / SPC channels
  constexpr int in_rows = TR + 2 * DL, QPR = 10, LCOLS = 4 * QPR, NPIX = in_rows * LCOLS;
  // pitch of a (part, group) block: a multiple of 16 entries, so that the 8 + 8 lanes of one ds_read_b128 bank group (lanes of channel group 0
  // and of group 1, same pixels) land on disjoint banks; round 3's NPIX + 1 made EVERY fragment read a 2-way conflict (8 LDS cycles for 4:
  // SQ_LDS_BANK_CONFLICT 47 % of SQ_LDS_IDX_ACTIVE).  Conflict-free reads change nothing measurable (1276 vs 1275 pairs/s) -- the LDS is 37 %
  // busy either way -- and neither does pinning the issue order of reads and MFMAs with sched_group_barrier (1243: worse), DESIGN.md section 4.
  constexpr int NPIXP = NPIX;
  static_assert(NPIXP % 16 == 0, "block pitch");
  constexpr int SLOTS = in_rows * QPR;                  // (row, quad) pairs; threads [0, SLOTS) stage channels 0-7, [SLOTS, 2 SLOTS) 8-15
  static_assert(SPC * SLOTS <= 256, "staging slots");
  constexpr int COB = CB * 16;
  constexpr int WV = 3 * X6_SLOTS * 2 * COB;            // 16-byte weight vectors per chunk
  constexpr int RWN = (WV + 255) / 256;
  u32x4* in6 = lds6;                                    // [part][group][NPIXP]
  u32x4* w6s = lds6 + 3 * 2 * NPIXP;                    // [part][slot][group][COB] (+ dump)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  // XCD-aware placement: consecutive workgroup ids go round the eight XCDs (each with its own L2), so workgroup L works on
  // slot (L % 8) * (total / 8) + L / 8 of the (batch, plane, tile) order: an XCD gets a contiguous band of the image and its L2
  // serves the halo rows / columns that neighbouring tiles share (5-10 % on the 272x480 layers; TS_X6_XCD=0 = p.kspan & 1
  // switches it off for A/B runs).
  unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  {
    const unsigned total = gridDim.x * gridDim.y * gridDim.z, per = total / 8;
    if (p.xcd && lin < per * 8) lin = (lin % 8) * per + lin / 8;
  }
  const int tile = lin % gridDim.x;
  const int od = (lin / gridDim.x) % gridDim.y;
  const int bz = lin / (gridDim.x * gridDim.y);
  const int cog = bz % p.co_groups;
  // split-K (p.ksplit > 1): slice ks of the input channels, raw sums to the workspace, conv_splitk_finish adds the slices.  A grid of
  // a few dozen workgroups runs ONE workgroup per CU, i.e. without the co-resident workgroup whose matrix phase hides this one's
  // staging: every 16-channel chunk is then an exposed load -> split -> commit -> multiply sequence of 5-7 us.
  const int ks = (bz / p.co_groups) % p.ksplit, b = bz / (p.co_groups * p.ksplit);
  const int kbeg = ks * p.kspan, kend = min(p.Cin, kbeg + p.kspan);
  const int co0 = cog * COB;
  const int ty0 = (tile / p.tiles_x) * TR, tx0 = (tile % p.tiles_x) * 32;
  const unsigned HW = static_cast<unsigned>(p.H) * p.W;
  const unsigned cstride_b = static_cast<unsigned>(p.in_cstride) * 4u;

  const bool stager = threadIdx.x < SPC * SLOTS;
  const int spart = static_cast<int>(threadIdx.x) / SLOTS;                    // which 16 / SPC channels of the chunk this thread stages
  const int sg = TR == 8 ? spart : (spart >> 1), shalf = spart & 1;           // channel group; TR == 4: its first / second four channels
  const int sslot = static_cast<int>(threadIdx.x) - spart * SLOTS;
  const int srow = sslot / QPR, squad = sslot - srow * QPR;
  unsigned goff = kOOB;
  {
    const int gy = ty0 - DL + srow, gx = tx0 - 4 + 4 * squad;
    if (stager && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
      goff = (static_cast<unsigned>(od) * HW + static_cast<unsigned>(gy) * p.W + gx) * 4u;
  }
  const int lpix = (sg * NPIXP) + srow * LCOLS + 4 * squad;                   // + part * 2 * NPIXP + pixel in quad

  unsigned woff[RWN];
  int wl[RWN];
#pragma unroll
  for (int q = 0; q < RWN; ++q) {
    const int v = threadIdx.x + 256 * q;
    const int col = v % COB, r = v / COB;               // r = (part * 10 + slot) * 2 + group
    const bool ok = v < WV && co0 + col < p.coutp;
    woff[q] = ok ? static_cast<unsigned>(r * p.coutp + co0 + col) * 16u : kOOB;
    wl[q] = v < WV ? v : WV;
  }
  const __amdgpu_buffer_rsrc_t xr = ig_rsrc(x + static_cast<long long>(b) * p.in_bstride, p.in_bytes);
  const __amdgpu_buffer_rsrc_t wr = ig_rsrc(w6, p.w_bytes);
  const unsigned wchunk_b = static_cast<unsigned>(3 * X6_SLOTS * 2 * p.coutp) * 16u;

  // fragment bases (u32x4 units): B = pixel of this lane in each 16-pixel block, group kq & 1; A = channel j, group kq & 1
  int boff[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb)
    boff[pb] = (kq & 1) * NPIXP + (TR == 8 ? wave * 2 + (pb >> 1) : wave) * LCOLS + (4 - DL) + (pb & 1) * 16 + j;
  const int aoff = (kq & 1) * COB + j;
  const int tsel = kq >> 1;                             // which tap of a step's pair

  v4f acc[CB][NPB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) acc[cb][pb] = v4f{0.f, 0.f, 0.f, 0.f};
  float esc[CB][4], esh[CB][4];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = min(co0 + cb * 16 + kq * 4 + r, p.coutp - 1);
      esc[cb][r] = scale ? scale[co] : 1.f;
      esh[cb][r] = shift ? shift[co] : 0.f;
    }

  constexpr int NCH = 16 / SPC;                         // channels per staging thread
  v4f rin[NCH];
  u32x4 rw[RWN];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // channels past Cin re-read the last real one; their weights are zero
      const unsigned co = static_cast<unsigned>(min(c0 + spart * NCH + c, p.Cin - 1)) * cstride_b;
      rin[c] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, goff == kOOB ? kOOB : goff + co, 0, 0));
    }
    const unsigned wso = static_cast<unsigned>(c0 / X6_NC) * wchunk_b;
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
          if constexpr (TR == 8) in6[pt * 2 * NPIXP + lpix + i] = u32x4{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
          else reinterpret_cast<u32x2*>(in6 + pt * 2 * NPIXP + lpix + i)[shalf] = u32x2{part[pt][0], part[pt][1]};
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RWN; ++q) w6s[wl[q]] = rw[q];
  };

  fetch(kbeg);
  for (int c0 = kbeg; c0 < kend; c0 += X6_NC) {
    __syncthreads();
    commit();
    __syncthreads();
    if (c0 + X6_NC < kend) fetch(c0 + X6_NC);
    // Five steps of two half-steps (pixel blocks {0,1} and {2,3}).  The fragments of the NEXT half-step are read from LDS before
    // the 12 CB MFMAs of the current one issue (a wave otherwise alternates 18 ds_read_b128 and 48 MFMAs).  The kernel's duration
    // did not change with it -- what bounds it is the co-resident workgroups' staging, DESIGN.md section 4 -- but neither did the
    // register count, so the shorter dependent chain stays.
    // the chunk's products are summed in accumulators of their own and added to the running sum with an ordinary fp32 add: the
    // matrix core aligns the 32 products of an instruction to the largest addend, the running sum included, and drops what falls
    // below its last bit -- against a chunk-sized partial sum that costs far fewer bits than against the sum of all chunks
    v4f part[CB][NPB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) part[cb][pb] = v4f{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][3][CB], bv[2][3][2];
    auto load_a = [&](int s, int buf) {
      const int slot = 2 * s + tsel;
#pragma unroll
      for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
          a[buf][pt][cb] = __builtin_bit_cast(bf16x8, w6s[((pt * X6_SLOTS + slot) * 2) * COB + aoff + cb * 16]);
    };
    auto load_b = [&](int s, int h, int buf) {
      const int slot = 2 * s + tsel;
      const int tap = slot < 9 ? slot : 8;                                // slot 9: zero weights, any valid pixel
      const int toff = (tap / 3) * DL * LCOLS + (tap % 3) * DL;
#pragma unroll
      for (int pt = 0; pt < 3; ++pt)
#pragma unroll
        for (int k = 0; k < 2; ++k) bv[buf][pt][k] = __builtin_bit_cast(bf16x8, in6[pt * 2 * NPIXP + boff[2 * h + k] + toff]);
    };
    constexpr int NH = NPB / 2;                                           // half-steps per step (pixel-block pairs of a wave)
    load_a(0, 0);
    load_b(0, 0, 0);
#pragma unroll
    for (int u = 0; u < 5 * NH; ++u) {                                    // half-step u = NH s + h
      const int s = u / NH, h = u % NH;
      if (u + 1 < 5 * NH) {
        load_b((u + 1) / NH, (u + 1) % NH, (u + 1) & 1);
        if (h == NH - 1) load_a(s + 1, (s + 1) & 1);
      }
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            part[cb][2 * h + k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][PA[t]][cb], bv[u & 1][PB[t]][k], part[cb][2 * h + k], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) acc[cb][pb] += part[cb][pb];
  }

  const size_t hw_o = static_cast<size_t>(p.Ho) * p.Wo;
  const bool split = p.ksplit > 1;
  const unsigned oplane = static_cast<unsigned>(p.Do) * static_cast<unsigned>(hw_o);
  const __amdgpu_buffer_rsrc_t yr = split
      ? ig_rsrc(p.partial + (static_cast<size_t>(ks) * p.B + b) * p.Cout * oplane, p.part_bytes)
      : ig_rsrc(y + static_cast<long long>(b) * p.out_bstride, p.out_bytes);
  const unsigned ocs = split ? oplane * 4u : static_cast<unsigned>(p.out_cstride) * 4u;
  const float* ab = (p.addend && !split) ? p.addend + static_cast<size_t>(b) * p.add_bstride : nullptr;
  float outv[CB][NPB][4];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb) {
    const int oy = ty0 + (TR == 8 ? wave * 2 + (pb >> 1) : wave), ox = tx0 + (pb & 1) * 16 + j;
    const bool inside = oy < p.Ho && ox < p.Wo;
    const unsigned ppix = static_cast<unsigned>(oy) * p.Wo + ox;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + cb * 16 + kq * 4 + r;
        float v = acc[cb][pb][r];
        if (ab) v += ab[static_cast<size_t>(min(co, p.Cout - 1)) * p.add_cstride + static_cast<size_t>(od) * p.add_dstride + (inside ? ppix : 0u)];
        outv[cb][pb][r] = split ? v : apply_act(v * esc[cb][r] + esh[cb][r], p.act, p.act_param, co);
      }
  }
  __syncthreads();                                      // the last chunk's fragments are consumed: the tile buffers become the staging area
  float* epi = reinterpret_cast<float*>(lds6);
  epilogue_stage<CB, NPB>(epi, outv, wave, kq, j);
  __syncthreads();
  const unsigned obase = static_cast<unsigned>(od) * static_cast<unsigned>(hw_o);
  epilogue_flush<CB, 256, TR * 8>(epi, yr, ocs, co0, p.Cout, [&](int q) {
    const int oy = ty0 + (q >> 3), ox = tx0 + (q & 7) * 4;
    return (oy < p.Ho && ox < p.Wo) ? (obase + static_cast<unsigned>(oy) * p.Wo + ox) * 4u : kOOB;       // W % 4 == 0: whole quads
  });
}

// sums the split-K partials in a fixed order (deterministic) and applies scale / shift / activation.
// grid.y = (batch element, output channel): no 64-bit division per element
__global__ void __launch_bounds__(256)
conv_splitk_finish(const float* __restrict__ partial, const float* __restrict__ scale, const float* __restrict__ shift,
                   float* __restrict__ y, int B, int Cout, long long plane, int ksplit, int act, float act_param,
                   long long out_bstride, long long out_cstride, const float* __restrict__ addend, long long add_bstride,
                   long long hw_o) {
  const int bc = blockIdx.y;
  const int co = bc % Cout, b = bc / Cout;
  const long long n = static_cast<long long>(B) * Cout * plane;
  const float* pp = partial + static_cast<long long>(bc) * plane;
  float* yp = y + b * out_bstride + co * out_cstride;
  const float* ap = addend ? addend + b * add_bstride + co * hw_o : nullptr;
  const float sc = scale ? scale[co] : 1.f, sh = shift ? shift[co] : 0.f;
  const unsigned uplane = static_cast<unsigned>(plane), uhw = static_cast<unsigned>(hw_o);
  for (unsigned px = blockIdx.x * blockDim.x + threadIdx.x; px < uplane; px += gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int k = 0; k < ksplit; ++k) v += pp[k * n + px];
    if (ap) v += ap[px % uhw];
    yp[px] = apply_act(v * sc + sh, act, act_param, co);
  }
}

// Upper bound on the K chunk (8 | 16 | 32) for the launches that follow on this host thread, see
// ts_conv_set_chunk_cap.
thread_local int g_chunk_cap = 32;
// grids below this many workgroups take the two-chunks-in-flight form of the NC = 8 kernel (TS_CONV_PF_MAX_WGS, 0 = never)
// Environment switches are read by NAMED functions: hipcc 7.2 resolved a third namespace-scope `= [] { ... }()` initialiser of
// this file to the FIRST such lambda (a bool came out holding 192 -- the default of g_pf_max_wgs -- and tested false).
const long long g_pf_max_wgs = env_ll("TS_CONV_PF_MAX_WGS", 512);     // measured again with the 64-pixel tiles in place (their grids count four-fold): 256 / 512 / 1024 = 1191 / 1189 / 1176 pairs/s, one pass at a time 846 / 846 / 841
// grids of at most this many 8 x 32 workgroups run the plain (1,3,3) forms on 4 x 16 pixel tiles (launch_ig); measured 0 / 64 / 128 / 256 / 512: 1163 / 1170 / 1174 / 1181 / 1181 pairs/s, one pass at a time 808 / 816 / 823 / 833 / 835
const long long g_small_hw = env_ll("TS_CONV_HW_SMALL_WGS", 256);
// small grids of Cout <= 8 layers with 64+ input channels: 4 x 16 tiles (plain form, half of each MFMA idle, no split-K) instead of the
// row-paired 8 x 32 form -- 64 -> 2 on 14 x 34 x 60 (the coarse prediction heads) 14.1 -> 10.1 us, 128 -> 8 on 136 x 240 20.1 -> 16.5;
// 32 -> 2 on 7 x 68 x 120 would lose (9.4 -> 10.4)
const bool g_small_over_pairing = env_ll("TS_CONV_SMALL_OVER_PAIRING", 1) != 0;
// TS_CONV_ROW_PAIRING=0 switches the Cout <= 8 row pairing off (A/B measurements)
const bool g_row_pairing = env_not_zero("TS_CONV_ROW_PAIRING");

template <int CB, int MODE, int KT, int ST, int DL, int NC, int PR = 0, int PF = 0, int TP = 256>
int launch_one(const float* x, const float* w, const float* scale, const float* shift, float* y, const IG& p,
               dim3 grid, hipStream_t st) {
  using G = Geom<MODE, KT, ST, DL, TP>;
  constexpr int WP = (CB * 16) | 16;
  constexpr int KTW = PR ? 12 : KT;
  constexpr size_t lds = (static_cast<size_t>(NC) * G::chan_elems + static_cast<size_t>(KTW) * NC * WP + 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "ig_conv_kernel: tile does not fit the LDS");
  auto kern = &ig_conv_kernel<CB, MODE, KT, ST, DL, NC, PR, PF, TP>;
  if (lds > 64 * 1024) {
    static bool raised = false;      // per instantiation
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      raised = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, w, scale, shift, y, p);
  return ts::launched("ig_conv_kernel");
}

template <int CB, int MODE, int KT, int ST, int DL, int TP = 256>
int launch_nc(long long wgs, const float* x, const float* w, const float* scale, const float* shift, float* y, const IG& p,
              dim3 grid, hipStream_t st) {
  // K-chunk size.  A workgroup pays one global-memory round trip per chunk, so the chunk is made as
  // long as possible -- but only while every workgroup of the grid stays resident (LDS is what limits
  // that): co-resident workgroups hide each other's round trips, queued ones do not.
  using G = Geom<MODE, KT, ST, DL, TP>;
  constexpr int WP = (CB * 16) | 16;
  constexpr size_t per_ch = (static_cast<size_t>(G::chan_elems) + static_cast<size_t>(KT) * WP) * sizeof(float);
  constexpr size_t lds_cu = 160 * 1024;
  const size_t per_cu = static_cast<size_t>((wgs + ts::kNumCU - 1) / ts::kNumCU);
  if constexpr (MODE == MODE_HW && TP == 256) {
    // 3-channel image layers: a 4-channel chunk (one k = 4 MFMA step per tap) instead of 8 with five zero channels
    if (p.Cin <= 4 && p.ksplit == 1) return launch_one<CB, MODE, KT, ST, DL, 4>(x, w, scale, shift, y, p, grid, st);
  }
  const int max_nc = g_chunk_cap;
  if constexpr (MODE == MODE_HW && KT == 9 && ST == 1 && CB == 1 && TP == 256) {
    if (p.Cout <= 8 && g_row_pairing) {           // both output rows of a wave in one 16-row MFMA (see ig_conv_kernel, PR)
      constexpr size_t per_ch2 = (static_cast<size_t>(G::chan_elems) + 12 * WP) * sizeof(float);
      auto fits2 = [&](int nc) { return nc <= max_nc && (nc * per_ch2 + 16) * per_cu <= lds_cu && p.kspan >= nc; };
      if (fits2(32)) return launch_one<CB, MODE, KT, ST, DL, 32, 1>(x, w, scale, shift, y, p, grid, st);
      if (fits2(16)) return launch_one<CB, MODE, KT, ST, DL, 16, 1>(x, w, scale, shift, y, p, grid, st);
      return launch_one<CB, MODE, KT, ST, DL, 8, 1>(x, w, scale, shift, y, p, grid, st);
    }
  }
  auto fits = [&](int nc) { return nc <= max_nc && (nc * per_ch + 16) * per_cu <= lds_cu && p.kspan >= nc; };
  if constexpr (32 * per_ch + 16 <= lds_cu) { if (fits(32)) return launch_one<CB, MODE, KT, ST, DL, 32, 0, 0, TP>(x, w, scale, shift, y, p, grid, st); }
  if constexpr (16 * per_ch + 16 <= lds_cu) { if (fits(16)) return launch_one<CB, MODE, KT, ST, DL, 16, 0, 0, TP>(x, w, scale, shift, y, p, grid, st); }
  if constexpr (CB <= 2) {
    if (wgs < g_pf_max_wgs && p.kspan > 8) return launch_one<CB, MODE, KT, ST, DL, 8, 0, 1, TP>(x, w, scale, shift, y, p, grid, st);
  }
  return launch_one<CB, MODE, KT, ST, DL, 8, 0, 0, TP>(x, w, scale, shift, y, p, grid, st);
}

template <int MODE, int KT, int ST, int DL>
int launch_ig(const float* x, const float* w, const float* scale, const float* shift, float* y, IG p, int B,
              int grid_x, int grid_y, hipStream_t st) {
  {   // output extents behind the epilogue's buffer descriptors (32-bit offsets per batch element)
    const unsigned long long plane = static_cast<unsigned long long>(p.Do) * p.Ho * p.Wo;
    const unsigned long long out_b = (static_cast<unsigned long long>(p.Cout - 1) * p.out_cstride + plane) * 4ull;
    const unsigned long long part_b = static_cast<unsigned long long>(p.Cout) * plane * 4ull;
    TS_REQUIRE(out_b < 0x7fffffffull && part_b < 0x7fffffffull && p.out_cstride >= 0, TS_ERR_UNSUPPORTED,
               "conv: a batch element of y spans 2 GiB or more");
    p.out_bytes = static_cast<unsigned>(out_b); p.part_bytes = static_cast<unsigned>(part_b);
  }
  // widest channel block that still leaves enough workgroups to fill the chip
  const int need = (p.Cout + 15) / 16;                 // 16-channel blocks
  int cb = need >= 4 ? 4 : (need >= 2 ? 2 : 1);
  auto groups = [&](int c) { return (need + c - 1) / c; };
  const long long tiles = static_cast<long long>(grid_x) * grid_y * B * p.ksplit;
  while (cb > 1 && tiles * groups(cb) < ts::kNumCU + ts::kNumCU / 2) cb >>= 1;
  p.co_groups = groups(cb);
  p.B = B;
  static const int xcd = env_not_zero("TS_CONV_XCD") ? 1 : 0;
  p.xcd = xcd;
  const long long wgs = tiles * p.co_groups;
  if constexpr (MODE == MODE_D) {
    // few dozen workgroups: 64-pixel tiles (Geom, TP) -- four times the workgroups, a quarter of the K loop's matrix and staging work each
    static const long long small = env_ll("TS_CONV_D_SMALL_WGS", 256);   // measured 0 / 64 / 128 / 256 / 512 / 1024: 1129 / 1156 / 1165 / 1165 / 1162 / 1153 pairs/s, one pass at a time 768 / 795 / 798 / 812 / 811 / 809
    if (wgs <= small && cb == 1) {
      const int gx64 = (p.H * p.W + 63) / 64;
      const dim3 grid64(gx64, grid_y, B * p.co_groups * p.ksplit);
      return launch_nc<1, MODE, KT, ST, DL, 64>(4 * wgs, x, w, scale, shift, y, p, grid64, st);
    }
  }
  if constexpr (MODE == MODE_HW) {
    // the same for plain (1,3,3) layers: 4 x 16 pixel tiles instead of 8 x 32 (not the row-paired Cout <= 8 form, not the 3-channel image layer)
    if (wgs <= g_small_hw && cb == 1 && p.Cin > 4 && !(p.Cout <= 8 && ST == 1 && g_row_pairing && !(g_small_over_pairing && p.Cin >= 64))) {
      p.tiles_x = (p.Wo + 15) / 16;
      const int gx64 = ((p.Ho + 3) / 4) * p.tiles_x;
      const dim3 grid64(gx64, grid_y, B * p.co_groups * p.ksplit);
      return launch_nc<1, MODE, KT, ST, DL, 64>(4 * wgs, x, w, scale, shift, y, p, grid64, st);
    }
  }
  if constexpr (MODE == MODE_HWT) {
    if (wgs <= g_small_hw && cb == 1) {                   // transposed form: 4 x 16 INPUT pixels per tile, four parity classes each
      p.tiles_x = (p.W + 15) / 16;
      const int gx64 = ((p.H + 3) / 4) * p.tiles_x * 4;
      const dim3 grid64(gx64, grid_y, B * p.co_groups * p.ksplit);
      return launch_nc<1, MODE, KT, ST, DL, 64>(4 * wgs, x, w, scale, shift, y, p, grid64, st);
    }
  }
  const dim3 grid(grid_x, grid_y, B * p.co_groups * p.ksplit);
  if (cb == 4) return launch_nc<4, MODE, KT, ST, DL>(wgs, x, w, scale, shift, y, p, grid, st);
  if (cb == 2) return launch_nc<2, MODE, KT, ST, DL>(wgs, x, w, scale, shift, y, p, grid, st);
  return launch_nc<1, MODE, KT, ST, DL>(wgs, x, w, scale, shift, y, p, grid, st);
}

// Weight re-layout for the convolution kernels in ONE launch (the training path re-lays every layer's weight three times per
// step -- forward, backward-data, and the framework needed reshape + permute + zeros + copy = four launches for each):
//   out[a][t][b] = b < nb ? w[a * sa + b * sb + (flip ? T - 1 - t : t) * st] : 0       out is [A][T][bpad]
__global__ void __launch_bounds__(256)
weight_layout_kernel(const float* __restrict__ w, float* __restrict__ out, int A, int T, int nb, int bpad, long long sa, long long sb,
                     long long st, int flip) {
  // (32-bit index arithmetic: the host refuses arrays of 2^31 elements or more; the 64-bit divisions this loop used to do per element
  // were most of the instructions of a launch that runs ~175 times per training step)
  const unsigned n = static_cast<unsigned>(A) * static_cast<unsigned>(T) * static_cast<unsigned>(bpad);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned r = i / static_cast<unsigned>(bpad);
    const int b = static_cast<int>(i - r * static_cast<unsigned>(bpad));
    const unsigned a = r / static_cast<unsigned>(T);
    const int t = static_cast<int>(r - a * static_cast<unsigned>(T));
    out[i] = b < nb ? w[static_cast<long long>(a) * sa + b * sb + (flip ? T - 1 - t : t) * st] : 0.f;
  }
}

// The same for a whole table of weights in ONE launch (a training step re-lays every convolution weight once per
// optimizer update: ~280 launches of the kernel above otherwise).  blockIdx.y = table entry.
struct WeightLayoutDesc {            // == ts_weight_layout_desc of include/ts_hip.h (64 bytes)
  const float* w; float* out;
  int A, T, nb, bpad;
  long long sa, sb, st;
  int flip, reserved;
};
static_assert(sizeof(WeightLayoutDesc) == 64, "table entry layout");

__global__ void __launch_bounds__(256)
weight_layout_many_kernel(const WeightLayoutDesc* __restrict__ table) {
  const WeightLayoutDesc d = table[blockIdx.y];
  const long long n = static_cast<long long>(d.A) * d.T * d.bpad;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i % d.bpad);
    const long long r = i / d.bpad;
    const int t = static_cast<int>(r % d.T), a = static_cast<int>(r / d.T);
    d.out[i] = b < d.nb ? d.w[a * d.sa + b * d.sb + (d.flip ? d.T - 1 - t : t) * d.st] : 0.f;
  }
}

// The general form (ABI 9): an entry writes `nb` columns starting at column `col0` of a destination whose row (a) and tap (t)
// pitches are given -- never the padding around them.  With it every derived weight array of the inference engine is a table
// entry over the framework's own parameter: a concatenation of two layers along Cout (two entries, col0 = 0 and Cout_a), the
// block-diagonal pair of the prediction heads, the 1x1 "Q" weights of the warp-commuted first layer (tap pitch = Cout) -- and
// the whole engine re-folds after an optimizer step in ONE launch (aggregation/native.py, Tape).
struct WeightLayoutDesc2 {           // == ts_weight_layout_desc2 of include/ts_hip.h (80 bytes)
  const float* w; float* out;
  int A, T, nb, col0;
  long long sa, sb, st;
  long long out_sa, out_st;
  int flip, reserved;
};
static_assert(sizeof(WeightLayoutDesc2) == 80, "table entry layout");

__global__ void __launch_bounds__(256)
weight_layout_many2_kernel(const WeightLayoutDesc2* __restrict__ table) {
  const WeightLayoutDesc2 d = table[blockIdx.y];
  const long long n = static_cast<long long>(d.A) * d.T * d.nb;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int b = static_cast<int>(i % d.nb);
    const long long r = i / d.nb;
    const int t = static_cast<int>(r % d.T), a = static_cast<int>(r / d.T);
    d.out[a * d.out_sa + t * d.out_st + d.col0 + b] = d.w[a * d.sa + b * d.sb + (d.flip ? d.T - 1 - t : t) * d.st];
  }
}

// extent checks shared by the entry points: buffer addressing is 32-bit per batch element
bool ig_extent(IG& p, int KT) {
  const unsigned long long in_b = (static_cast<unsigned long long>(p.Cin - 1) * p.in_cstride + static_cast<unsigned long long>(p.D) * p.H * p.W) * 4ull;
  const unsigned long long w_b = static_cast<unsigned long long>(p.Cin) * KT * p.coutp * 4ull;
  if (in_b >= 0x7fffffffull || w_b >= 0x7fffffffull || p.in_cstride < 0) return false;
  p.in_bytes = static_cast<unsigned>(in_b); p.w_bytes = static_cast<unsigned>(w_b);
  p.out_bytes = p.part_bytes = 0;             // set by ig_out_extent once the output geometry is known
  return true;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the two convolution families on the matrix cores:
//     dW[co][ci][tap] = sum_{b, d, px} dY[b][co][d][px] * X[b][ci][d'][px shifted by tap]
// i.e. per tap a GEMM whose reduction runs over the pixels: A = dY (16 channels x 4 pixels),
// B = X (4 pixels x 16 channels), D = a 16x16 block of dW.
//   workgroup = (a run of pixel tiles) x (16 input channels) x (all <= 64 output channels, all taps);
//   the (tap, 16-output-channel block) pairs are dealt round-robin to the four waves, each keeping its
//   blocks of dW in registers across all of the workgroup's tiles and writing them ONCE at the end as a
//   partial [ci block][workgroup][item][r][lane] of the caller's workspace; wgrad_finish sums the
//   workgroups' partials in a fixed order (deterministic; the first version added them to dW with fp32
//   atomics -- 5-9 million device-scope atomics per layer, 270-365 us of the 2-D layers' 300-380 us).
//   Per tile: dY [pixels][channels] and X [haloed pixels][16 channels] are staged in LDS pixel-major
//   (odd pitches: the transposing writes and both fragment reads are conflict-free or 2-way at worst).
// ------------------------------------------------------------------------------------------------
struct WG {
  int B, Cin, Cout, D, H, W, Do, Ho, Wo;
  int stride, dil, pad, k;
  int tiles_x, tiles_per_plane, ntiles;       // ntiles = B * Do * tiles_per_plane
  int cop;                                    // LDS pitch of a dY pixel row (== 17 mod 32)
  int vec;                                    // MODE_HW: 16-byte staging of X (W % 4 == 0, aligned planes)
  long long x_bstride, x_cstride, dy_bstride, dy_cstride;
  unsigned x_bytes, dy_bytes;
};

template <int MODE, int KT, int ST, int DL>
__global__ void __launch_bounds__(256)
wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, const WG p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CIP = 17;                                        // LDS pitch of an X pixel (16 channels + 1)
  constexpr int TRW = (MODE == MODE_HW) ? (ST == 2 ? 4 : 8) : 8; // tile rows (MODE_HW) -- 32 columns; MODE_D: 256 pixels
  constexpr int NPX = (MODE == MODE_HW) ? TRW * 32 : 256;        // output pixels per tile
  constexpr int in_rows = (MODE == MODE_HW) ? (TRW - 1) * ST + 2 * DL + 1 : 1;
  constexpr int in_cols = (MODE == MODE_HW) ? 31 * ST + 2 * DL + 1 : 256;
  constexpr int NIN = (MODE == MODE_HW) ? in_rows * in_cols : KT * 256;      // staged input pixels per channel
  constexpr int RQ = (NIN + 255) / 256;
  float* xs = lds;                                               // [NIN][CIP]
  float* dys = lds + NIN * CIP;                                  // [NPX][cop]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 15, kq = lane >> 4;
  const int ci0 = blockIdx.y * 16;
  const int cob = (p.Cout + 15) / 16, nitems = KT * cob;
  const unsigned HW = static_cast<unsigned>(p.H) * p.W, HWo = static_cast<unsigned>(p.Ho) * p.Wo;

  // Work split (round 3): a wave owns ONE 16-output-channel block and ALL taps of it, over a share of the tile's pixels -- 1, 2 or 4
  // blocks across the waves, the pixels split 4, 2 or 1 ways.  Per 4-pixel step that is one dY fragment and KT input fragments for
  // KT MFMAs into KT independent accumulators: (1 + KT) / KT LDS reads per MFMA and no dependent MFMA chain (the first version dealt
  // (tap, block) pairs to the waves: two LDS reads per MFMA, every MFMA of a pair waiting for the one before it).  Worth 4 % on the
  // largest layer (184 -> 177 us): what bounds this kernel is its staging -- 136-byte runs of a 34-pixel haloed tile row per channel
  // plane, i.e. half-used cache lines on 264 MB of reads at the 1/4 level -- not the matrix phase; prefetching the next tile into
  // registers under the MFMAs made it SLOWER (227 us: 143 VGPRs, one workgroup less per CU).
  const int cbsplit = cob >= 3 ? 4 : cob, pxsplit = 4 / cbsplit;
  const int cb = wave % cbsplit, part_px = wave / cbsplit;
  const bool working = cb < cob;
  v4f acc[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int sp = tile % p.tiles_per_plane;
    const int od = (tile / p.tiles_per_plane) % p.Do, b = tile / (p.tiles_per_plane * p.Do);
    const __amdgpu_buffer_rsrc_t xr = ig_rsrc(x + static_cast<size_t>(b) * p.x_bstride, p.x_bytes);
    const __amdgpu_buffer_rsrc_t yr = ig_rsrc(dy + static_cast<size_t>(b) * p.dy_bstride, p.dy_bytes);
    int ty0 = 0, tx0 = 0;
    if (MODE == MODE_HW) { ty0 = (sp / p.tiles_x) * TRW; tx0 = (sp % p.tiles_x) * 32; }
    __syncthreads();                                             // the previous tile's fragments are consumed
    // ---- stage X: 16 channels of the haloed input tile (zero padding / ragged channels via out-of-range offsets)
    // Vector form (round 5; W % 4 == 0, 16-byte aligned planes): a tile row as ALIGNED quads from column tx0 ST - 4 on -- one
    // dwordx4 per (row, quad, channel) instead of four dword gathers, whole cache lines instead of 136-byte runs (this staging, not
    // the matrix phase, is what the kernel's duration follows: the first layers' 50-198 MB inputs streamed at 0.6-0.8 TB/s).
    if constexpr (MODE == MODE_HW) {
      if (p.vec) {
        constexpr int QPR = (in_cols + 4 - DL + 3) / 4;                 // quads per tile row
        constexpr int SL = in_rows * QPR;                              // (row, quad) slots of a channel plane
        constexpr int CPT = (SL <= 128) ? 8 : 16;                      // channels per thread: two threads share a slot where they fit
        const int slot = (SL <= 128) ? static_cast<int>(threadIdx.x & 127) : static_cast<int>(threadIdx.x);
        const int cfirst = (SL <= 128) ? static_cast<int>(threadIdx.x >> 7) * 8 : 0;
        if (slot < SL) {
          const int cy = slot / QPR, qx = slot - cy * QPR;
          const int gy = ty0 * ST - DL + cy, gx4 = tx0 * ST - 4 + 4 * qx;
          const bool rowok = gy >= 0 && gy < p.H && gx4 >= 0 && gx4 < p.W;      // W % 4 == 0: a quad is inside the image or outside it
          const unsigned off = rowok ? (static_cast<unsigned>(od) * HW + static_cast<unsigned>(gy) * p.W + gx4) * 4u : kOOB;
          u32x4 v[CPT];
#pragma unroll
          for (int c = 0; c < CPT; ++c) {
            const int ch = ci0 + cfirst + c;
            const unsigned so = static_cast<unsigned>(min(ch, p.Cin - 1)) * static_cast<unsigned>(p.x_cstride) * 4u;
            v[c] = __builtin_amdgcn_raw_buffer_load_b128(xr, (ch < p.Cin) ? off : kOOB, so, 0);
          }
          const int cx0 = 4 * qx - (4 - DL);                             // tile column of the quad's first element
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cx = cx0 + e;
            if (cx >= 0 && cx < in_cols) {
#pragma unroll
              for (int c = 0; c < CPT; ++c) xs[(cy * in_cols + cx) * CIP + cfirst + c] = __uint_as_float(v[c][e]);
            }
          }
        }
      }
    }
    if (MODE != MODE_HW || !p.vec) {
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const int i = threadIdx.x + 256 * q;
      unsigned off = kOOB;
      if (i < NIN) {
        if (MODE == MODE_HW) {
          const int cy = i / in_cols, cx = i - cy * in_cols;
          const int gy = ty0 * ST - DL + cy, gx = tx0 * ST - DL + cx;
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) off = (static_cast<unsigned>(od) * HW + static_cast<unsigned>(gy) * p.W + gx) * 4u;
        } else {
          const int t = i >> 8, px = sp * 256 + (i & 255);
          const int id = od * p.stride + t * p.dil - p.pad;
          if (id >= 0 && id < p.D && px < static_cast<int>(HW)) off = (static_cast<unsigned>(id) * HW + px) * 4u;
        }
      }
      float v[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const unsigned so = static_cast<unsigned>(min(ci0 + c, p.Cin - 1)) * static_cast<unsigned>(p.x_cstride) * 4u;
        v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (ci0 + c < p.Cin) ? off : kOOB, so, 0));
      }
      if (i < NIN) {
#pragma unroll
        for (int c = 0; c < 16; ++c) xs[i * CIP + c] = v[c];
      }
    }
    }
    // ---- stage dY: every output channel of the tile's pixels
    for (int i = threadIdx.x; i < NPX; i += 256) {
      unsigned off = kOOB;
      if (MODE == MODE_HW) {
        const int oy = ty0 + i / 32, ox = tx0 + (i & 31);
        if (oy < p.Ho && ox < p.Wo) off = (static_cast<unsigned>(od) * HWo + static_cast<unsigned>(oy) * p.Wo + ox) * 4u;
      } else {
        const int px = sp * 256 + i;
        if (px < static_cast<int>(HWo)) off = (static_cast<unsigned>(od) * HWo + px) * 4u;
      }
      for (int c0 = 0; c0 < cob * 16; c0 += 16) {
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const unsigned so = static_cast<unsigned>(min(c0 + c, p.Cout - 1)) * static_cast<unsigned>(p.dy_cstride) * 4u;
          v[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (c0 + c < p.Cout) ? off : kOOB, so, 0));
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) dys[i * p.cop + c0 + c] = v[c];
      }
    }
    __syncthreads();
    // ---- this wave's channel block: all taps over its share of the tile's pixels, four pixels per MFMA
    if (working) {
      const int nstep = (NPX / 4) / pxsplit, s0 = part_px * nstep;
      const float* ap = dys + kq * p.cop + cb * 16 + j;
      const float* bp = xs + (MODE == MODE_HW ? kq * ST : kq) * CIP + j;
#pragma unroll 2
      for (int step = s0; step < s0 + nstep; ++step) {
        int xo;
        if (MODE == MODE_HW) xo = ((step >> 3) * ST * in_cols + (step & 7) * 4 * ST) * CIP;
        else xo = step * 4 * CIP;
        const float a = ap[step * 4 * p.cop];
        float bv[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const int toff = (MODE == MODE_HW) ? ((t / 3) * DL * in_cols + (t % 3) * DL) : t * 256;
          bv[t] = bp[xo + toff * CIP];
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[t], acc[t], 0, 0, 0);
      }
    }
  }
  // ---- the pixel shares of a channel block meet in LDS (fixed order: share 0 + 1 (+ 2 + 3)), then the block leaves as this
  // workgroup's partial: row (item = tap * cob + cb, r) of 64 lanes; lane holds co = 16 cb + 4 kq + r, ci = ci0 + j
  if (pxsplit > 1) {
    __syncthreads();                                              // the last tile's fragments are consumed: LDS is free
    float* red = lds;                                             // [wave][KT][4][64]
    if (working && part_px > 0) {
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * KT + t) * 4 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (working && part_px == 0) {
      for (int q = 1; q < pxsplit; ++q) {
        const int w2 = cb + q * cbsplit;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] += red[((w2 * KT + t) * 4 + r) * 64 + lane];
      }
    }
  }
  if (working && part_px == 0) {
    float* mine = part + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * nitems * 256;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int item = t * cob + cb;
#pragma unroll
      for (int r = 0; r < 4; ++r) mine[(item * 4 + r) * 64 + lane] = acc[t][r];
    }
  }
}

// dW[co][ci][tap] = sum over the workgroups of one input-channel block, in launch order.
//   grid (rows = items x 4, ci blocks), 256 threads: wave w adds workgroups w, w+4, ... of its row, the four
//   wave sums are combined in wave order.
__global__ void __launch_bounds__(256)
wgrad_finish(const float* __restrict__ part, float* __restrict__ dw, int gx, int nitems, int cob, int Cin, int Cout, int KT) {
  __shared__ float red[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x, cib = blockIdx.y;
  const float* src = part + (static_cast<size_t>(cib) * gx * nitems * 4 + row) * 64 + lane;
  const size_t pitch = static_cast<size_t>(nitems) * 256;
  float s = 0.f;
#pragma unroll 8
  for (int g = wave; g < gx; g += 4) s += src[g * pitch];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int item = row >> 2, r = row & 3;
    const int tap = item / cob, cb = item - tap * cob;
    const int co = cb * 16 + (lane >> 4) * 4 + r, ci = cib * 16 + (lane & 15);
    if (co < Cout && ci < Cin) dw[(static_cast<size_t>(co) * Cin + ci) * KT + tap] = t;
  }
}

// ---- deferred finish (round 5) -------------------------------------------------------------------------------------------------
// A training step's backward runs ~95 weight-gradient kernels, each followed by its own wgrad_finish launch of a few microseconds.
// Nothing reads a weight gradient before the optimizer, so inside ts_conv_wgrad_defer(1) ... (0) the finish launches are not issued:
// their descriptors are kept and ONE wgrad_finish_many launch sums every layer's partials at the end of backward
// (ts_conv_wgrad_take -> table -> ts_conv_wgrad_finish_many).  Same sums in the same order as wgrad_finish.
struct WgradFinishDesc {             // == ts_wgrad_finish_desc of include/ts_hip.h (48 bytes)
  const float* part; float* dw;
  int gx, nitems, cob, Cin, Cout, KT;
  int ciblocks, block0;              // block0: first workgroup of this entry in the flat grid (rows x ciblocks workgroups per entry)
};
static_assert(sizeof(WgradFinishDesc) == 48, "table entry layout");
thread_local bool g_wgrad_defer = false;      // set around single calls by the thread that issues them (autograd's backward thread)
std::mutex g_wgrad_mutex;                     // the kept descriptors are the process's: backward runs on autograd's thread, the
std::vector<WgradFinishDesc> g_wgrad_pending; // step that collects them on the caller's

__global__ void __launch_bounds__(256)
wgrad_finish_many(const WgradFinishDesc* __restrict__ table, int n) {
  __shared__ float red[4][64];
  // the entry of this workgroup: last one with block0 <= blockIdx.x (n <= a few hundred, the table sits in L2)
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block0 <= static_cast<int>(blockIdx.x)) lo = mid; else hi = mid - 1;
  }
  const WgradFinishDesc d = table[lo];
  const int local = static_cast<int>(blockIdx.x) - d.block0;
  const int rows = d.nitems * 4;
  const int row = local % rows, cib = local / rows;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* src = d.part + (static_cast<size_t>(cib) * d.gx * d.nitems * 4 + row) * 64 + lane;
  const size_t pitch = static_cast<size_t>(d.nitems) * 256;
  float s = 0.f;
#pragma unroll 8
  for (int g = wave; g < d.gx; g += 4) s += src[g * pitch];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0) {
    const float t = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    const int item = row >> 2, r = row & 3;
    const int tap = item / d.cob, cb = item - tap * d.cob;
    const int co = cb * 16 + (lane >> 4) * 4 + r, ci = cib * 16 + (lane & 15);
    if (co < d.Cout && ci < d.Cin) d.dw[(static_cast<size_t>(co) * d.Cin + ci) * d.KT + tap] = t;
  }
}

// workgroups per input-channel block: the chip filled about twice over all blocks (every workgroup ends
// with one partial of `items` KiB, so more of them only lengthens wgrad_finish)
// Round 5: as many workgroups as are RESIDENT at once -- floor(160 KB / LDS footprint) per CU, at most 4 -- and never one more: the
// round-3 rule (ceil(2 x 256 / blocks)) gave the coarse first layer 24 x 22 = 528 workgroups of 73 KB, i.e. one full round of 512 and
// a tail of 16 (120 us; 23 x 22 = 506: 94 us), and the Cout <= 16 first layers of the sampled levels (40 KB: three fit) two per CU
// (304 -> 8 on 5 x 136 x 240: 262 -> 214 us with three).  Layers whose workgroups fit once per CU keep two rounds.
// TS_WGRAD_GROUPS_PER_CU > 0 forces the old rule with that many per CU.
int wgrad_groups(int ciblocks, size_t lds_bytes) {
  static const long long per_cu_env = env_ll("TS_WGRAD_GROUPS_PER_CU", 0);
  if (per_cu_env > 0) return (static_cast<int>(per_cu_env) * ts::kNumCU + ciblocks - 1) / ciblocks;
  int fit = static_cast<int>((160 * 1024) / (lds_bytes ? lds_bytes : 1));
  fit = fit < 1 ? 1 : (fit > 4 ? 4 : fit);
  const int target = (fit >= 2 ? fit : 2) * ts::kNumCU;
  const int g = target / ciblocks;
  return g < 1 ? 1 : g;
}
int wgrad_groups_max(int ciblocks) {      // what the workspace is sized for
  static const long long per_cu_env = env_ll("TS_WGRAD_GROUPS_PER_CU", 0);
  const int per_cu = per_cu_env > 4 ? static_cast<int>(per_cu_env) : 4;
  return (per_cu * ts::kNumCU + ciblocks - 1) / ciblocks;
}

template <int MODE, int KT, int ST, int DL>
int launch_wgrad(const float* x, const float* dy, float* dw, WG p, void* workspace, size_t workspace_bytes, hipStream_t st) {
  constexpr int TRW = (MODE == MODE_HW) ? (ST == 2 ? 4 : 8) : 8;
  constexpr int NPX = (MODE == MODE_HW) ? TRW * 32 : 256;
  constexpr int NIN = (MODE == MODE_HW) ? ((TRW - 1) * ST + 2 * DL + 1) * (31 * ST + 2 * DL + 1) : KT * 256;
  const int c16 = (p.Cout + 15) / 16 * 16;
  p.cop = c16 + 1;
  while (p.cop % 32 != 17) ++p.cop;
  const size_t lds = (static_cast<size_t>(NIN) * 17 + static_cast<size_t>(NPX) * p.cop) * sizeof(float);
  TS_REQUIRE(lds <= 160 * 1024, TS_ERR_UNSUPPORTED, "conv bwd_weight: tile does not fit the LDS");
  if (MODE == MODE_HW) {
    p.tiles_x = (p.Wo + 31) / 32;
    p.tiles_per_plane = ((p.Ho + TRW - 1) / TRW) * p.tiles_x;
  } else {
    p.tiles_x = 1;
    p.tiles_per_plane = (p.Ho * p.Wo + 255) / 256;
  }
  p.ntiles = p.B * p.Do * p.tiles_per_plane;
  static const bool vec_ok = env_not_zero("TS_WGRAD_VEC");
  p.vec = (MODE == MODE_HW && vec_ok && p.W % 4 == 0 && ts::aligned16(x) && p.x_cstride % 4 == 0 && p.x_bstride % 4 == 0) ? 1 : 0;
  const int ciblocks = (p.Cin + 15) / 16;
  const int cob = (p.Cout + 15) / 16, nitems = KT * cob;
  int gx = wgrad_groups(ciblocks, lds);
  if (gx > p.ntiles) gx = p.ntiles;
  if (gx < 1) gx = 1;
  const size_t need = static_cast<size_t>(gx) * ciblocks * nitems * 256 * sizeof(float);
  TS_REQUIRE(workspace != nullptr && workspace_bytes >= need, TS_ERR_SHAPE,
             "conv bwd_weight: workspace of %zu bytes, %zu needed (ts_conv3d_bwd_weight_workspace_bytes)", workspace_bytes, need);
  float* part = static_cast<float*>(workspace);
  auto kern = &wgrad_kernel<MODE, KT, ST, DL>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipLaunchKernelGGL(kern, dim3(gx, ciblocks), dim3(256), lds, st, x, dy, part, p);
  if (g_wgrad_defer) {             // the finish joins the step's one wgrad_finish_many launch (ts_conv_wgrad_take)
    {
      std::lock_guard<std::mutex> lock(g_wgrad_mutex);
      g_wgrad_pending.push_back(WgradFinishDesc{part, dw, gx, nitems, cob, p.Cin, p.Cout, KT, ciblocks, 0});
    }
    return ts::launched("wgrad_kernel");
  }
  hipLaunchKernelGGL(wgrad_finish, dim3(nitems * 4, ciblocks), dim3(256), 0, st, part, dw, gx, nitems, cob, p.Cin, p.Cout, KT);
  return ts::launched("wgrad_kernel");
}

bool wg_extent(WG& p) {
  const unsigned long long xb = (static_cast<unsigned long long>(p.Cin - 1) * p.x_cstride + static_cast<unsigned long long>(p.D) * p.H * p.W) * 4ull;
  const unsigned long long yb = (static_cast<unsigned long long>(p.Cout - 1) * p.dy_cstride + static_cast<unsigned long long>(p.Do) * p.Ho * p.Wo) * 4ull;
  if (xb >= 0x7fffffffull || yb >= 0x7fffffffull || p.x_cstride < 0 || p.dy_cstride < 0) return false;
  p.x_bytes = static_cast<unsigned>(xb); p.dy_bytes = static_cast<unsigned>(yb);
  return true;
}

int cout_bucket(int cout) {
  for (int b : {8, 16, 32, 64})
    if (cout <= b) return b;
  return cout > 0 ? (cout + 15) / 16 * 16 : -1;       // wide outputs (backward-data of the first layers): 16-channel blocks
}

// Split-K factor of a (1,3,3) convolution: long reductions on grids too small to fill the chip are cut
// into slices handled by separate workgroups (partials summed in a fixed order afterwards).
int conv_hw_ksplit(int B, int Cin, int Cout, int D, int Ho, int Wo, int stride) {
  const long long tiles = static_cast<long long>((Ho + 7) / 8) * ((Wo + 31) / 32) * D * B;
  const int groups = (Cout + 15) / 16;
  int ks = 1;
  // layers that will run on 4 x 16 pixel tiles (launch_ig) have four times the workgroups already: with those, slices pay only while
  // the grid stays below one workgroup per CU (tools/exp/f32_splitk_bench.py with the small tiles in place: 64 -> 64 on 6 x 17 x 30 in
  // two slices 10.6 us, unsplit 8.8; 128 -> 64 on 68 x 120 in four 28.6, unsplit 24.7; 128 -> 16 on 68 x 120 12.6 vs 13.1)
  const bool small_form = Cin > 4 && !(Cout <= 8 && stride == 1 && g_row_pairing && !(g_small_over_pairing && Cin >= 64)) && tiles * groups <= g_small_hw;
  if (small_form) {
    const long long tiles64 = static_cast<long long>((Ho + 3) / 4) * ((Wo + 15) / 16) * D * B;
    while (ks < 8 && tiles64 * groups * ks < ts::kNumCU && Cin / (ks * 2) >= 32) ks *= 2;
    return ks;
  }
  // up to 2 workgroups per CU (round 3, tools/exp/f32_splitk_bench.py: every split layer of config 2 gains 1-38 us from it except the
  // row-paired 176 -> 8 on 5 x 136 x 240, whose 680 workgroups were cut in two under the former 3-per-CU bound: 68.7 vs 65.7 us unsplit)
  while (ks < 8 && tiles * groups * ks < 2 * ts::kNumCU && Cin / (ks * 2) >= 32) ks *= 2;
  return ks;
}
}  // namespace

// x [B,Cin,D,H,W] -> y [B,Cout,D,Ho,Wo]; w_t is [Cin][3][3][CoutPad] with CoutPad = ts_conv_cout_pad(Cout)
// (zero padded), scale/shift [CoutPad].  Channel/batch strides are in elements so that x / y may be
// channel slices of larger tensors (concatenation without a copy).  transposed != 0: stride-2
// ConvTranspose3d(1,3,3) with padding 1, output_padding 1 (Ho = 2H, Wo = 2W).
namespace {
// out_h / out_w: real output size of the transposed form (2H-1 or 2H; 0 = 2H, i.e. output_padding 1)
int conv_hw_impl(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                 int B, int Cin, int Cout, int D, int H, int W, int stride, int dilation,
                 int transposed, int act, float act_param,
                 long long in_bstride, long long in_cstride, long long out_bstride,
                 long long out_cstride, const float* addend, long long addend_bstride,
                 void* workspace, size_t workspace_bytes, int out_h, int out_w, void* stream,
                 long long addend_cstride = 0, long long addend_dstride = 0) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw: non-positive size");
  TS_REQUIRE(stride == 1 || stride == 2, TS_ERR_UNSUPPORTED, "conv3d_hw: stride must be 1 or 2");
  TS_REQUIRE(dilation == 1 || dilation == 2, TS_ERR_UNSUPPORTED, "conv3d_hw: dilation must be 1 or 2");
  TS_REQUIRE(!(stride == 2 && dilation == 2), TS_ERR_UNSUPPORTED, "conv3d_hw: stride 2 with dilation 2");
  TS_REQUIRE(!transposed || (stride == 2 && dilation == 1), TS_ERR_UNSUPPORTED, "conv3d_hw: transposed form is stride 2, dilation 1");
  TS_REQUIRE(act >= 0 && act <= 4, TS_ERR_SHAPE, "conv3d_hw: unknown activation");
  TS_REQUIRE(D <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw: grid too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(y);
  const int bucket = cout_bucket(Cout);
  hipStream_t st = ts::as_stream(stream);
  IG p;
  p.Cin = Cin; p.Cout = Cout; p.coutp = bucket; p.D = D; p.H = H; p.W = W; p.Do = D;
  p.stride = stride; p.dil = dilation; p.pad = dilation; p.k = 3; p.transposed = transposed;
  p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  p.co_groups = 1; p.ksplit = 1; p.kspan = Cin; p.partial = nullptr;
  p.addend = addend; p.add_bstride = addend_bstride;
  p.add_cstride = addend_dstride ? addend_cstride : static_cast<long long>((H - 1) / stride + 1) * ((W - 1) / stride + 1);
  p.add_dstride = addend_dstride;
  TS_REQUIRE(!(addend && transposed), TS_ERR_UNSUPPORTED, "conv3d_hw: no addend in the transposed form");
  TS_REQUIRE(ig_extent(p, 9), TS_ERR_UNSUPPORTED, "conv3d_hw: a batch element of x spans 2 GiB or more");
  if (transposed) {
    p.Ho = out_h ? out_h : 2 * H; p.Wo = out_w ? out_w : 2 * W;
    TS_REQUIRE(p.Ho >= 2 * H - 1 && p.Ho <= 2 * H && p.Wo >= 2 * W - 1 && p.Wo <= 2 * W, TS_ERR_SHAPE,
               "conv3d_hw: transposed output %dx%d is not 2H-1 | 2H", p.Ho, p.Wo);
    p.tiles_x = (W + 31) / 32;
    const int tiles = ((H + 7) / 8) * p.tiles_x;
    return launch_ig<MODE_HWT, 9, 1, 1>(x, w_t, scale, shift, y, p, B, tiles * 4, D, st);
  }
  p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
  p.tiles_x = (p.Wo + 31) / 32;
  const int tiles = ((p.Ho + 7) / 8) * p.tiles_x;
  const int ksplit = addend_dstride ? 1 : conv_hw_ksplit(B, Cin, Cout, D, p.Ho, p.Wo, stride);      // the finishing pass adds a D-invariant term only
  const size_t need = static_cast<size_t>(ksplit) * B * Cout * D * p.Ho * p.Wo * sizeof(float);
  const bool split = ksplit > 1 && workspace != nullptr && workspace_bytes >= need;
  if (split) {
    p.ksplit = ksplit;
    p.kspan = ((Cin + ksplit - 1) / ksplit + 7) / 8 * 8;
    p.partial = reinterpret_cast<float*>(workspace);
  }
  int rc;
  if (stride == 2) rc = launch_ig<MODE_HW, 9, 2, 1>(x, w_t, scale, shift, y, p, B, tiles, D, st);
  else if (dilation == 2) rc = launch_ig<MODE_HW, 9, 1, 2>(x, w_t, scale, shift, y, p, B, tiles, D, st);
  else rc = launch_ig<MODE_HW, 9, 1, 1>(x, w_t, scale, shift, y, p, B, tiles, D, st);
  if (rc || !split) return rc;
  const long long plane = static_cast<long long>(D) * p.Ho * p.Wo;
  TS_REQUIRE(static_cast<long long>(B) * Cout <= 65535 && plane < (1ll << 31), TS_ERR_UNSUPPORTED, "conv3d_hw: split-K finish grid too large");
  long long blocks = (plane + 255) / 256;
  const long long cap = (4096 + static_cast<long long>(B) * Cout - 1) / (static_cast<long long>(B) * Cout);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(conv_splitk_finish, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(B * Cout)), dim3(256), 0, st, p.partial, scale, shift, y,
                     B, Cout, plane, ksplit, act, act_param, out_bstride, out_cstride, addend, addend_bstride,
                     static_cast<long long>(p.Ho) * p.Wo);
  return ts::launched("conv_splitk_finish");
}

// ------------------------------------------------------------------------------------------------
// The first (1,3,3) layer of a sampled level WITHOUT its warped input volume.
//
// Reference: cost = cat[left (repeated over D), warp(right, disp_d), corr] -> Conv3d(1,3,3) + BN + SiLU
// (architecture/modeling/aggregation/TemporalStereo/precise.py:88-91, fine.py:96-103, utils/block_cost.py:47-81).
// The warp is a two-tap linear interpolation along x with per-pixel weights that do not depend on the channel
// (inverse_warp_3d.py:41-56), and the convolution contracts over channels -- the two commute:
//     sum_c W[co][c][t] ((1-f) R[c][x0] + f R[c][x0+1])  ==  (1-f) Q[t][co][x0] + f Q[t][co][x0+1],
//     Q[t][co] = sum_c W[co][c][t] R[c]          (a 1x1 convolution of the right map: ONCE per pixel, not once per candidate)
// so the warped half of the volume (C channels x D candidates: 84 of the 149 MB the 1/4 level wrote and read back per pair) is
// never built, and its share of the layer's multiply-adds drops by the factor D.  What is left per output (co, d, y, x) is a gather:
//     T = left term + sum_{t=(ky,kx)} lerp(Q[t][co][y+ky-1], (x+kx-1) - disp[d][y+ky-1][x+kx-1])        (zero where the tap's pixel
// is outside the image: the convolution's zero padding; zero for columns outside the row: the warp's zeros padding), added to
// the convolution over the correlation channels as a per-depth-plane addend of ig_conv_kernel.
// The tap arithmetic is block_cost.hip's (the reference's normalise / un-normalise float sequence), so tap positions round identically.
// One lane per output pixel of one candidate; the two columns of a tap are one 8-byte load (dword-aligned buffer load).
// ------------------------------------------------------------------------------------------------

template <int DL>
__global__ void __launch_bounds__(256)
warp_gather_kernel(const float* __restrict__ q, const float* __restrict__ disp, const float* __restrict__ base,
                   float* __restrict__ out, int Cout, int D, int H, int W, long long base_bstride) {
  const int cogs = (Cout + 7) / 8;
  // Workgroup order: (batch item, channel group, pixel block, candidate) with the candidate fastest, dealt to the eight XCDs in
  // contiguous bands (consecutive workgroup ids go round the XCDs, each with its own 4 MB L2): an XCD then gathers from ONE band of
  // rows of Q for all candidates (1/8 of the 9.4 MB at the 1/4 level) instead of from all of it.
  unsigned lin = blockIdx.x;
  {
    const unsigned per = gridDim.x / 8;
    if (lin < per * 8) lin = (lin % 8) * per + lin / 8;
  }
  const int d = static_cast<int>(lin % D);
  const unsigned nblk = (static_cast<unsigned>(H) * W + 255u) / 256u;
  const unsigned blk = (lin / D) % nblk;
  const int bc = static_cast<int>(lin / (static_cast<unsigned>(D) * nblk));
  const int cog = bc % cogs, b = bc / cogs;
  const unsigned HW = static_cast<unsigned>(H) * W;
  const unsigned pix = blk * 256u + threadIdx.x;
  const bool live = pix < HW;
  const int y = live ? static_cast<int>(pix / W) : 0, x = live ? static_cast<int>(pix - (pix / W) * W) : 0;
  const __amdgpu_buffer_rsrc_t qrs = ig_rsrc(q + static_cast<size_t>(b) * 9 * Cout * HW, static_cast<unsigned>(9 * Cout) * HW * 4u);
  const __amdgpu_buffer_rsrc_t drs = ig_rsrc(disp + (static_cast<size_t>(b) * D + d) * HW, HW * 4u);
  const float Wm1 = static_cast<float>(W - 1);
  // the candidates of the nine tap pixels, requested together (outside the image: out of range -> 0, and the tap's weights are 0)
  float dv[9];
  bool in[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int qy = y + (t / 3 - 1) * DL, qx = x + (t % 3 - 1) * DL;
    in[t] = live && qy >= 0 && qy < H && qx >= 0 && qx < W;
    dv[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(drs, in[t] ? (static_cast<unsigned>(qy) * W + qx) * 4u : kOOB, 0, 0));
  }
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int co = min(cog * 8 + c, Cout - 1);
    acc[c] = (base && live) ? base[static_cast<size_t>(b) * base_bstride + static_cast<size_t>(co) * HW + pix] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int qy = y + (t / 3 - 1) * DL, qx = x + (t % 3 - 1) * DL;
    // same float sequence as block_cost.hip tap4<true> (inverse_warp_3d.py:41-47 and grid_sample's un-normalisation)
    const float xs = static_cast<float>(qx) + (-dv[t]);
    const float gx = (xs / Wm1 * 2.f) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * Wm1;
    ix = fminf(fmaxf(ix, -2.f), static_cast<float>(W) + 1.f);
    const float fl = floorf(ix);
    const float f = ix - fl;
    const int xi = static_cast<int>(fl);
    // the pair (xb, xb + 1) that covers the taps inside the row; a tap outside it has weight 0
    const int xb = min(max(xi, 0), W - 2);
    float wa = 0.f, wb = 0.f;
    if (xi >= 0 && xi <= W - 2) { wa = 1.f - f; wb = f; }
    else if (xi == -1) wa = f;                       // only the right tap (column 0) is inside
    else if (xi == W - 1) wb = 1.f - f;              // only the left tap (column W-1) is inside
    if (!in[t]) { wa = 0.f; wb = 0.f; }
    const unsigned voff = in[t] ? (static_cast<unsigned>(qy) * W + xb) * 4u : 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned plane = static_cast<unsigned>(t * Cout + min(cog * 8 + c, Cout - 1));      // uniform
      const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(qrs, voff, plane * HW * 4u, 0));
      acc[c] += wa * __uint_as_float(v.x) + wb * __uint_as_float(v.y);
    }
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int co = cog * 8 + c;
      if (co < Cout) out[((static_cast<size_t>(b) * Cout + co) * D + d) * HW + pix] = acc[c];
    }
  }
}

// ---- x6 (bf16-split) form of the stride-1 (1,3,3) convolution ------------------------------------------------------
template <int CB, int DL, int TR = 8>
int launch_x6(const float* x, const void* w6, const float* scale, const float* shift, float* y, const IG& p, dim3 grid, hipStream_t st) {
  constexpr int NPIXP = (TR + 2 * DL) * 40;
  constexpr size_t lds = (static_cast<size_t>(3) * 2 * NPIXP + 3 * X6_SLOTS * 2 * CB * 16 + 1) * 16;
  static_assert(lds <= 80 * 1024, "ig_conv_x6_kernel: two workgroups per CU");
  auto kern = &ig_conv_x6_kernel<CB, DL, TR>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, static_cast<const u32x4*>(w6), scale, shift, y, p);
  return ts::launched("ig_conv_x6_kernel");
}

size_t x6_weight_bytes(int Cin, int Cout) {
  const int bucket = cout_bucket(Cout);
  return static_cast<size_t>((Cin + X6_NC - 1) / X6_NC) * 3 * X6_SLOTS * 2 * bucket * 16;
}
}  // namespace

namespace {
// Split-K factor of the x6 kernel.  Measured layer by layer at config 2 (tools/exp/x6_splitk_bench.py, isolated launches): the split
// pays where a LONG reduction meets a small grid -- 352 -> 32 on 12 x 34 x 60 (120 workgroups, 22 chunks): f32 kernel with its own
// split-K 70 us, x6 unsplit 87 us, x6 in four slices 50 us; 256 -> 64 on 34 x 60 (20 workgroups): 22 / 64 / 19 us -- and costs 4-5 us
// (the finishing launch) where the reduction is short: 32 -> 32 on 120-136 workgroups 16 -> 17.5-21 us, 64 -> 16 on 252: 17 -> 21 us.
// So: only reductions of 16+ chunks (Cin >= 256), as many slices (2 | 4 | 8, two chunks each at least) as keep the grid within the
// chip's 2 x 256 workgroup slots (+ 1/8).  TS_X6_KSPLIT=1 switches it off, =N forces N slices where the channel count allows.
int x6_ksplit(int B, int Cin, int Cout, int D, int H, int W) {
  static const long long forced = env_ll("TS_X6_KSPLIT", 0);
  const int nchunk = (Cin + X6_NC - 1) / X6_NC;
  const int need = (Cout + 15) / 16, cb = need >= 2 ? 2 : 1;
  const long long wgs = static_cast<long long>((H + 7) / 8) * ((W + 31) / 32) * D * B * ((need + cb - 1) / cb);
  int ks = 1;
  if (forced > 0) {
    while (ks * 2 <= forced && ks * 2 <= nchunk && ks < 8) ks *= 2;
  } else if (nchunk >= 16) {
    while (ks < 8 && wgs * ks * 2 <= 2 * ts::kNumCU + ts::kNumCU / 4 && nchunk >= ks * 4) ks *= 2;
  }
  // every slice must own at least one chunk
  while (ks > 1 && ((nchunk + ks - 1) / ks) * (ks - 1) >= nchunk) ks /= 2;
  return ks;
}
}  // namespace

// bytes of the split-K workspace ts_conv3d_hw_x6_fwd wants for this shape (0: the layer runs unsplit)
extern "C" size_t ts_conv3d_hw_x6_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int ks = x6_ksplit(B, Cin, Cout, D, H, W);
  return ks > 1 ? static_cast<size_t>(ks) * B * Cout * D * H * W * sizeof(float) : 0;
}

extern "C" int ts_conv3d_hw_x6_supported(int Cin, int Cout, int W, int stride, int dilation, int transposed) {
  // Cout <= 8: the row-paired f32 kernel (both output rows of a wave in one 16-row MFMA) is as fast at batch 1 and faster at 4
  return Cin >= X6_NC && Cout > 8 && Cout <= 512 && W > 0 && W % 4 == 0 && stride == 1 && (dilation == 1 || dilation == 2) &&
         !transposed;
}

extern "C" size_t ts_conv3d_hw_x6_weight_bytes(int Cin, int Cout) {
  return (Cin > 0 && Cout > 0 && Cout <= 512) ? x6_weight_bytes(Cin, Cout) : 0;
}

extern "C" int ts_conv3d_hw_x6_weight_split(const float* w_t, void* w6, int Cin, int Cout, void* stream) {
  TS_REQUIRE(Cin > 0 && Cout > 0 && Cout <= 512, TS_ERR_SHAPE, "conv3d_hw_x6_weight_split: bad channel counts");
  TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(w6);
  const int bucket = cout_bucket(Cout), nchunk = (Cin + X6_NC - 1) / X6_NC;
  const int n = nchunk * X6_SLOTS * 2 * bucket;
  hipLaunchKernelGGL(weight_split6_kernel, dim3((n + 255) / 256), dim3(256), 0, ts::as_stream(stream), w_t,
                     static_cast<u32x4*>(w6), Cin, bucket, nchunk, 0, 0ll, 0ll, 0ll, 0);
  return ts::launched("weight_split6_kernel");
}

// Layout + split in one launch, straight from the framework's weight (training: the parameters move every step, so both would be
// launched per call): element (ci, co, tap) of the CONVOLUTION being run is w[ci * stride_ci + co * stride_co + tap * stride_t], taps
// reversed when `flip` (the input gradient of a stride-1 layer is the convolution of dy with the flipped taps and ci / co exchanged).
extern "C" int ts_conv3d_hw_x6_weight_split_from(const float* w, void* w6, int Cin, int Cout, long long stride_ci, long long stride_co,
                                                 long long stride_t, int flip, void* stream) {
  TS_REQUIRE(Cin > 0 && Cout > 0 && Cout <= 512, TS_ERR_SHAPE, "conv3d_hw_x6_weight_split_from: bad channel counts");
  TS_REQUIRE_PTR(w); TS_REQUIRE_PTR(w6);
  const int bucket = cout_bucket(Cout), nchunk = (Cin + X6_NC - 1) / X6_NC;
  const int n = nchunk * X6_SLOTS * 2 * bucket;
  hipLaunchKernelGGL(weight_split6_kernel, dim3((n + 255) / 256), dim3(256), 0, ts::as_stream(stream), w,
                     static_cast<u32x4*>(w6), Cin, bucket, nchunk, Cout, stride_ci, stride_co, stride_t, flip);
  return ts::launched("weight_split6_kernel");
}

extern "C" int ts_conv3d_hw_x6_fwd(const float* x, const void* w6, const float* scale, const float* shift, float* y,
                                   int B, int Cin, int Cout, int D, int H, int W, int dilation, int act, float act_param,
                                   long long in_bstride, long long in_cstride, long long out_bstride,
                                   long long out_cstride, const float* addend, long long addend_bstride,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw_x6: non-positive size");
  TS_REQUIRE(ts_conv3d_hw_x6_supported(Cin, Cout, W, 1, dilation, 0), TS_ERR_UNSUPPORTED,
             "conv3d_hw_x6: needs Cin >= 16, 8 < Cout <= 512, W %% 4 == 0, dilation 1 | 2 (Cin=%d Cout=%d W=%d dilation=%d)", Cin, Cout,
             W, dilation);
  TS_REQUIRE(act >= 0 && act <= 4, TS_ERR_SHAPE, "conv3d_hw_x6: unknown activation");
  TS_REQUIRE(D <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw_x6: grid too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w6); TS_REQUIRE_PTR(y);
  IG p;
  p.Cin = Cin; p.Cout = Cout; p.coutp = cout_bucket(Cout); p.D = D; p.H = H; p.W = W; p.Do = D; p.Ho = H; p.Wo = W;
  p.stride = 1; p.dil = dilation; p.pad = dilation; p.k = 3; p.transposed = 0;
  p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  static const int no_xcd = env_not_zero("TS_X6_XCD") ? 0 : 1;
  p.ksplit = 1; p.kspan = (Cin + X6_NC - 1) / X6_NC * X6_NC; p.partial = nullptr; p.B = B; p.xcd = !no_xcd;
  p.addend = addend; p.add_bstride = addend_bstride; p.add_cstride = static_cast<long long>(H) * W; p.add_dstride = 0;
  TS_REQUIRE(ig_extent(p, 9), TS_ERR_UNSUPPORTED, "conv3d_hw_x6: a batch element of x spans 2 GiB or more");
  const size_t wb = x6_weight_bytes(Cin, Cout);
  TS_REQUIRE(wb < 0x7fffffffull, TS_ERR_UNSUPPORTED, "conv3d_hw_x6: weight array too large");
  p.w_bytes = static_cast<unsigned>(wb);
  {
    const unsigned long long plane = static_cast<unsigned long long>(D) * H * W;
    const unsigned long long out_b = (static_cast<unsigned long long>(Cout - 1) * out_cstride + plane) * 4ull;
    TS_REQUIRE(out_b < 0x7fffffffull && out_cstride >= 0, TS_ERR_UNSUPPORTED, "conv3d_hw_x6: a batch element of y spans 2 GiB or more");
    p.out_bytes = static_cast<unsigned>(out_b); p.part_bytes = 0;
  }
  p.tiles_x = (W + 31) / 32;
  const int tiles = ((H + 7) / 8) * p.tiles_x;
  const int need = (Cout + 15) / 16;
  const int cb = need >= 2 ? 2 : 1;
  p.co_groups = (need + cb - 1) / cb;
  // split-K only with a workspace of the size ts_conv3d_hw_x6_workspace_bytes names (none / too small: unsplit, same result up to
  // the order of the fp32 additions)
  const int ksplit = x6_ksplit(B, Cin, Cout, D, H, W);
  const long long plane = static_cast<long long>(D) * H * W;
  const size_t need_ws = static_cast<size_t>(ksplit) * B * Cout * plane * sizeof(float);
  const bool split = ksplit > 1 && workspace != nullptr && workspace_bytes >= need_ws &&
                     static_cast<unsigned long long>(Cout) * plane * 4ull < 0x7fffffffull && static_cast<long long>(B) * Cout <= 65535;
  if (split) {
    const int nchunk = (Cin + X6_NC - 1) / X6_NC;
    p.ksplit = ksplit;
    p.kspan = (nchunk + ksplit - 1) / ksplit * X6_NC;
    p.partial = reinterpret_cast<float*>(workspace);
    p.part_bytes = static_cast<unsigned>(static_cast<unsigned long long>(Cout) * plane * 4ull);
  }
  const dim3 grid(tiles, D, B * p.co_groups * p.ksplit);
  hipStream_t st = ts::as_stream(stream);
  int rc;
  // The ping-pong form (conv_x6p.hip: one 512-thread workgroup per CU, two halves one phase apart, 32 x 32 x 16 MFMAs, weights by LDS-DMA)
  // for unsplit dilation-1 layers with 17+ output channels (its A operand is 32 channels wide) and SiLU / ReLU / no activation, from
  // TS_X6P_MIN_WGS 8 x 32 tiles;
  // TS_X6P=0 keeps every layer on ig_conv_x6_kernel.
  static const bool x6p_on = env_not_zero("TS_X6P");
  static const long long x6p_min = env_ll("TS_X6P_MIN_WGS", 128);        // of its 8 x 32 work items: 136 (128 -> 32 on 136 x 240) 29.3 vs 37.5 us, 72 (128 -> 64 on 68 x 120) 27.0 vs 24.7
  static const bool x6p_split_on = env_not_zero("TS_X6P_KSPLIT");
  if (x6p_on && dilation == 1 && Cout > 16 && act <= ACT_RELU) {
    ts::X6P q;
    q.Cin = Cin; q.Cout = Cout; q.coutp = p.coutp; q.B = B; q.D = D; q.H = H; q.W = W; q.act = act; q.act_param = act_param;
    q.in_bstride = in_bstride; q.in_cstride = in_cstride; q.out_bstride = out_bstride; q.out_cstride = out_cstride;
    q.in_bytes = p.in_bytes; q.w_bytes = p.w_bytes; q.out_bytes = p.out_bytes;
    q.addend = addend; q.add_bstride = addend_bstride; q.add_cstride = p.add_cstride; q.xcd = p.xcd; q.tiles_x = 0; q.co_groups = 0;
    const int nchunk = (Cin + X6_NC - 1) / X6_NC;
    q.ksplit = 1; q.kspan = nchunk; q.partial = nullptr; q.part_bytes = 0;
    // its own split-K: long reductions (16+ chunks) on grids under a round of its one-per-CU workgroups are cut into 2 | 4 | 8 slices of two
    // chunks at least while the items still fit the round (352 -> 32 on 12 x 34 x 60: 120 items x 2; 256 -> 64 on 34 x 60: 20 x 8),
    // within the workspace the caller sized by ts_conv3d_hw_x6_workspace_bytes
    int xks = 1;
    const long long t8 = ts::x6p_grid(q);
    const unsigned long long slice_b = static_cast<unsigned long long>(B) * Cout * plane * sizeof(float);
    if (x6p_split_on && nchunk >= 16 && workspace != nullptr && static_cast<unsigned long long>(Cout) * plane * 4ull < 0x7fffffffull &&
        static_cast<long long>(B) * Cout <= 65535)
      while (xks < 8 && t8 * xks * 2 <= ts::kNumCU + ts::kNumCU / 8 && nchunk >= xks * 4 && slice_b * (xks * 2) <= workspace_bytes) xks *= 2;
    while (xks > 1 && ((nchunk + xks - 1) / xks) * (xks - 1) >= nchunk) xks /= 2;         // every slice owns a chunk
    if (t8 * xks >= x6p_min && D <= 65535 && static_cast<long long>(B) * ((Cout + 31) / 32) <= 65535) {
      if (xks > 1) {
        q.ksplit = xks; q.kspan = (nchunk + xks - 1) / xks;
        q.partial = reinterpret_cast<float*>(workspace);
        q.part_bytes = static_cast<unsigned>(static_cast<unsigned long long>(Cout) * plane * 4ull);
      }
      const int rc6 = ts::x6p_launch(x, w6, scale, shift, y, q, stream);
      if (rc6 || xks == 1) return rc6;
      long long blocks = (plane + 255) / 256;
      const long long cap = (4096 + static_cast<long long>(B) * Cout - 1) / (static_cast<long long>(B) * Cout);
      if (blocks > cap) blocks = cap;
      hipLaunchKernelGGL(conv_splitk_finish, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(B * Cout)), dim3(256), 0, st, q.partial, scale, shift,
                         y, B, Cout, plane, xks, act, act_param, out_bstride, out_cstride, addend, addend_bstride, static_cast<long long>(H) * W);
      return ts::launched("conv_splitk_finish");
    }
  }
  // grids under 3/4 of a round of 8-row workgroups (2 per CU): 4-row tiles (ig_conv_x6_kernel, TR); TS_X6_TR=8 | 4 forces one form
  static const long long tr_env = env_ll("TS_X6_TR", 0);
  static const long long tr4_below = env_ll("TS_X6_TR4_BELOW", 3 * ts::kNumCU / 2);       // measured 384 / 576 / 768 / 1100: 1297 / 1287 / 1286 / 1273 pairs/s with three passes in flight, 934 / 935 / 936 / 942 one at a time
  const long long wgs8 = static_cast<long long>(tiles) * D * B * p.co_groups * p.ksplit;
  const bool tr4 = dilation == 1 && (tr_env ? tr_env == 4 : wgs8 < tr4_below);
  if (tr4) {
    const dim3 grid4(((H + 3) / 4) * p.tiles_x, D, B * p.co_groups * p.ksplit);
    rc = cb == 2 ? launch_x6<2, 1, 4>(x, w6, scale, shift, y, p, grid4, st) : launch_x6<1, 1, 4>(x, w6, scale, shift, y, p, grid4, st);
  } else if (cb == 2) rc = dilation == 2 ? launch_x6<2, 2>(x, w6, scale, shift, y, p, grid, st) : launch_x6<2, 1>(x, w6, scale, shift, y, p, grid, st);
  else rc = dilation == 2 ? launch_x6<1, 2>(x, w6, scale, shift, y, p, grid, st) : launch_x6<1, 1>(x, w6, scale, shift, y, p, grid, st);
  if (rc || !split) return rc;
  long long blocks = (plane + 255) / 256;
  const long long cap = (4096 + static_cast<long long>(B) * Cout - 1) / (static_cast<long long>(B) * Cout);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(conv_splitk_finish, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(B * Cout)), dim3(256), 0, st, p.partial, scale, shift, y,
                     B, Cout, plane, ksplit, act, act_param, out_bstride, out_cstride, addend, addend_bstride, static_cast<long long>(H) * W);
  return ts::launched("conv_splitk_finish");
}

extern "C" int ts_conv3d_hw_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                                int B, int Cin, int Cout, int D, int H, int W, int stride, int dilation,
                                int transposed, int act, float act_param,
                                long long in_bstride, long long in_cstride, long long out_bstride,
                                long long out_cstride, const float* addend, long long addend_bstride,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return conv_hw_impl(x, w_t, scale, shift, y, B, Cin, Cout, D, H, W, stride, dilation, transposed, act, act_param,
                      in_bstride, in_cstride, out_bstride, out_cstride, addend, addend_bstride, workspace, workspace_bytes,
                      0, 0, stream);
}

// ---- first layer of a sampled level from [corr | Q] (see warp_gather_kernel) ----------------------------------------------
extern "C" size_t ts_conv3d_hw_warp_workspace_bytes(int B, int Cout, int D, int H, int W) {
  if (B <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  return ts::round_up(static_cast<size_t>(B) * Cout * D * H * W * sizeof(float), 256);
}

extern "C" int ts_conv3d_hw_warp_fwd(const float* corr, const float* w_t, const float* scale, const float* shift, float* y,
                                     const float* q, const float* disp, const float* base,
                                     int B, int Cc, int Cout, int D, int H, int W, int dilation, int act, float act_param,
                                     long long in_bstride, long long in_cstride, long long out_bstride, long long out_cstride,
                                     long long base_bstride, void* workspace, size_t workspace_bytes, void* stream) {
  TS_REQUIRE(B > 0 && Cc > 0 && Cout > 0 && D > 0 && H > 0 && W >= 2, TS_ERR_SHAPE, "conv3d_hw_warp: bad size");
  TS_REQUIRE(dilation == 1 || dilation == 2, TS_ERR_UNSUPPORTED, "conv3d_hw_warp: dilation must be 1 or 2");
  TS_REQUIRE(D <= 65535 && static_cast<long long>(B) * ((Cout + 7) / 8) <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw_warp: grid too large");
  TS_REQUIRE_PTR(corr); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(y); TS_REQUIRE_PTR(q); TS_REQUIRE_PTR(disp); TS_REQUIRE_PTR(workspace);
  TS_REQUIRE(workspace_bytes >= ts_conv3d_hw_warp_workspace_bytes(B, Cout, D, H, W), TS_ERR_SHAPE, "conv3d_hw_warp: workspace too small");
  TS_REQUIRE(9ull * Cout * H * W * 4ull < 0x7fffffffull, TS_ERR_UNSUPPORTED, "conv3d_hw_warp: a batch element of q spans 2 GiB or more");
  hipStream_t st = ts::as_stream(stream);
  float* T = reinterpret_cast<float*>(workspace);
  const long long HW = static_cast<long long>(H) * W;
  const long long wgs = ((HW + 255) / 256) * D * B * ((Cout + 7) / 8);
  TS_REQUIRE(wgs < (1ll << 31), TS_ERR_UNSUPPORTED, "conv3d_hw_warp: grid too large");
  const dim3 grid(static_cast<unsigned>(wgs));
  if (dilation == 1) hipLaunchKernelGGL(warp_gather_kernel<1>, grid, dim3(256), 0, st, q, disp, base, T, Cout, D, H, W, base_bstride);
  else hipLaunchKernelGGL(warp_gather_kernel<2>, grid, dim3(256), 0, st, q, disp, base, T, Cout, D, H, W, base_bstride);
  if (int rc = ts::launched("warp_gather_kernel")) return rc;
  return conv_hw_impl(corr, w_t, scale, shift, y, B, Cc, Cout, D, H, W, 1, dilation, 0, act, act_param, in_bstride, in_cstride,
                      out_bstride, out_cstride, T, static_cast<long long>(Cout) * D * HW, nullptr, 0, 0, 0, stream,
                      static_cast<long long>(D) * HW, HW);
}

// Gradient w.r.t. the input of ts_conv3d_hw_fwd (no scale / shift / activation: the raw convolution).
// (B, Cin, Cout, D, H, W, stride, dilation, transposed) describe the FORWARD call; dy has the forward
// output's shape, dx [B,Cin,D,H,W].  w_b is the weight re-laid for the backward pass by the host,
// [Cout][9][ts_conv_cout_pad(Cin)]:
//   forward stride 1        : w_b[co][t][ci] = W[co][ci][8 - t]   (correlation with the flipped taps)
//   forward stride 2        : w_b[co][t][ci] = W[co][ci][t]       (runs as the stride-2 transposed form)
//   forward transposed (s2) : w_b[co][t][ci] = W_T[ci][co][t]     (runs as a stride-2 convolution of dy)
extern "C" int ts_conv3d_hw_bwd_data(const float* dy, const float* w_b, float* dx, int B, int Cin, int Cout, int D, int H,
                                     int W, int stride, int dilation, int transposed, long long dy_bstride,
                                     long long dy_cstride, long long dx_bstride, long long dx_cstride, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw_bwd_data: non-positive size");
  TS_REQUIRE(stride == 1 || stride == 2, TS_ERR_UNSUPPORTED, "conv3d_hw_bwd_data: stride must be 1 or 2");
  if (transposed)            // y = convT(x): [2H, 2W]  ->  dx = stride-2 convolution of dy
    return conv_hw_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, D, 2 * H, 2 * W, 2, 1, 0, ACT_NONE, 0.f, dy_bstride,
                        dy_cstride, dx_bstride, dx_cstride, nullptr, 0, nullptr, 0, 0, 0, stream);
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (stride == 2)           // dx = transposed convolution of dy, cropped to the forward input
    return conv_hw_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, D, Ho, Wo, 2, 1, 1, ACT_NONE, 0.f, dy_bstride,
                        dy_cstride, dx_bstride, dx_cstride, nullptr, 0, nullptr, 0, H, W, stream);
  return conv_hw_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, D, Ho, Wo, 1, dilation, 0, ACT_NONE, 0.f, dy_bstride,
                      dy_cstride, dx_bstride, dx_cstride, nullptr, 0, nullptr, 0, 0, 0, stream);
}

// Gradient w.r.t. the weight of the NON-transposed ts_conv3d_hw_fwd: x [B,Cin,D,H,W], dy [B,Cout,D,Ho,Wo]
// -> dw [Cout][Cin][9] (torch layout, OVERWRITTEN).  Cout <= 64.  The transposed form's weight gradient is
// the same call with the roles of x and dy exchanged (x := dy of the 2H x 2W output, dy := x, stride 2),
// which yields [Cin][Cout][9] -- ConvTranspose3d's own layout.
// Upper bound of the partial-sum workspace of both bwd_weight entry points (taps = 9 | k).
extern "C" size_t ts_conv3d_bwd_weight_workspace_bytes(int Cin, int Cout, int taps) {
  if (Cin <= 0 || Cout <= 0 || taps <= 0) return 0;
  const int ciblocks = (Cin + 15) / 16, cob = (Cout + 15) / 16;
  return static_cast<size_t>(wgrad_groups_max(ciblocks)) * ciblocks * taps * cob * 256 * sizeof(float);
}

extern "C" int ts_conv3d_hw_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H,
                                       int W, int stride, int dilation, long long x_bstride, long long x_cstride,
                                       long long dy_bstride, long long dy_cstride, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw_bwd_weight: non-positive size");
  TS_REQUIRE(Cout <= 64, TS_ERR_UNSUPPORTED, "conv3d_hw_bwd_weight: Cout=%d > 64", Cout);
  TS_REQUIRE((stride == 1 && (dilation == 1 || dilation == 2)) || (stride == 2 && dilation == 1), TS_ERR_UNSUPPORTED,
             "conv3d_hw_bwd_weight: stride/dilation outside {1/1, 1/2, 2/1}");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(dy); TS_REQUIRE_PTR(dw);
  WG p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.D = D; p.H = H; p.W = W; p.Do = D;
  p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
  p.stride = stride; p.dil = dilation; p.pad = dilation; p.k = 3;
  p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.dy_bstride = dy_bstride; p.dy_cstride = dy_cstride;
  TS_REQUIRE(wg_extent(p), TS_ERR_UNSUPPORTED, "conv3d_hw_bwd_weight: a batch element spans 2 GiB or more");
  hipStream_t st = ts::as_stream(stream);
  if (stride == 2) return launch_wgrad<MODE_HW, 9, 2, 1>(x, dy, dw, p, workspace, workspace_bytes, st);
  if (dilation == 2) return launch_wgrad<MODE_HW, 9, 1, 2>(x, dy, dw, p, workspace, workspace_bytes, st);
  return launch_wgrad<MODE_HW, 9, 1, 1>(x, dy, dw, p, workspace, workspace_bytes, st);
}

// x [B,Cin,Din,H,W] -> y [B,Cout,Dout,H,W]; w_t is [Cin][k][CoutPad].  k in {1,3,5}.  transposed != 0:
// ConvTranspose3d(3,1,1) stride 2, padding 1, output_padding 1 (Dout = 2 Din).
namespace {
// out_d: real output depth of the transposed form (2Din-1 or 2Din; 0 = 2Din)
int conv_d_impl(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                int B, int Cin, int Cout, int Din, int H, int W, int k, int stride, int dilation,
                int padding, int transposed, int act, float act_param,
                long long in_bstride, long long in_cstride, long long out_bstride,
                long long out_cstride, int out_d, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Din > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_d: non-positive size");
  TS_REQUIRE(k == 1 || k == 3 || k == 5, TS_ERR_UNSUPPORTED, "conv3d_d: k must be 1, 3 or 5");
  TS_REQUIRE(stride >= 1 && stride <= 2 && dilation >= 1 && padding >= 0, TS_ERR_UNSUPPORTED, "conv3d_d: bad stride/dilation/padding");
  TS_REQUIRE(!transposed || (k == 3 && stride == 2 && dilation == 1 && padding == 1), TS_ERR_UNSUPPORTED,
             "conv3d_d: transposed form is k=3, stride 2, padding 1");
  TS_REQUIRE(act >= 0 && act <= 3, TS_ERR_SHAPE, "conv3d_d: unknown activation");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(y);
  const int bucket = cout_bucket(Cout);
  const int Dout = transposed ? (out_d ? out_d : 2 * Din) : (Din + 2 * padding - dilation * (k - 1) - 1) / stride + 1;
  TS_REQUIRE(Dout > 0 && Dout <= 65535, TS_ERR_SHAPE, "conv3d_d: bad output depth");
  TS_REQUIRE(!transposed || (Dout >= 2 * Din - 1 && Dout <= 2 * Din), TS_ERR_SHAPE, "conv3d_d: transposed depth %d is not 2D-1 | 2D", Dout);
  hipStream_t st = ts::as_stream(stream);
  IG p;
  p.Cin = Cin; p.Cout = Cout; p.coutp = bucket; p.D = Din; p.H = H; p.W = W; p.Do = Dout; p.Ho = H; p.Wo = W;
  p.stride = stride; p.dil = dilation; p.pad = padding; p.k = k; p.transposed = transposed;
  p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  p.tiles_x = 1; p.co_groups = 1; p.ksplit = 1; p.kspan = Cin; p.partial = nullptr;
  p.addend = nullptr; p.add_bstride = 0; p.add_cstride = 0; p.add_dstride = 0;
  TS_REQUIRE(ig_extent(p, k), TS_ERR_UNSUPPORTED, "conv3d_d: a batch element of x spans 2 GiB or more");
  const int tiles = (H * W + 255) / 256;
  if (k == 1) return launch_ig<MODE_D, 1, 1, 1>(x, w_t, scale, shift, y, p, B, tiles, Dout, st);
  if (k == 3) return launch_ig<MODE_D, 3, 1, 1>(x, w_t, scale, shift, y, p, B, tiles, Dout, st);
  return launch_ig<MODE_D, 5, 1, 1>(x, w_t, scale, shift, y, p, B, tiles, Dout, st);
}
}  // namespace

extern "C" int ts_conv3d_d_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                               int B, int Cin, int Cout, int Din, int H, int W, int k, int stride, int dilation,
                               int padding, int transposed, int act, float act_param,
                               long long in_bstride, long long in_cstride, long long out_bstride,
                               long long out_cstride, void* stream) {
  return conv_d_impl(x, w_t, scale, shift, y, B, Cin, Cout, Din, H, W, k, stride, dilation, padding, transposed, act,
                     act_param, in_bstride, in_cstride, out_bstride, out_cstride, 0, stream);
}

// Gradient w.r.t. the input of ts_conv3d_d_fwd (raw convolution).  The geometry arguments describe the
// FORWARD call; dy has the forward output's shape, dx [B,Cin,Din,H,W].  w_b [Cout][k][ts_conv_cout_pad(Cin)]:
//   forward stride 1        : w_b[co][t][ci] = W[co][ci][k-1-t]
//   forward stride 2 (k = 3, padding 1) : w_b[co][t][ci] = W[co][ci][t]      (transposed form, cropped to Din)
//   forward transposed      : w_b[co][t][ci] = W_T[ci][co][t]               (stride-2 convolution of dy)
extern "C" int ts_conv3d_d_bwd_data(const float* dy, const float* w_b, float* dx, int B, int Cin, int Cout, int Din, int H,
                                    int W, int k, int stride, int dilation, int padding, int transposed,
                                    long long dy_bstride, long long dy_cstride, long long dx_bstride, long long dx_cstride,
                                    void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Din > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_d_bwd_data: non-positive size");
  if (transposed)
    return conv_d_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, 2 * Din, H, W, 3, 2, 1, 1, 0, ACT_NONE, 0.f, dy_bstride,
                       dy_cstride, dx_bstride, dx_cstride, 0, stream);
  const int Dout = (Din + 2 * padding - dilation * (k - 1) - 1) / stride + 1;
  TS_REQUIRE(Dout > 0, TS_ERR_SHAPE, "conv3d_d_bwd_data: bad forward geometry");
  if (stride == 2) {
    TS_REQUIRE(k == 3 && dilation == 1 && padding == 1, TS_ERR_UNSUPPORTED, "conv3d_d_bwd_data: stride 2 needs k=3, padding 1");
    return conv_d_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, Dout, H, W, 3, 2, 1, 1, 1, ACT_NONE, 0.f, dy_bstride,
                       dy_cstride, dx_bstride, dx_cstride, Din, stream);
  }
  TS_REQUIRE(stride == 1, TS_ERR_UNSUPPORTED, "conv3d_d_bwd_data: stride must be 1 or 2");
  return conv_d_impl(dy, w_b, nullptr, nullptr, dx, B, Cout, Cin, Dout, H, W, k, 1, dilation, dilation * (k - 1) - padding, 0,
                     ACT_NONE, 0.f, dy_bstride, dy_cstride, dx_bstride, dx_cstride, 0, stream);
}

// Gradient w.r.t. the weight of the NON-transposed ts_conv3d_d_fwd -> dw [Cout][Cin][k] (OVERWRITTEN), Cout <= 64;
// transposed form: exchange x and dy as for ts_conv3d_hw_bwd_weight.
extern "C" int ts_conv3d_d_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int Din, int H,
                                      int W, int k, int stride, int dilation, int padding, long long x_bstride,
                                      long long x_cstride, long long dy_bstride, long long dy_cstride, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Din > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_d_bwd_weight: non-positive size");
  TS_REQUIRE(Cout <= 64, TS_ERR_UNSUPPORTED, "conv3d_d_bwd_weight: Cout=%d > 64", Cout);
  TS_REQUIRE(k == 1 || k == 3 || k == 5, TS_ERR_UNSUPPORTED, "conv3d_d_bwd_weight: k must be 1, 3 or 5");
  TS_REQUIRE(stride >= 1 && stride <= 2 && dilation >= 1 && padding >= 0, TS_ERR_UNSUPPORTED, "conv3d_d_bwd_weight: bad stride/dilation/padding");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(dy); TS_REQUIRE_PTR(dw);
  WG p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.D = Din; p.H = H; p.W = W; p.Ho = H; p.Wo = W;
  p.Do = (Din + 2 * padding - dilation * (k - 1) - 1) / stride + 1;
  TS_REQUIRE(p.Do > 0, TS_ERR_SHAPE, "conv3d_d_bwd_weight: bad output depth");
  p.stride = stride; p.dil = dilation; p.pad = padding; p.k = k;
  p.x_bstride = x_bstride; p.x_cstride = x_cstride; p.dy_bstride = dy_bstride; p.dy_cstride = dy_cstride;
  TS_REQUIRE(wg_extent(p), TS_ERR_UNSUPPORTED, "conv3d_d_bwd_weight: a batch element spans 2 GiB or more");
  hipStream_t st = ts::as_stream(stream);
  if (k == 1) return launch_wgrad<MODE_D, 1, 1, 1>(x, dy, dw, p, workspace, workspace_bytes, st);
  if (k == 3) return launch_wgrad<MODE_D, 3, 1, 1>(x, dy, dw, p, workspace, workspace_bytes, st);
  return launch_wgrad<MODE_D, 5, 1, 1>(x, dy, dw, p, workspace, workspace_bytes, st);
}

// ConvTranspose2d(kernel 4, stride 2, padding 1) of UNet (module.py:453-457): x [B,Cin,H,W] -> y [B,Cout,2H,2W]
// (y may be a channel slice: out_bstride in elements); w_t [Cin][4][4][CoutPad], CoutPad = ts_conv_cout_pad(Cout) = 8 | 16 | 32
// (round 5: this entry assumed 16 for Cout <= 8 while every caller lays weights out with ts_conv_cout_pad: wrong outputs for Cout <= 8).
extern "C" int ts_deconv2d_k4s2_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                                    int B, int Cin, int Cout, int H, int W, int act, long long out_bstride, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "deconv2d: non-positive size");
  TS_REQUIRE(Cout <= 32, TS_ERR_UNSUPPORTED, "deconv2d: Cout=%d > 32", Cout);
  TS_REQUIRE(act >= 0 && act <= 2, TS_ERR_SHAPE, "deconv2d: unknown activation");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(scale); TS_REQUIRE_PTR(shift); TS_REQUIRE_PTR(y);
  IG p;
  p.Cin = Cin; p.Cout = Cout; p.coutp = cout_bucket(Cout); p.D = 1; p.H = H; p.W = W; p.Do = 1; p.Ho = 2 * H; p.Wo = 2 * W;
  p.stride = 2; p.dil = 1; p.pad = 1; p.k = 4; p.transposed = 1; p.act = act; p.act_param = 0.f;
  p.in_bstride = static_cast<long long>(Cin) * H * W; p.in_cstride = static_cast<long long>(H) * W;
  p.out_bstride = out_bstride; p.out_cstride = 4ll * H * W;
  p.tiles_x = (W + 31) / 32; p.co_groups = 1; p.ksplit = 1; p.kspan = Cin; p.partial = nullptr;
  p.addend = nullptr; p.add_bstride = 0; p.add_cstride = 0; p.add_dstride = 0;
  TS_REQUIRE(ig_extent(p, 16), TS_ERR_UNSUPPORTED, "deconv2d: a batch element of x spans 2 GiB or more");
  const int tiles = ((H + 7) / 8) * p.tiles_x;
  return launch_ig<MODE_HWT, 16, 1, 1>(x, w_t, scale, shift, y, p, B, tiles * 4, 1, ts::as_stream(stream));
}

extern "C" int ts_conv_cout_pad(int cout) { return cout_bucket(cout); }

// Deferred weight-gradient finishes (see wgrad_finish_many).  defer(1): the ts_conv3d_*_bwd_weight calls that follow ON THIS HOST
// THREAD leave their partial sums in the caller's workspaces (which must stay alive) and dw unwritten; returns the previous setting.
extern "C" int ts_conv_wgrad_defer(int on) {
  const int was = g_wgrad_defer ? 1 : 0;
  g_wgrad_defer = on != 0;
  return was;
}
extern "C" int ts_conv_wgrad_pending(void) {
  std::lock_guard<std::mutex> lock(g_wgrad_mutex);
  return static_cast<int>(g_wgrad_pending.size());
}
// Moves the pending descriptors (of every thread of the process) into `host_table` (capacity in bytes; 48 bytes per entry, block0 filled in):
// *n entries, *total_blocks workgroups for ts_conv_wgrad_finish_many.  The list is cleared.
extern "C" int ts_conv_wgrad_take(void* host_table, size_t capacity_bytes, int* n, int* total_blocks) {
  TS_REQUIRE_PTR(host_table); TS_REQUIRE_PTR(n); TS_REQUIRE_PTR(total_blocks);
  std::lock_guard<std::mutex> lock(g_wgrad_mutex);
  const size_t cnt = g_wgrad_pending.size();
  TS_REQUIRE(cnt * sizeof(WgradFinishDesc) <= capacity_bytes, TS_ERR_SHAPE, "conv_wgrad_take: %zu entries do not fit %zu bytes", cnt, capacity_bytes);
  int blocks = 0;
  WgradFinishDesc* out = static_cast<WgradFinishDesc*>(host_table);
  for (size_t i = 0; i < cnt; ++i) {
    out[i] = g_wgrad_pending[i];
    out[i].block0 = blocks;
    blocks += out[i].nitems * 4 * out[i].ciblocks;
  }
  *n = static_cast<int>(cnt);
  *total_blocks = blocks;
  g_wgrad_pending.clear();
  return TS_OK;
}
// table: n entries (ts_wgrad_finish_desc) in DEVICE memory, as written by ts_conv_wgrad_take
extern "C" int ts_conv_wgrad_finish_many(const void* table, int n, int total_blocks, void* stream) {
  TS_REQUIRE(n > 0 && total_blocks > 0, TS_ERR_SHAPE, "conv_wgrad_finish_many: empty table");
  TS_REQUIRE_PTR(table);
  hipLaunchKernelGGL(wgrad_finish_many, dim3(static_cast<unsigned>(total_blocks)), dim3(256), 0, ts::as_stream(stream),
                     static_cast<const WgradFinishDesc*>(table), n);
  return ts::launched("wgrad_finish_many");
}

// Cap the K-chunk size (and with it the LDS footprint: 16-106 KB per workgroup at 32, 16-50 KB at 8) of the
// convolution launches that follow on this host thread.  Long chunks shorten a lone kernel's dependent
// chain; short chunks let workgroups of kernels on OTHER streams share a CU.  Thread-local, ordered with
// the launches, recordable in a plan.
extern "C" int ts_conv_set_chunk_cap(int cap) {
  TS_REQUIRE(cap == 8 || cap == 16 || cap == 32, TS_ERR_SHAPE, "conv_set_chunk_cap: %d is not 8, 16 or 32", cap);
  g_chunk_cap = cap;
  return TS_OK;
}

// bytes of scratch ts_conv3d_hw_fwd can use for split-K at this shape (0: it will not split)
extern "C" size_t ts_conv3d_hw_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride, int transposed) {
  if (transposed || B <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int ks = conv_hw_ksplit(B, Cin, Cout, D, Ho, Wo, stride);
  return ks > 1 ? static_cast<size_t>(ks) * B * Cout * D * Ho * Wo * sizeof(float) : 0;
}

extern "C" int ts_conv_weight_layout(const float* w, float* out, int A, int T, int nb, int bpad, long long stride_a,
                                     long long stride_b, long long stride_t, int flip, void* stream) {
  TS_REQUIRE(A > 0 && T > 0 && nb > 0 && bpad >= nb, TS_ERR_SHAPE, "conv_weight_layout: bad size");
  TS_REQUIRE_PTR(w); TS_REQUIRE_PTR(out);
  const long long n = static_cast<long long>(A) * T * bpad;
  TS_REQUIRE(n < (1ll << 31), TS_ERR_UNSUPPORTED, "conv_weight_layout: 2^31 elements or more");
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(weight_layout_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ts::as_stream(stream), w, out, A, T, nb,
                     bpad, stride_a, stride_b, stride_t, flip);
  return ts::launched("weight_layout_kernel");
}

// table: n entries of ts_weight_layout_desc in DEVICE memory; blocks_x: workgroups per entry (entries larger than
// blocks_x * 256 elements are covered by a grid-stride loop, smaller ones leave the surplus workgroups idle).
extern "C" int ts_conv_weight_layout_many2(const void* table, int n, int blocks_x, void* stream) {
  TS_REQUIRE(n > 0 && n <= 65535 && blocks_x > 0 && blocks_x <= 4096, TS_ERR_SHAPE, "conv_weight_layout_many2: bad table size");
  TS_REQUIRE_PTR(table);
  hipLaunchKernelGGL(weight_layout_many2_kernel, dim3(static_cast<unsigned>(blocks_x), static_cast<unsigned>(n)), dim3(256), 0,
                     ts::as_stream(stream), static_cast<const WeightLayoutDesc2*>(table));
  return ts::launched("weight_layout_many2_kernel");
}

extern "C" int ts_conv_weight_layout_many(const void* table, int n, int blocks_x, void* stream) {
  TS_REQUIRE(n > 0 && n <= 65535 && blocks_x > 0 && blocks_x <= 4096, TS_ERR_SHAPE, "conv_weight_layout_many: bad table size");
  TS_REQUIRE_PTR(table);
  hipLaunchKernelGGL(weight_layout_many_kernel, dim3(static_cast<unsigned>(blocks_x), static_cast<unsigned>(n)), dim3(256), 0,
                     ts::as_stream(stream), static_cast<const WeightLayoutDesc*>(table));
  return ts::launched("weight_layout_many_kernel");
}
