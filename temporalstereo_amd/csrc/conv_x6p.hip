// K3 -- the bf16-split ("x6") stride-1 (1,3,3) convolution rebuilt around what bounded ig_conv_x6_kernel (conv3d.hip): its three
// phases -- staging (global fetch, fp32 -> 3 x bf16 split, LDS commit), the matrix loop, the epilogue -- ADDED UP instead of hiding
// behind one another (profiles/r05_x6_split_ablation.txt: 74.5 + 132 + 60 us of a 249 us launch at 128 -> 32, 4 x 272 x 480), because
// nothing made the two co-resident workgroups of a CU alternate; and the matrix loop itself ran v_mfma_f32_16x16x32_bf16 over ten tap
// slots for nine taps.  Reference layers: the Conv3d wrappers of layers/basic_layers.py:194-235 in eval mode (BatchNorm folded),
// UNet.encoder / decoder (aggregation/TemporalStereo/module.py:424-492) and the (1,3,3) halves of DepthwiseConv3D (module.py:111-147).
//
// ig_conv_x6p_kernel ("p": ping-pong):
//   * ONE 512-thread workgroup per CU = two HALVES of four waves, one wave of each half on every SIMD.  A half owns HR x 32 output
//     pixels (HR = 8 or 4 rows; the workgroup's tile is 2 HR x 32) and 32 output channels, its own LDS input tile, and runs
//         stage(c) | compute(c) | stage(c + 1) | compute(c + 1) ...
//     with the OTHER half exactly one phase behind, a workgroup barrier between phases: while one half multiplies, the other half
//     splits and commits its next chunk (VALU + LDS writes beside the other wave's MFMAs on the same SIMD), and the epilogue of a
//     half falls into a phase in which the other half still multiplies.  The alternation the two independent workgroups of the old
//     kernel never found by themselves is forced.
//   * v_mfma_f32_32x32x16_bf16: K = 16 = ONE tap x 16 input channels, so nine taps are nine steps (no zero slot: -10 % MFMA work) and
//     half the LDS bytes per FLOP of 16 x 16 x 32.  A = weights (32 output channels x 16), B = pixels (16 x 32 pixels of one row),
//     D[channel][pixel].  (One instruction per 32.3 cycles from one wave per SIMD, tools/exp/mfma_issue_rate.hip; ~35 in this step's form, 40 stamped in the kernel.)
//   * the split weights (already [chunk][part][slot][group][co] x 8 bf16 in global memory, ts_conv3d_hw_x6_weight_split) go to LDS by
//     LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no commit pass, double-buffered and SHARED by the two halves (they
//     multiply the same chunk one phase apart).  The zero slot 9 of the global layout is simply not fetched.
//   * numerics as ig_conv_x6_kernel: six products per fp32 product (smallest first), a chunk's products summed in accumulators of
//     their own and added to the running sum with an fp32 add.  tests/test_conv_x6_gpu.py holds both kernels to the same fp64 bounds.
//   * PERSISTENT: at most 208 workgroups walk the (batch, plane, channel group, tile[, K slice]) order, XCD-banded; a tile's epilogue
//     (scale / shift / activation, wave-private LDS transposition, 16-byte stores) needs no barrier of its own and falls into a phase in
//     which the other half multiplies.  Split-K as work items (raw sums to the caller's workspace, conv_splitk_finish in conv3d.hip).
// LDS (16-byte units): weights [2 (HR = 4: 3) buffers][54 rows = (part, tap, group)][32 co] first (inside the first 64 KB: the DMA's
// M0 base), then per half the input tile [part 3][group 2][(HR + 2) x 40 pixels], the waves' output staging and scale / shift:
// 55,296 + 2 x 38,400 + 17,408 + 4,096 = 153,600 bytes at HR = 8.  DESIGN.md section 4 (K3x6p) has the measurements.
#include "conv_common.hpp"
#include "conv_x6p.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int XP_NC = 16;                 // input channels per chunk (= K of one MFMA)
constexpr int XP_WROWS = 3 * 9 * 2;       // LDS weight rows per chunk: (part, tap, group), 32 output channels x 16 bytes each
constexpr int XP_WBUF = XP_WROWS * 32;    // 16-byte units per weight buffer
constexpr int XP_GSLOTS = 10;             // tap slots of the GLOBAL weight layout (slot 9: zeros, not fetched)
constexpr int XP_MAXC = 512;              // widest layer (ts_conv3d_hw_x6_supported)

__device__ __forceinline__ void phase_barrier() {
  // LDS writes / reads of this wave are done (lgkmcnt), then the workgroup barrier.  NOT __syncthreads(): with an LDS-DMA or the next
  // chunk's fetch in flight its fence would drain vmcnt too.
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int HR>
struct XPG {
  static constexpr int in_rows = HR + 2, QPR = 10, LCOLS = 40, NPIX = in_rows * LCOLS;
  static constexpr int SLOTS = in_rows * QPR;           // (row, quad) staging slots; threads [0, SLOTS) stage channels 0-7, [SLOTS, 2 SLOTS) 8-15
  static constexpr int NPB = HR / 4;                    // 32-pixel blocks (= rows) per wave
  static constexpr int HALF_U = 3 * 2 * NPIX;           // 16-byte units of a half's input tile
  // HR = 4 has the LDS for a third weight buffer and needs it: its matrix phase is half as long, and with all 27 LDS-DMA instructions
  // of a step in half 1's stage phase (100-185 cycles of issue each) that phase was the longer one.  Three buffers let BOTH halves issue
  // their share in their own stage phase (the buffer being filled was last read two steps ago).  HR = 8: two buffers, half 1 issues.
  static constexpr int NWB = HR == 4 ? 3 : 2;
  // staging threads per (row, quad): HR = 8: two (8 channels each, 16-byte LDS stores); HR = 4: four (4 channels each, 8-byte
  // stores by lane PAIRS of one pixel unit) -- 240 of 256 threads stage instead of 120
  static constexpr int SPC = HR == 4 ? 4 : 2;
  static constexpr int NCH = 16 / SPC;
  // weights, the two halves' tiles, 8 waves x 8 channels of output staging, scale + shift
  static constexpr int TILE0 = NWB * XP_WBUF;           // first 16-byte unit of the halves' tiles
  static constexpr size_t lds_bytes = (static_cast<size_t>(TILE0) + 2 * HALF_U) * 16 + (8 * 8 * (NPB * 32 + 4) + 2 * XP_MAXC) * 4;
  static_assert(SPC * SLOTS <= 256, "staging slots");
};

// one tile of the work list: (batch element, depth plane, output-channel group, 2 HR x 32 pixel tile)
struct XPTile {
  int b, od, co0, ty0, tx0;
  int c0, c1;       // chunk range of the work item (split-K: a slice of the input channels, raw sums to the workspace)
  int ks;
};

template <int HR>
__global__ void __launch_bounds__(512, 2)
ig_conv_x6p_kernel(const float* __restrict__ x, const u32x4* __restrict__ w6, const float* __restrict__ scale,
                   const float* __restrict__ shift, float* __restrict__ y, const ts::X6P p) {
  using G = XPG<HR>;
  extern __shared__ __attribute__((aligned(16))) u32x4 ldsp[];
  constexpr int NPIX = G::NPIX, LCOLS = G::LCOLS, NPB = G::NPB;
  const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
  const int wave = tid >> 6, lane = tid & 63;
  const int px = lane & 31, grp = lane >> 5;
  u32x4* wls = ldsp;                                          // [buffer][row][32]
  u32x4* in6 = ldsp + G::TILE0 + half * G::HALF_U;         // [part][group][NPIX]

  // ---- the work list.  The launch is PERSISTENT: min(tiles, 256) workgroups, each walking its share of the
  // (batch, plane, channel group, tile) order, so that a workgroup's prologue (first fetch, first weights: ~8 k of ~96 k cycles) is
  // paid once, and a half starts its next tile while the other half still multiplies the last chunk of this one.
  // XCD-aware: workgroup g runs on XCD g % 8; an XCD owns a contiguous band of the order and its 32 workgroups work on 32
  // consecutive tiles at any time (the halo rows / columns neighbouring tiles share are served by that XCD's L2).
  const int total = p.total_tiles;
  const int nwg = gridDim.x, g = blockIdx.x;
  int band0, band1, first, step;
  if (p.xcd && nwg % 8 == 0) {
    const int per = (total + 7) / 8, xcd = g & 7;
    band0 = xcd * per; band1 = min(total, band0 + per);
    first = band0 + (g >> 3); step = nwg >> 3;
  } else {
    band0 = 0; band1 = total; first = g; step = nwg;
  }
  const int nchunk = (p.Cin + XP_NC - 1) / XP_NC;
  auto tile_at = [&](int L) {
    XPTile t;
    const int tile = L % p.tiles_pp;
    int r = L / p.tiles_pp;
    const int cog = r % p.co_groups; r /= p.co_groups;
    t.od = r % p.D; r /= p.D;
    t.b = r % p.B; t.ks = r / p.B;
    t.c0 = t.ks * p.kspan; t.c1 = min(nchunk, t.c0 + p.kspan);
    t.co0 = cog * 32;
    t.ty0 = (tile / p.tiles_x) * (2 * HR) + half * HR;
    t.tx0 = (tile % p.tiles_x) * 32;
    return t;
  };
  const unsigned HW = static_cast<unsigned>(p.H) * p.W;
  const unsigned cstride_b = static_cast<unsigned>(p.in_cstride) * 4u;

  // ---- staging deal: thread -> (channel group, row, aligned quad) of the half's (HR + 2) x 40 input tile
  constexpr int SPC = G::SPC, NCH = G::NCH;
  const bool stager = tid < SPC * G::SLOTS;
  // HR = 8: thread -> (group sg, slot); HR = 4: lanes 2j, 2j + 1 are the two 4-channel halves (shalf) of one (group, slot)
  const int sg = SPC == 2 ? tid / G::SLOTS : tid / (2 * G::SLOTS);
  const int shalf = SPC == 2 ? 0 : (tid & 1);
  const int sslot = SPC == 2 ? tid - sg * G::SLOTS : (tid - sg * 2 * G::SLOTS) >> 1;
  const int srow = sslot / G::QPR, squad = sslot - srow * G::QPR;
  // LDS pixel units are stored with the pixel-in-quad index XORed by bit 1 of the quad index (swz below): the eight lanes of a
  // ds_write_b128 bank group are eight consecutive quads, 64 bytes apart -- four of them on each half of the 32 write banks (4-way
  // conflicts, ~1.2 k cycles of a 2.5 k-cycle commit); with the swizzle 2-way, which a 16-byte store hides.  A function of the quad
  // index modulo 4 only, so any 16 consecutive logical pixels still cover the 16 units of a 256-byte LDS row once: the
  // fragment reads stay conflict-free.
  const int lpix = sg * NPIX + srow * LCOLS + 4 * squad;
  const int wsw = (sslot >> 1) & 1;
  const __amdgpu_buffer_rsrc_t wr = ig_rsrc(w6, p.w_bytes);
  const unsigned wchunk_b = static_cast<unsigned>(3 * XP_GSLOTS * 2 * p.coutp) * 16u;

  // experiment: cycle stamps of the first wave of each half of ONE workgroup (TS_X6P_TRACE=<workgroup id>), its first tiles
  unsigned long long* trc = (p.trace && g == p.trace_wg && (tid == 0)) ? p.trace + half * 256 : nullptr;
  int ntr = 0;
  auto stamp = [&]() {
    if (trc && ntr < 250) trc[ntr] = __builtin_amdgcn_s_memtime();
    ++ntr;
  };
  stamp();
  // experiment (TS_X6P_TRACE=-2): start / end stamp of EVERY workgroup
  unsigned long long* wgt = (p.trace && p.trace_wg == -2 && threadIdx.x == 0 && g < 8000) ? p.trace + 512 + 4 * g : nullptr;
  if (wgt) { wgt[0] = __builtin_amdgcn_s_memtime(); wgt[3] = __builtin_amdgcn_s_memrealtime(); }

  // the tile whose chunks are being FETCHED (one step ahead of the tile being multiplied)
  __amdgpu_buffer_rsrc_t xr;
  unsigned goff = kOOB;
  auto aim = [&](const XPTile& t) {
    xr = ig_rsrc(x + static_cast<long long>(t.b) * p.in_bstride, p.in_bytes);
    const int gy = t.ty0 - 1 + srow, gx = t.tx0 - 4 + 4 * squad;
    goff = (stager && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        ? (static_cast<unsigned>(t.od) * HW + static_cast<unsigned>(gy) * p.W + gx) * 4u : kOOB;
  };
  v4f rin[NCH];
  auto fetch = [&](int c) {
#pragma unroll
    for (int e = 0; e < NCH; ++e) {
      // channels past Cin re-read the last real one; their weights are zero
      const unsigned co = static_cast<unsigned>(min(c * XP_NC + sg * 8 + shalf * 4 + e, p.Cin - 1)) * cstride_b;
      rin[e] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xr, (goff == kOOB || (p.dbg & 1)) ? kOOB : goff + co, 0, 0));
    }
  };
  // weight DMA: instruction i (0..26) = (part, tap) = (i / 9, i % 9), its 64 lanes = (group, co); wave w of half 1 issues i = w, w + 4, ...
  // share: -1 = all 27 instructions by this half's four waves; 0 | 1 = the even | odd ones (both halves take part, NWB == 3)
  auto dma_weights = [&](int c, int co0, int buf, int share) {
    u32x4* dst = wls + buf * XP_WBUF;
    const unsigned cbase = static_cast<unsigned>(c) * wchunk_b;
    const bool lane_ok = co0 + px < p.coutp && !(p.dbg & 16);
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const int i = share < 0 ? wave + 4 * q : 2 * (wave + 4 * q) + share;
      if (i < 27 && (share < 0 || q < 4)) {
        const int pt = i / 9, tap = i - pt * 9;
        const unsigned row = static_cast<unsigned>((pt * XP_GSLOTS + tap) * 2 + grp);
        const unsigned off = lane_ok ? (row * p.coutp + co0 + px) * 16u + cbase : kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_ptr_t)(dst + i * 64), 16, off, 0, 0, 0);
      }
    }
  };
  auto commit = [&]() {
    if (stager && !(p.dbg & 4)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned part[3][NCH / 2];
#pragma unroll
        for (int e = 0; e < NCH / 2; ++e) split6(rin[2 * e][i], rin[2 * e + 1][i], part[0][e], part[1][e], part[2][e]);
#pragma unroll
        for (int pt = 0; pt < 3; ++pt) {
          if constexpr (SPC == 2) in6[pt * 2 * NPIX + lpix + (i ^ wsw)] = u32x4{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
          else reinterpret_cast<u32x2*>(in6 + pt * 2 * NPIX + lpix + (i ^ wsw))[shalf] = u32x2{part[pt][0], part[pt][1]};
        }
      }
    }
  };

  // fragment bases (16-byte units): B = pixel px of each of the wave's rows at each tap, group grp; A = channel px, group grp
  auto swz = [](int P) { return (P & ~3) | ((P & 3) ^ ((P >> 3) & 1)); };      // logical pixel unit of the tile -> stored unit
  // a fragment of output row R, tap (ky, kx) starts at tile row R + ky: two address registers per kx (even / odd tile row of the
  // wave's first row), rows two further down are +80 units as an immediate (80 = 10 x 8: the swizzle's bit is unchanged)
  int rbase[2][3];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) rbase[rr][kx] = grp * NPIX + swz((wave * NPB + rr) * LCOLS + 3 + kx + px);
  const int aoff = grp * 32 + px;

  constexpr int CH = NPB == 1 ? 2 : 1;                  // accumulator chains per pixel block (HR = 4: one block per wave, two chains in turn)
  constexpr int NACC = CH * NPB;
  f32x16 acc[NPB];

  // ---- epilogue of one tile: scale / shift / activation on the accumulators; then, eight channels at a time, through a WAVE-PRIVATE
  // 2 KB of LDS into [channel][pixel] order, so that a lane stores 16 bytes and an instruction eight whole 128-byte rows.  Stored
  // straight from the 32 x 32 C/D layout (32 dword stores per lane) the epilogue took ~9 k cycles, issue-bound
  // (tools/exp/x6p_trace.py).  Wave-private: no barrier of its own -- it simply lengthens this half's next stage phase while the
  // other half multiplies.  scale / shift were copied to LDS when the workgroup started.
  constexpr int EPITCH = NPB * 32 + 4;
  float* scr = reinterpret_cast<float*>(ldsp + G::TILE0 + 2 * G::HALF_U) + (threadIdx.x >> 6) * (8 * EPITCH);
  const float* ss = reinterpret_cast<const float*>(ldsp + G::TILE0 + 2 * G::HALF_U) + 8 * 8 * EPITCH;
  auto epilogue = [&](const XPTile& t) {
    // split-K (p.ksplit > 1): the slice's RAW sums go to the workspace, [slice][batch][channel][plane]; conv_splitk_finish (conv3d.hip)
    // adds the slices in a fixed order and applies addend / scale / shift / activation
    const bool split = p.ksplit > 1;
    const __amdgpu_buffer_rsrc_t yr = split
        ? ig_rsrc(p.partial + (static_cast<size_t>(t.ks) * p.B + t.b) * p.Cout * (static_cast<size_t>(p.D) * HW), p.part_bytes)
        : ig_rsrc(y + static_cast<long long>(t.b) * p.out_bstride, p.out_bytes);
    const unsigned ocs = split ? static_cast<unsigned>(p.D) * HW * 4u : static_cast<unsigned>(p.out_cstride) * 4u;
    const float* ab = (p.addend && !split) ? p.addend + static_cast<size_t>(t.b) * p.add_bstride : nullptr;
    const unsigned obase = static_cast<unsigned>(t.od) * HW;
    // (uniform conditions outside the element loops: inside, hipcc put a branch and an s_waitcnt vmcnt(0) -- which also waits for the
    // previous round's STORES -- in front of every element)
    if (ab) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          const int oy = t.ty0 + wave * NPB + pb, ox = t.tx0 + px;
          acc[pb][r] += ab[static_cast<size_t>(min(t.co0 + (r & 3) + 8 * (r >> 2) + 4 * grp, p.Cout - 1)) * p.add_cstride +
                           ((oy < p.H && ox < p.W) ? static_cast<unsigned>(oy) * p.W + ox : 0u)];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {                       // accumulator register r holds channel (r & 3) + 8 (r >> 2) + 4 grp (32 x 32 C/D map)
      const int cc = min(t.co0 + (r & 3) + 8 * (r >> 2) + 4 * grp, p.coutp - 1);
      const float sc = split ? 1.f : ss[cc], sh = split ? 0.f : ss[XP_MAXC + cc];
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb) acc[pb][r] = acc[pb][r] * sc + sh;
    }
    if (split) {
    } else if (p.act == ACT_SILU) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb][r] = silu_fast(acc[pb][r]);
    } else if (p.act == ACT_RELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) acc[pb][r] = fmaxf(acc[pb][r], 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) scr[(j + 4 * grp) * EPITCH + pb * 32 + px] = acc[pb][4 * k + j];
#pragma unroll
      for (int m = 0; m < NPB; ++m) {
        const int i = lane + 64 * m;
        const int c8 = i / (8 * NPB), q = i % (8 * NPB);
        const v4f val = *reinterpret_cast<const v4f*>(scr + c8 * EPITCH + q * 4);
        const int oy = t.ty0 + wave * NPB + (q >> 3), ox = t.tx0 + (q & 7) * 4;
        const int co = t.co0 + 8 * k + c8;
        const unsigned off = (oy < p.H && ox < p.W && co < p.Cout && !(p.dbg & 8))
            ? (obase + static_cast<unsigned>(oy) * p.W + ox) * 4u + static_cast<unsigned>(co) * ocs : kOOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), yr, off, 0, 0);
      }
    }
  };

  // The weight DMA is half 1's job, issued in its STAGE phases: an LDS-DMA instruction costs its wave 100-185 cycles of issue while
  // the data paths are busy (MI355X_MICROARCH.md, instruction constants) -- seven of them at the head of half 0's matrix loop made
  // that loop 1.3 k cycles longer than half 1's (tools/exp/x6p_trace.py); a staging wave has the issue slots to spare.
  //   step s + 1 -> buffer (s + 1) & 1, issued in half 1's stage(s) (phase 2s + 1): the buffer's last reader was half 1's own
  //   compute(s - 1) (phase 2s); landed by the end of half 1's compute(s) (phase 2s + 2: vmcnt(8) = everything but the eight
  //   fetches issued after it), read from phase 2s + 3 on (half 0's compute(s + 1)).  Step 0 goes out in half 1's idle phase.
  // (a step = one chunk of one tile; steps run through the workgroup's tiles without a seam)
  if (first >= band1) return;
  {
    float* ssw = reinterpret_cast<float*>(ldsp + G::TILE0 + 2 * G::HALF_U) + 8 * 8 * EPITCH;
    for (int i = threadIdx.x; i < p.coutp; i += 512) {
      ssw[i] = scale ? scale[i] : 1.f;
      ssw[XP_MAXC + i] = shift ? shift[i] : 0.f;
    }
  }                           // (the whole workgroup: an XCD's band can be shorter than its workgroup count)
  XPTile cur = tile_at(first);
  aim(cur);
  fetch(cur.c0);
  constexpr int NWB = G::NWB;
  if (NWB == 3) dma_weights(cur.c0, cur.co0, 0, half);
  else if (half == 1) dma_weights(cur.c0, cur.co0, 0, -1);
  if (half == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();                                    // half 1 runs one phase behind
  }
  stamp();
  int wcur = 0, wnext = 1;                              // weight buffers of this step and the next (rotating through NWB)
  for (int L = first; L < band1; L += step) {
    const bool more = L + step < band1;
    const XPTile nxt = more ? tile_at(L + step) : cur;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
    for (int c = cur.c0; c < cur.c1; ++c, wcur = wnext, wnext = (wnext + 1 == NWB ? 0 : wnext + 1)) {
      const bool last = c + 1 == cur.c1;
      // ---- stage: the fetched registers have landed; split and commit
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp();
      if ((NWB == 3 || half == 1) && (!last || more))
        dma_weights(last ? nxt.c0 : c + 1, last ? nxt.co0 : cur.co0, wnext, NWB == 3 ? half : -1);
      commit();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp();
      phase_barrier();
      stamp();
      // ---- compute
      if (!last) fetch(c + 1);
      else if (more) { aim(nxt); fetch(nxt.c0); }
      const u32x4* wb = wls + wcur * XP_WBUF;
      f32x16 part[NACC];
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[a][r] = 0.f;
      bf16x8 af[2][3], bf[2][3][NPB];
      auto load_frag = [&](int tap, int buf) {
        // in the order of first use (products (0,2) (2,0) (1,1) ...): the step's first MFMAs wait for the OLDEST reads only
        constexpr int OA[3] = {0, 2, 1}, OB[3] = {2, 0, 1};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          af[buf][OA[k]] = __builtin_bit_cast(bf16x8, wb[(OA[k] * 9 + tap) * 64 + aoff]);
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) bf[buf][OB[k]][pb] = __builtin_bit_cast(bf16x8, in6[OB[k] * 2 * NPIX + rbase[(pb + tap / 3) & 1][tap % 3] + ((pb + tap / 3) >> 1) * 80]);
        }
      };
      if (!(p.dbg & 2)) {
        load_frag(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3 + 3 * NPB, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) load_frag(tap + 1, (tap + 1) & 1);
          constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int pb = 0; pb < NPB; ++pb) {
              const int a = pb * CH + (t % CH);
              part[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap & 1][PA[t]], bf[tap & 1][PB[t]][pb], part[a], 0, 0, 0);
            }
          // issue order of the step: the next tap's fragment reads go out one behind each of the first MFMAs (a read issues in the 32
          // cycles the matrix pipe is busy anyway).  Left to itself hipcc sinks every read to just above its first use: ~200 exposed
          // cycles per tap, 5.3 k cycles per chunk where the MFMAs alone are 4.3 k (tools/exp/x6p_trace.py).
          if (tap + 1 < 9) {
#pragma unroll
            for (int k = 0; k < 3 + 3 * NPB; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (6 * NPB > 3 + 3 * NPB) __builtin_amdgcn_sched_group_barrier(0x008, 6 * NPB - (3 + 3 * NPB), 0);       // (a group of size 0 undoes the pinning: HR = 4 ran read bursts and exposed waits)
          } else {
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NPB, 0);
          }
        }
      }
      // the chunk's products were summed apart from the running sum (ig_conv_x6_kernel, conv3d.hip: the matrix core aligns an
      // instruction's products to the largest addend, the accumulator included)
#pragma unroll
      for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = part[pb * CH][r];
#pragma unroll
          for (int k = 1; k < CH; ++k) v += part[pb * CH + k][r];
          acc[pb][r] += v;
        }
      if (trc) asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[NPB - 1][15]));
      stamp();
      // the next step's weights (this half's share of them) are in LDS: everything but the fetches issued after them has landed
      if (NWB == 3 || half == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NCH) : "memory");
      phase_barrier();
      stamp();
    }
    epilogue(cur);
    stamp();
    cur = nxt;
  }
  if (wgt) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wgt[1] = __builtin_amdgcn_s_memtime(); wgt[2] = __builtin_amdgcn_s_memrealtime(); }
  if (half == 0) phase_barrier();                       // both halves pass the same number of barriers
}

template <int HR>
int launch_x6p(const float* x, const void* w6, const float* scale, const float* shift, float* y, const ts::X6P& p, int nwg, hipStream_t st) {
  constexpr size_t lds = XPG<HR>::lds_bytes;
  static_assert(lds <= 160 * 1024, "ig_conv_x6p_kernel: LDS");
  auto kern = &ig_conv_x6p_kernel<HR>;
  static const int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  (void)attr;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, x, static_cast<const u32x4*>(w6), scale, shift, y, p);
  return ts::launched("ig_conv_x6p_kernel");
}

}  // namespace

namespace ts {

static unsigned long long* trace_buf = nullptr;
constexpr size_t kTraceBytes = 4096 + 8000 * 32;

static long long tiles_of(const X6P& p, int hr) {
  return static_cast<long long>((p.H + 2 * hr - 1) / (2 * hr)) * ((p.W + 31) / 32) * p.D * p.B * ((p.Cout + 31) / 32) * (p.ksplit > 1 ? p.ksplit : 1);
}

int x6p_rows(const X6P& p) {
  // Half-tile rows, 8 or 4.  One workgroup per CU, so a launch lasts (rounds of 256 tiles) x (a tile's time); a 16 x 32 tile costs ~1 / 0.58
  // of two 8 x 32 ones (less halo, half the weight reads) but quantises worse: 128 -> 32 on 136 x 240 is 72 big tiles (one round at
  // 28 % of the chip, 43.9 us) or 136 small ones (29.3 us); 64 -> 64 on 2 x 136 x 240 is 288 / 544 tiles: 2 x 1 vs 3 x 0.58
  // (tools/exp/x6p_check.py, table in profiles/r06_x6p_layers.txt).  TS_X6P_HR=4 | 8 forces one.
  static const long long forced = env_ll("TS_X6P_HR", 0);
  if (forced == 4 || forced == 8) return static_cast<int>(forced);
  const long long r8 = (tiles_of(p, 8) + kNumCU - 1) / kNumCU, r4 = (tiles_of(p, 4) + kNumCU - 1) / kNumCU;
  return 100 * r8 <= 58 * r4 ? 8 : 4;
}

// 8 x 32 work items of the layer, slices included (the caller's threshold between this kernel and ig_conv_x6_kernel)
long long x6p_grid(const X6P& p) { return tiles_of(p, 4); }

int x6p_launch(const float* x, const void* w6, const float* scale, const float* shift, float* y, X6P p, void* stream) {
  const int hr = x6p_rows(p);
  static const long long dbg = env_ll("TS_X6P_DBG", 0);
  p.dbg = static_cast<int>(dbg);
  static const long long trace_wg = env_ll("TS_X6P_TRACE", -1);
  if (trace_wg != -1 && !trace_buf) { (void)hipMalloc(reinterpret_cast<void**>(&trace_buf), kTraceBytes); (void)hipMemset(trace_buf, 0, kTraceBytes); }
  p.trace = trace_buf; p.trace_wg = static_cast<int>(trace_wg);
  p.tiles_x = (p.W + 31) / 32;
  p.co_groups = (p.Cout + 31) / 32;
  p.tiles_pp = ((p.H + 2 * hr - 1) / (2 * hr)) * p.tiles_x;
  if (p.ksplit < 1) p.ksplit = 1;
  const long long total = static_cast<long long>(p.tiles_pp) * p.co_groups * p.D * p.B * p.ksplit;
  if (total > 0x3fffffff) return fail(TS_ERR_UNSUPPORTED, "conv3d_hw_x6: too many tiles");
  p.total_tiles = static_cast<int>(total);
  // persistent: one workgroup per CU (its 140-154 KB of LDS fill one), at most TS_X6P_WGS of them, and no more than the rounds need
  // (510 tiles under a cap of 224 are three rounds: 170 workgroups take three tiles each, the other CUs stay free for the launches of
  // the pass's other streams -- which cannot share a CU with this kernel's LDS footprint)
  // Cap 208 of 256: measured 256 / 232 / 224 / 208 / 192 / 176 / 160 / 128 -> 1318 / 1335 / 1322 / 1333 / 1330 / 1313 / 1305 / 1266 pairs/s with
  // three passes in flight, batch 4 1763 / 1804 / 1802 / 1798 / 1810 / 1771 / 1767 / 1742, one pass at a time 961-965 throughout (>= 160): what a
  // pass's throughput follows is CU-time, not this launch's latency, and a workgroup that walks more tiles pays its prologue once.
  static const long long max_wgs0 = env_ll("TS_X6P_WGS", 208);
  // ... except a split-K launch of <= 256 items, which runs in ONE round: the coarse level's first layer (352 -> 32 on 12 x 34 x 60: 120
  // tiles x 2 slices) is the longest launch on a pass's critical chain, 66.5 us as 120 workgroups x 2 items, 45.6 as 240 x 1; three
  // passes in flight do not notice (1540-1546 pairs/s either way), one pass at a time gains 1-1.5 % (982 -> 991-997).  The 255-tile
  // layers of the UNet stay under the cap: all of the chip for them costs the pass's other streams more than it saves (972 vs 982-990).
  static const long long splitk_all = env_ll("TS_X6P_SPLITK_ALL", 1);
  const long long max_wgs = (splitk_all && p.ksplit > 1 && total <= 256) ? 256 : max_wgs0;
  int nwg = static_cast<int>(total < max_wgs ? total : max_wgs);
  {
    const long long rounds = (total + nwg - 1) / nwg;
    const long long even = (total + rounds - 1) / rounds;
    nwg = static_cast<int>(((even + 7) / 8 * 8 <= nwg) ? (even + 7) / 8 * 8 : even);
  }
  hipStream_t st = as_stream(stream);
  return hr == 8 ? launch_x6p<8>(x, w6, scale, shift, y, p, nwg, st) : launch_x6p<4>(x, w6, scale, shift, y, p, nwg, st);
}

}  // namespace ts

// experiment (tools/exp/x6p_trace.py): the cycle stamps of the traced workgroup, 2 halves x 256, to the host; not part of the ABI header
extern "C" int ts_x6p_trace_read(unsigned long long* host) {
  if (!ts::trace_buf) return -1;
  (void)hipDeviceSynchronize();
  return static_cast<int>(hipMemcpy(host, ts::trace_buf, ts::kTraceBytes, hipMemcpyDeviceToHost));
}
