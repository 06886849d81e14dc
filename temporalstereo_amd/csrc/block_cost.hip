// K1 -- cost-volume construction for gfx950 (MI355X).
//
// Replaces block_cost()  architecture/modeling/aggregation/utils/block_cost.py:16-83 of the
// reference (int path :34-45, sampled path :47-58 through inverse_warp_3d.py:4-58, multi-scale
// group correlation :66-81).  Where the reference materialises the shifted/warped right volume
// twice, pools it three times and concatenates, this is ONE streaming pass:
//
//   kernel A  (block_cost_main):  workgroup = (4 feature rows) x (one group of 8 channels) x all
//       candidates.  The 8x4 right-feature rows are staged once into LDS in a 4-way
//       de-interleaved layout (element x lives at [x&3][x>>2]) so that a wavefront whose lanes
//       own 4 consecutive pixels each gathers conflict-free; each lane owns a 4x4 pixel block of
//       one candidate, emits the main channels and the scale-0 group correlation with 16-byte
//       coalesced stores, and reduces the 2x2 / 4x4 pooled differences of its own block in
//       registers (no cross-lane traffic) into two tiny pooled maps.
//   kernel B  (block_cost_upsample): bilinear (align_corners) expansion of the pooled maps into
//       the scale-1/2 channel blocks, again 16 bytes per lane.
//
// HBM-bound: algorithmic bytes = inputs once + output once; the only
// extra traffic is the pooled maps (< 2 % of the output).  There is no inter-workgroup reuse, so
// no XCD-aware block remap is needed here (cdna guide T1: 0 % on ops without shared panels).
#include <cstdlib>

#include "ts_common.hpp"

namespace {

constexpr int GRP = 8;  // channels per correlation group (block_cost.py:8)
constexpr int TR = 4;   // feature rows per workgroup == pooling footprint of scale 2

struct Shape {
  int B, C, H, W, D, scales;
  int G;          // C / 8
  int mainC;      // C (int path) or 2C (sampled path; C when the reference half is omitted)
  int Ctot;       // mainC + scales * G
  int omit_ref;   // sampled path: 1 = do not write the D-fold broadcast of `left` (ts_block_cost_sampled_warped_fwd);
                  // 2 = correlation blocks only, neither half of the main channels (ts_block_cost_sampled_corr_fwd)
  int tch;        // sampled path: first channel of the warped half (C, or 0 when the reference half is omitted)
  int H1, W1, H2, W2;
  int nbx, nby;   // 4x4 pixel blocks
  int nbxp;       // nbx padded so that a wavefront does not straddle candidates (when cheap)
  int Wq, Wqp;    // quarter-row length and its padded LDS stride
  float rh1, rw1, rh2, rw2;  // align_corners scales (in-1)/(out-1) of the two pooled maps
};

template <bool VEC>
__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int x, int W) {
  if constexpr (VEC) {
    return *reinterpret_cast<const float4*>(row + x);
  } else {
    float4 v;
    v.x = (x + 0 < W) ? row[x + 0] : 0.f;
    v.y = (x + 1 < W) ? row[x + 1] : 0.f;
    v.z = (x + 2 < W) ? row[x + 2] : 0.f;
    v.w = (x + 3 < W) ? row[x + 3] : 0.f;
    return v;
  }
}

// MEASURED gfx950 HAZARD (tools/exp, DESIGN.md "toolchain notes"): a 128-bit VMEM store followed by
// an LDS load whose destination reuses the store's data VGPRs can have those VGPRs overwritten
// before the (back-pressured) memory pipe has read them -- lanes 12-15 / 28-31 / ... of the stored
// vector then carry the LDS data.  hipcc only pads VALU writers.  EXP_CNT covers the store's operand
// read, so `s_waitcnt expcnt(0)` right after the store closes the window.
// Second sighting (round 1, later): a buffer_store_dwordx4 WITH AN SGPR soffset immediately followed by a
// v_mul that writes its first data register -- LLVM's hazard recognizer exempts exactly that form
// (GCNHazardRecognizer: "only if the instruction is not using a register in the soffset field") and the
// scheduler had hoisted the v_mul above a separate fence.  The buffer stores of bst4 therefore carry their
// fence inside the same asm block; this helper remains for the plain global stores of st4.
__device__ __forceinline__ void store_data_fence() { __builtin_amdgcn_s_waitcnt(0xcf0f); }   // expcnt(0) only

template <bool VEC>
__device__ __forceinline__ void st4(float* __restrict__ row, int x, int W, float4 v) {
  if constexpr (VEC) {
    *reinterpret_cast<float4*>(row + x) = v;
    store_data_fence();
  } else {
    if (x + 0 < W) row[x + 0] = v.x;
    if (x + 1 < W) row[x + 1] = v.y;
    if (x + 2 < W) row[x + 2] = v.z;
    if (x + 3 < W) row[x + 3] = v.w;
  }
}

__device__ __forceinline__ void unpack(const float4 v, float (&a)[4]) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
__device__ __forceinline__ float4 pack(const float (&a)[4]) { return make_float4(a[0], a[1], a[2], a[3]); }


// Source taps of one output pixel: the two neighbouring right-feature columns (as LDS or row
// offsets) and their weights (0 outside the row == zeros padding).
template <bool SAMPLED, bool STAGE>
__device__ __forceinline__ void tap(int x, int d, float dispv, int W, float Wm1, int Wqp,
                                    int& o0, int& o1, float& w0, float& w1) {
  int xi;
  float f = 0.f;
  if constexpr (SAMPLED) {
    // same float sequence as the reference: normalise to [-1,1] (inverse_warp_3d.py:41-47)
    // and back (grid_sampler align_corners), so the tap position rounds identically
    const float xs = static_cast<float>(x) + (-dispv);
    const float gx = (xs / Wm1 * 2.f) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * Wm1;
    ix = fminf(fmaxf(ix, -2.f), static_cast<float>(W) + 1.f);   // keeps int conversion defined
    const float fl = floorf(ix);
    f = ix - fl;
    xi = static_cast<int>(fl);
  } else {
    xi = x - d;
  }
  const bool v0 = (xi >= 0) & (xi < W);
  const bool v1 = (xi + 1 >= 0) & (xi + 1 < W);
  w0 = v0 ? (1.f - f) : 0.f;
  w1 = v1 ? f : 0.f;
  const int i0 = min(max(xi, 0), W - 1);
  const int i1 = min(max(xi + 1, 0), W - 1);
  if constexpr (STAGE) {
    o0 = (i0 & 3) * Wqp + (i0 >> 2);
    o1 = (i1 & 3) * Wqp + (i1 >> 2);
  } else {
    o0 = i0;
    o1 = i1;
  }
}

// cooperative copy of the 8x4 right-feature rows of this workgroup into LDS, de-interleaved
// [c][r][x&3][x>>2]: a float4 read from global is scattered to the four quarter-rows
template <bool VEC>
__device__ __forceinline__ void stage_right_rows(float* lds, const float* __restrict__ Rg, int y0, int H, int W,
                                                 size_t HW, int Wq, int Wqp) {
  const int n = GRP * TR * Wq;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int j = i % Wq, cr = i / Wq;
    const int y = y0 + (cr & (TR - 1));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < H) v = ld4<VEC>(Rg + static_cast<size_t>(cr >> 2) * HW + static_cast<size_t>(y) * W, 4 * j, W);
    float* dst = lds + static_cast<size_t>(cr) * 4 * Wqp + j;
    dst[0] = v.x;
    dst[Wqp] = v.y;
    dst[2 * Wqp] = v.z;
    dst[3 * Wqp] = v.w;
  }
  __syncthreads();
}

// SAMPLED: per-pixel fractional candidates (block_cost.py:47-58); else integer shift d (:34-45).
// VEC:     W % 4 == 0 and 16-byte aligned bases -> float4 traffic.
// STAGE:   the workgroup's feature rows live in LDS: all 8x4 right rows (de-interleaved for the
//          gather) and the left rows two at a time (linear, read back with ds_read_b128; rows 2-3
//          are fetched into registers during the prologue and swapped in at half time).  After the
//          prologue the only vector-memory traffic of a wave is its store stream plus one
//          prefetched candidate row, so a wave never waits on its own stores (vmcnt is in-order and
//          counts stores on CDNA4).  ~47 KiB at W=240 -> three workgroups per CU, so the 544
//          workgroups of the 136x240 level are co-resident in a single round.
//          Rows too wide for that (or needing more than one pass of items) read both maps
//          through L1/L2 instead.
// NP:      left float4 values each thread carries across the first half (STAGE only).
template <bool SAMPLED, bool VEC, bool STAGE, int NP>
__global__ void __launch_bounds__(512)
block_cost_main(const float* __restrict__ L, const float* __restrict__ R,
                const float* __restrict__ disp, float* __restrict__ out,
                float* __restrict__ P1, float* __restrict__ P2, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const size_t HW = static_cast<size_t>(H) * W;
  const float* Lg = L + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Rg = R + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const int Wl = 4 * s.Wq;                                    // LDS row length of the left rows
  float* ldsL = lds + static_cast<size_t>(GRP) * TR * 4 * s.Wqp;   // [c][r & 1][Wl]

  float4 lpre[NP];   // left rows 2,3 in flight across the first half
  if constexpr (STAGE) {
    const int n = GRP * TR * s.Wq;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int j = i % s.Wq, cr = i / s.Wq;
      const int r = cr & (TR - 1), c = cr >> 2;
      const int y = y0 + r;
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), lv = rv;
      if (y < H) {
        const size_t off = static_cast<size_t>(c) * HW + static_cast<size_t>(y) * W;
        rv = ld4<VEC>(Rg + off, 4 * j, W);
        if (r < 2) lv = ld4<VEC>(Lg + off, 4 * j, W);
      }
      float* dst = lds + static_cast<size_t>(cr) * 4 * s.Wqp + j;
      dst[0] = rv.x;
      dst[s.Wqp] = rv.y;
      dst[2 * s.Wqp] = rv.z;
      dst[3 * s.Wqp] = rv.w;
      if (r < 2) *reinterpret_cast<float4*>(ldsL + static_cast<size_t>(c * 2 + r) * Wl + 4 * j) = lv;
    }
    const int n2 = GRP * 2 * s.Wq;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = threadIdx.x + p * blockDim.x;
      lpre[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < n2) {
        const int j = i % s.Wq, cr = i / s.Wq;          // cr = c*2 + (r-2)
        const int y = y0 + 2 + (cr & 1);
        if (y < H) lpre[p] = ld4<VEC>(Lg + static_cast<size_t>(cr >> 1) * HW + static_cast<size_t>(y) * W, 4 * j, W);
      }
    }
    __syncthreads();
  }

  const float Wm1 = static_cast<float>(W - 1);
  const int nitems = s.nbxp * D;
  // STAGE runs exactly one item per thread (the host guarantees nitems <= blockDim) so that the
  // half-time barrier is uniform; the fallback loops.
  for (int item = threadIdx.x; item < (STAGE ? blockDim.x : nitems); item += blockDim.x) {
    const int d = item / s.nbxp;
    const int bx = item - d * s.nbxp;
    const bool live = (item < nitems) && (bx < s.nbx);
    if (!STAGE && !live) continue;
    const int x4 = bx * 4;
    float* plane0 = out + (static_cast<size_t>(b) * s.Ctot * D + (live ? d : 0)) * HW;  // channel 0, candidate d
    const size_t cstride = static_cast<size_t>(D) * HW;                                 // one channel
    const float* drow = SAMPLED ? disp + (static_cast<size_t>(b) * D + (live ? d : 0)) * HW : nullptr;

    float s1[GRP][2];   // 2x2 pooled difference sums of the current row pair
    float s2[GRP];      // 4x4 pooled difference sums
#pragma unroll
    for (int c = 0; c < GRP; ++c) s1[c][0] = s1[c][1] = s2[c] = 0.f;
    const size_t pbase = ((static_cast<size_t>(b) * s.G + g) * D + d);

    float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (SAMPLED) {
      if (live) dnext = ld4<VEC>(drow + static_cast<size_t>(y0) * W, x4, W);
    }

#pragma unroll 1
    for (int r = 0; r < TR; ++r) {
      const int y = y0 + r;
      if constexpr (STAGE) {
        if (r == 2) {   // swap left rows 2,3 into the LDS slots of rows 0,1
          __syncthreads();
          const int n2 = GRP * 2 * s.Wq;
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const int i = threadIdx.x + p * blockDim.x;
            if (i < n2) {
              const int j = i % s.Wq, cr = i / s.Wq;
              *reinterpret_cast<float4*>(ldsL + static_cast<size_t>(cr) * Wl + 4 * j) = lpre[p];
            }
          }
          __syncthreads();
        }
      }
      if (y < H && live) {
        float dv[4];
        unpack(dnext, dv);
        if constexpr (SAMPLED) {   // prefetch the next candidate row ahead of this row's stores
          if (r + 1 < TR && y + 1 < H) dnext = ld4<VEC>(drow + static_cast<size_t>(y + 1) * W, x4, W);
        }
        int o0[4], o1[4];
        float w0[4], w1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          tap<SAMPLED, STAGE>(x4 + k, d, dv[k], W, Wm1, s.Wqp, o0[k], o1[k], w0[k], w1[k]);
        float g0[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t rowoff = static_cast<size_t>(y) * W;
#pragma unroll
        for (int c = 0; c < GRP; ++c) {
          float4 lv4;
          const float* src;
          if constexpr (STAGE) {
            lv4 = *reinterpret_cast<const float4*>(ldsL + static_cast<size_t>(c * 2 + (r & 1)) * Wl + x4);
            src = lds + static_cast<size_t>(c * TR + r) * 4 * s.Wqp;
          } else {
            lv4 = ld4<VEC>(Lg + c * HW + rowoff, x4, W);
            src = Rg + c * HW + rowoff;
          }
          float lv[4], tv[4], ev[4];
          unpack(lv4, lv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = w0[k] * src[o0[k]];
            if constexpr (SAMPLED) t += w1[k] * src[o1[k]];
            tv[k] = t;
            ev[k] = lv[k] - t;
          }
          float* pl = plane0 + static_cast<size_t>(g * GRP + c) * cstride + rowoff;
          if constexpr (SAMPLED) {
            if (!s.omit_ref) st4<VEC>(pl, x4, W, lv4);                             // reference half
            if (s.omit_ref < 2) st4<VEC>(pl + static_cast<size_t>(s.tch) * cstride, x4, W, pack(tv));  // warped half
          } else {
            st4<VEC>(pl, x4, W, make_float4(-ev[0] * ev[0], -ev[1] * ev[1], -ev[2] * ev[2], -ev[3] * ev[3]));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) g0[k] += ev[k] * ev[k];
          s1[c][0] += ev[0] + ev[1];
          s1[c][1] += ev[2] + ev[3];
        }
        st4<VEC>(plane0 + static_cast<size_t>(s.mainC + g) * cstride + rowoff, x4, W,
                 make_float4(-g0[0], -g0[1], -g0[2], -g0[3]));
      }
      if (r & 1) {   // a row pair is complete: emit its 2x2 cells, fold it into the 4x4 sums
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < GRP; ++c) {
          const float a = s1[c][0] * 0.25f, e = s1[c][1] * 0.25f;
          a0 += a * a;
          a1 += e * e;
          s2[c] += s1[c][0] + s1[c][1];
          s1[c][0] = s1[c][1] = 0.f;
        }
        const int py = 2 * by + (r >> 1);
        if (live && s.scales > 1 && py < s.H1) {
          float* prow = P1 + (pbase * s.H1 + py) * s.W1;
          if (2 * bx < s.W1) prow[2 * bx] = -a0;
          if (2 * bx + 1 < s.W1) prow[2 * bx + 1] = -a1;
        }
      }
    }
    if (live && s.scales > 2 && by < s.H2 && bx < s.W2) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < GRP; ++c) {
        const float m = s2[c] * 0.0625f;
        acc += m * m;
      }
      P2[(pbase * s.H2 + by) * s.W2 + bx] = -acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The fast path (all shipped configs): rows narrow enough for LDS staging, one item per thread.
//
// LDS layout
//   R4  [half h][row r][x & 3][x >> 2] of float4 = the 4 channels 4h..4h+3 of one right pixel.
//       One tap of one pixel is then TWO ds_read_b128 for all 8 channels (instead of 8 ds_read_b32),
//       and lanes that own consecutive 4-pixel blocks read consecutive 16-byte slots (conflict-free).
//   Lx  [channel c][row & 1][x] plain rows of the left map, read back as one float4 per channel;
//       rows 2,3 ride in registers through the first half and are swapped in at half time, which
//       keeps the workgroup at ~47 KiB (W=240) so that three of them share a CU.
// Addressing: every output plane base is wave-uniform (SGPR); a lane carries ONE 32-bit element
// offset (candidate, row, column) for all 17 planes it stores to.
// ------------------------------------------------------------------------------------------------

// ---- buffer (SRSRC) addressing: wave-uniform base in SGPRs + one 32-bit per-lane byte offset ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// 4 floats at element offset `eoff` (+ uniform element offset `uoff`); elements with x+k >= W are
// skipped in the non-VEC form.
template <bool VEC>
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, unsigned eoff, unsigned uoff, int x, int W, float4 v) {
  if constexpr (VEC) {
    u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
    // Store and fence are ONE asm block so that nothing can be scheduled between them: the registers of `u` are
    // only free for reuse after the block (see store_data_fence -- hipcc had moved a v_mul that overwrites the
    // first data register in front of the separate fence: element .x of one channel came out corrupted, and only
    // for some batch sizes / candidate counts).  LLVM assumes a buffer store with an SGPR soffset has no
    // store-data hazard at all and pads nothing here.  (The fence is free: a build without it times the same, 47.2 us.)
    const unsigned vo = eoff * 4u, so = uoff * 4u;
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1\n\ts_waitcnt expcnt(0)"
                 : : "v"(u), "v"(vo), "s"(r), "s"(so) : "memory");
  } else {
    if (x + 0 < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.x), r, eoff * 4u, uoff * 4u, 0);
    if (x + 1 < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.y), r, eoff * 4u + 4u, uoff * 4u, 0);
    if (x + 2 < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.z), r, eoff * 4u + 8u, uoff * 4u, 0);
    if (x + 3 < W) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v.w), r, eoff * 4u + 12u, uoff * 4u, 0);
  }
}

template <bool VEC>
__device__ __forceinline__ float4 bld4(__amdgpu_buffer_rsrc_t r, unsigned eoff, int x, int W) {
  if constexpr (VEC) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, eoff * 4u, 0, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  } else {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x + 0 < W) v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u, 0, 0));
    if (x + 1 < W) v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u + 4u, 0, 0));
    if (x + 2 < W) v.z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u + 8u, 0, 0));
    if (x + 3 < W) v.w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, eoff * 4u + 12u, 0, 0));
    return v;
  }
}

// One output pixel's two source columns as float4-slot offsets inside an R4 row, packed 16+16 bits,
// plus the fraction.  Columns outside the row map to the zero slot (index Wq of quarter-row 0), so
// zeros padding needs no weight masking.
template <bool SAMPLED>
__device__ __forceinline__ void tap4(int x, int d, float dispv, int W, float Wm1, int Wq, int Wqp,
                                     unsigned& packed, float& f) {
  int xi;
  f = 0.f;
  if constexpr (SAMPLED) {
    // same float sequence as the reference: normalise to [-1,1] (inverse_warp_3d.py:41-47)
    // and back (grid_sampler align_corners), so the tap position rounds identically
    const float xs = static_cast<float>(x) + (-dispv);
    const float gx = (xs / Wm1 * 2.f) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * Wm1;
    ix = fminf(fmaxf(ix, -2.f), static_cast<float>(W) + 1.f);   // keeps int conversion defined
    const float fl = floorf(ix);
    f = ix - fl;
    xi = static_cast<int>(fl);
  } else {
    xi = x - d;
  }
  const unsigned a0 = (xi >= 0 && xi < W) ? static_cast<unsigned>((xi & 3) * Wqp + (xi >> 2)) : static_cast<unsigned>(Wq);
  const int xj = xi + 1;
  const unsigned a1 = (xj >= 0 && xj < W) ? static_cast<unsigned>((xj & 3) * Wqp + (xj >> 2)) : static_cast<unsigned>(Wq);
  packed = a0 | (a1 << 16);
}

__device__ __forceinline__ float comp(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

template <bool SAMPLED, bool VEC, int NP, bool REF, bool MAIN = true>   // REF: write the reference (left-repeat) half; MAIN: write the warped half / the int path's main channels
// VEC: <= 128 VGPRs, three 5-wave workgroups per CU; ragged widths (scalar loads / stores with their own masks) get 168
// registers instead of spilling
__global__ void __launch_bounds__(512, VEC ? 4 : 3)
block_cost_fast(const float* __restrict__ L, const float* __restrict__ R,
                const float* __restrict__ disp, float* __restrict__ out,
                float* __restrict__ P1, float* __restrict__ P2, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const unsigned HW = static_cast<unsigned>(H) * W;
  const float* Lg = L + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Rg = R + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const int Wq = s.Wq, Wqp = s.Wqp, Wl = 4 * s.Wq;
  float4* ldsR4 = reinterpret_cast<float4*>(lds);                       // [2][TR][4][Wqp]
  // left rows: two at a time with rows 2,3 carried in registers (NP <= 3: what every shipped geometry uses), or --
  // few candidates on wide rows, where a thread would carry 4-8 float4 and spill -- all four rows in LDS
  constexpr bool CARRY = NP <= 3;
  constexpr int LROWS = CARRY ? 2 : 4;
  float* ldsL = lds + static_cast<size_t>(2) * TR * 4 * Wqp * 4;        // [GRP][LROWS][Wl]
  const int tid = threadIdx.x, nthr = blockDim.x;

  // ---- prologue: stage right rows (channel-packed), left rows 0,1; fetch left rows 2,3 ----------
  // Load order = need order: the left rows first (they go to LDS as they are and -- REF -- straight back out as the
  // D-fold repeat of the reference half, block_cost.py:51), then the right rows, whose transposition into the
  // channel-packed layout happens while those stores drain.  The reference half is 47 % of the sampled volume's
  // bytes and depends on nothing but `left`: streaming it from the staging threads puts it under the load latency
  // of the right rows instead of in the tap loop (vmcnt is in-order: the right-row loads are older than the stores,
  // so waiting for them does not wait for the stores).
  const unsigned dHW = static_cast<unsigned>(D) * HW;                         // one channel (uniform)
  // the whole [Ctot, D, H, W] slab of this batch item behind one descriptor (host checks < 4 GiB)
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(out + static_cast<size_t>(b) * s.Ctot * dHW,
                                                 static_cast<unsigned>(s.Ctot) * dHW * 4u);
  const int n2 = GRP * 2 * Wq;
  // Branch-free requests: a lane with nothing to fetch (past the item count / below the image) asks for an
  // out-of-range element and gets 0 back; a predicated load would be a branch with its own wait.
  constexpr unsigned OOR = 0x3ffffff0u;            // element offset whose byte offset (+12 for the scalar forms) is beyond any descriptor
  const __amdgpu_buffer_rsrc_t lrs = make_rsrc(Lg, static_cast<unsigned>(GRP) * HW * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(Rg, static_cast<unsigned>(GRP) * HW * 4u);
  float4 lfirst[NP], lpre[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = tid + p * nthr;
    const int j = i % Wq, cr = i / Wq;            // cr = c*2 + (row & 1)
    const int ya = y0 + (cr & 1), yb = ya + 2;
    const unsigned ea = static_cast<unsigned>(cr >> 1) * HW + static_cast<unsigned>(ya) * W + 4u * j;
    lfirst[p] = bld4<VEC>(lrs, (i < n2 && ya < H) ? ea : OOR, 4 * j, W);
    lpre[p] = bld4<VEC>(lrs, (i < n2 && yb < H) ? ea + 2u * W : OOR, 4 * j, W);
  }
  // right rows: the first pass of items is requested now (its data lands while the reference half is stored), the
  // others -- narrow workgroups only -- follow the classic load / transpose loop
  float rv[4][4];
  const int nR = 2 * TR * Wq;
  {
    const int j = tid % Wq, hr = tid / Wq;         // hr = h*TR + r
    const int r = hr & (TR - 1), h = hr >> 2;
    const int y = y0 + r;
    const unsigned er = static_cast<unsigned>(h * 4) * HW + static_cast<unsigned>(y) * W + 4u * j;
    const bool ok = tid < nR && y < H;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) unpack(bld4<VEC>(rrs, ok ? er + cc * HW : OOR, 4 * j, W), rv[cc]);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int i = tid + p * nthr;
    const int j = i % Wq, cr = i / Wq;
    if (i < n2) {
      *reinterpret_cast<float4*>(ldsL + ((cr >> 1) * LROWS + (cr & 1)) * Wl + 4 * j) = lfirst[p];
      if constexpr (!CARRY) *reinterpret_cast<float4*>(ldsL + ((cr >> 1) * LROWS + (cr & 1) + 2) * Wl + 4 * j) = lpre[p];
    }
    if constexpr (SAMPLED && REF) {
      const int ya = y0 + (cr & 1), yb = ya + 2;
      const unsigned e0 = static_cast<unsigned>(g * GRP + (cr >> 1)) * dHW + static_cast<unsigned>(ya) * W + 4u * j;
      const bool oka = i < n2 && ya < H, okb = i < n2 && yb < H;
      unsigned eA = oka ? e0 : OOR, eB = okb ? e0 + 2u * W : OOR;       // out of range = the store is dropped
      const unsigned sA = oka ? HW : 0u, sB = okb ? HW : 0u;
      for (int dd = 0; dd < D; ++dd, eA += sA, eB += sB) {
        bst4<VEC>(orsrc, eA, 0u, 4 * j, W, lfirst[p]);
        bst4<VEC>(orsrc, eB, 0u, 4 * j, W, lpre[p]);
      }
    }
  }
  if (tid < nR) {
    const int j = tid % Wq, hr = tid / Wq;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ldsR4[(hr * 4 + k) * Wqp + j] = make_float4(rv[0][k], rv[1][k], rv[2][k], rv[3][k]);
  }
  for (int i = tid + nthr; i < nR; i += nthr) {
    const int j = i % Wq, hr = i / Wq;
    const int r = hr & (TR - 1), h = hr >> 2;
    const int y = y0 + r;
    const unsigned er = static_cast<unsigned>(h * 4) * HW + static_cast<unsigned>(y) * W + 4u * j;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) unpack(bld4<VEC>(rrs, y < H ? er + cc * HW : OOR, 4 * j, W), rv[cc]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ldsR4[(hr * 4 + k) * Wqp + j] = make_float4(rv[0][k], rv[1][k], rv[2][k], rv[3][k]);
  }
  if (tid < 2 * TR) ldsR4[(tid * 4) * Wqp + Wq] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero slot of each row
  __syncthreads();

  // ---- one item (candidate d, 4x4 block bx) per thread ------------------------------------------
  const int item = tid;
  const int d = item / s.nbxp;
  const int bx = item - d * s.nbxp;
  const bool live = (d < D) && (bx < s.nbx);
  const int x4 = bx * 4;
  const float Wm1 = static_cast<float>(W - 1);
  const __amdgpu_buffer_rsrc_t drsrc = make_rsrc(SAMPLED ? disp + static_cast<size_t>(b) * dHW : out, dHW * 4u);
  unsigned loff = (live ? static_cast<unsigned>(d) : 0u) * HW + static_cast<unsigned>(y0) * W + x4;

  float s1[GRP][2], s2[GRP];
#pragma unroll
  for (int c = 0; c < GRP; ++c) s1[c][0] = s1[c][1] = s2[c] = 0.f;
  // pooled maps of this (b, g): one descriptor each, a lane carries one 32-bit element offset per map
  const size_t pplane = (static_cast<size_t>(b) * s.G + g) * D;
  const __amdgpu_buffer_rsrc_t p1rs = make_rsrc(P1 + pplane * s.H1 * s.W1, static_cast<unsigned>(D) * s.H1 * s.W1 * 4u);
  const __amdgpu_buffer_rsrc_t p2rs = make_rsrc(P2 + pplane * s.H2 * s.W2, static_cast<unsigned>(D) * s.H2 * s.W2 * 4u);
  const unsigned p1off = (live ? static_cast<unsigned>(d) : 0u) * s.H1 * s.W1 + 2u * bx;
  const unsigned p2off = (live ? static_cast<unsigned>(d) : 0u) * s.H2 * s.W2 + bx;

  float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (SAMPLED) {
    if (live) dnext = bld4<VEC>(drsrc, loff, x4, W);
  }

#pragma unroll 1
  for (int r = 0; r < TR; ++r, loff += W) {
    const int y = y0 + r;
    if constexpr (CARRY) {
      if (r == 2) {   // swap left rows 2,3 into the LDS slots of rows 0,1
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int i = tid + p * nthr;
          if (i < n2) *reinterpret_cast<float4*>(ldsL + (i / Wq) * Wl + 4 * (i % Wq)) = lpre[p];
        }
        __syncthreads();
      }
    }
    if (y < H && live) {
      float dv[4];
      unpack(dnext, dv);
      if constexpr (SAMPLED) {   // prefetch the next candidate row ahead of this row's stores
        if (r + 1 < TR && y + 1 < H) dnext = bld4<VEC>(drsrc, loff + W, x4, W);
      }
      unsigned op[4];
      float fr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) tap4<SAMPLED>(x4 + k, d, dv[k], W, Wm1, Wq, Wqp, op[k], fr[k]);
      float g0[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4* rrow = ldsR4 + (h * TR + r) * 4 * Wqp;
        float4 ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ta[k] = rrow[op[k] & 0xffffu];
          if constexpr (SAMPLED) tb[k] = rrow[op[k] >> 16];
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = h * 4 + cc;
          const float4 lv4 = *reinterpret_cast<const float4*>(ldsL + (c * LROWS + (CARRY ? (r & 1) : r)) * Wl + x4);
          float lv[4], tv[4], ev[4];
          unpack(lv4, lv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = comp(ta[k], cc);
            if constexpr (SAMPLED) t = (1.f - fr[k]) * t + fr[k] * comp(tb[k], cc);
            tv[k] = t;
            ev[k] = lv[k] - t;
          }
          const unsigned plane = static_cast<unsigned>(g * GRP + c) * dHW;        // uniform (SGPR)
          if constexpr (SAMPLED) {
            if constexpr (REF) {   // the reference half left with the staging threads (prologue)
              bst4<VEC>(orsrc, loff, plane + static_cast<unsigned>(C) * dHW, x4, W, pack(tv));    // warped half
            } else if constexpr (MAIN) {
              bst4<VEC>(orsrc, loff, plane, x4, W, pack(tv));                                     // warped half only
            }
          } else {
            bst4<VEC>(orsrc, loff, plane, x4, W,
                      make_float4(-ev[0] * ev[0], -ev[1] * ev[1], -ev[2] * ev[2], -ev[3] * ev[3]));
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) g0[k] += ev[k] * ev[k];
          s1[c][0] += ev[0] + ev[1];
          s1[c][1] += ev[2] + ev[3];
        }
      }
      bst4<VEC>(orsrc, loff, static_cast<unsigned>(s.mainC + g) * dHW, x4, W,
                make_float4(-g0[0], -g0[1], -g0[2], -g0[3]));
    }
    if (r & 1) {   // a row pair is complete: emit its 2x2 cells, fold it into the 4x4 sums
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int c = 0; c < GRP; ++c) {
        const float a = s1[c][0] * 0.25f, e = s1[c][1] * 0.25f;
        a0 += a * a;
        a1 += e * e;
        s2[c] += s1[c][0] + s1[c][1];
        s1[c][0] = s1[c][1] = 0.f;
      }
      const int py = 2 * by + (r >> 1);
      if (live && s.scales > 1 && py < s.H1) {
        const unsigned o = (p1off + static_cast<unsigned>(py) * s.W1) * 4u;
        if (2 * bx < s.W1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-a0), p1rs, o, 0, 0);
        if (2 * bx + 1 < s.W1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-a1), p1rs, o + 4u, 0, 0);
      }
    }
  }
  if (live && s.scales > 2 && by < s.H2 && bx < s.W2) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < GRP; ++c) {
      const float m = s2[c] * 0.0625f;
      acc += m * m;
    }
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-acc), p2rs, (p2off + static_cast<unsigned>(by) * s.W2) * 4u, 0, 0);
  }
}

// The correlation blocks alone (ts_block_cost_sampled_corr_fwd: what the pipeline launches since the first layer takes the
// warped half in pre-contracted form), one lane = one ROW of a 4x4 block for ALL candidates.
// block_cost_fast gives a lane one candidate and all four rows; with no main channels to store, what bounds it is LDS traffic and the
// staging prologue (round 5 ablation at the 1/4 level: 20.3 us, of which 11.5 before the first tap and 2.4 each for loads and stores).
// Here the eight left values of a lane's four pixels are loaded ONCE, straight into registers (no left rows in LDS, no half-time
// swap and its two barriers, a third less LDS read traffic), the candidates are a loop, and the four rows of a block are the four
// lanes of a quad: the 2x2 / 4x4 block means (block_cost.py:66-73) are two quad permutes (DPP, VALU rate) instead of registers
// carried over rows.  Sums are formed in the order block_cost_fast forms them (a pair of rows, then the two pairs).
// NRP: right-row staging items per lane (2 * TR * Wq float4-quads over the workgroup), all requested before the first is written.
template <int NRP, int NT = 256>      // NT: workgroup size bound (512: maps of 260-512 columns, e.g. KITTI's 1/4 level)
__global__ void __launch_bounds__(NT)
block_cost_corr_rows(const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
                     float* __restrict__ out, float* __restrict__ P1, float* __restrict__ P2, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const unsigned HW = static_cast<unsigned>(H) * W;
  const int Wq = s.Wq, Wqp = s.Wqp;
  float4* ldsR4 = reinterpret_cast<float4*>(lds);                       // [2][TR][4][Wqp]
  const int tid = threadIdx.x, nthr = blockDim.x;
  constexpr unsigned OOR = 0x3ffffff0u;
  const unsigned dHW = static_cast<unsigned>(D) * HW;
  const __amdgpu_buffer_rsrc_t lrs = make_rsrc(L + (static_cast<size_t>(b) * C + g * GRP) * HW, static_cast<unsigned>(GRP) * HW * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(R + (static_cast<size_t>(b) * C + g * GRP) * HW, static_cast<unsigned>(GRP) * HW * 4u);
  const __amdgpu_buffer_rsrc_t drs = make_rsrc(disp + static_cast<size_t>(b) * dHW, dHW * 4u);
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(out + static_cast<size_t>(b) * s.Ctot * dHW, static_cast<unsigned>(s.Ctot) * dHW * 4u);

  // this lane's row of this lane's block
  const int r = tid & 3, bx = tid >> 2;
  const int y = y0 + r, x4 = bx * 4;
  const bool live = bx < s.nbx && y < H;
  const unsigned poff = static_cast<unsigned>(y) * W + x4;               // element offset inside a plane

  // ---- requests, oldest first in the order they are needed: right rows (staging), left values, first candidates ----
  const int nR = 2 * TR * Wq;
  float4 rq[NRP][4];
#pragma unroll
  for (int p = 0; p < NRP; ++p) {
    const int i = tid + p * nthr;
    const int j = i % Wq, hr = i / Wq;              // hr = h*TR + row
    const int yy = y0 + (hr & (TR - 1)), h = hr >> 2;
    const unsigned er = static_cast<unsigned>(h * 4) * HW + static_cast<unsigned>(yy) * W + 4u * j;
    const bool ok = i < nR && yy < H;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) rq[p][cc] = bld4<true>(rrs, ok ? er + cc * HW : OOR, 4 * j, W);
  }
  float lv[GRP][4];
#pragma unroll
  for (int c = 0; c < GRP; ++c) unpack(bld4<true>(lrs, live ? static_cast<unsigned>(c) * HW + poff : OOR, x4, W), lv[c]);
  float4 dnext = bld4<true>(drs, live ? poff : OOR, x4, W);
#pragma unroll
  for (int p = 0; p < NRP; ++p) {
    const int i = tid + p * nthr;
    if (i < nR) {
      const int j = i % Wq, hr = i / Wq;
      float a[4][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) unpack(rq[p][cc], a[cc]);
#pragma unroll
      for (int k = 0; k < 4; ++k) ldsR4[(hr * 4 + k) * Wqp + j] = make_float4(a[0][k], a[1][k], a[2][k], a[3][k]);
    }
  }
  if (tid < 2 * TR) ldsR4[(tid * 4) * Wqp + Wq] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero slot of each row
  __syncthreads();

  const float Wm1 = static_cast<float>(W - 1);
  const size_t pplane = (static_cast<size_t>(b) * s.G + g) * D;
  const __amdgpu_buffer_rsrc_t p1rs = make_rsrc(P1 + pplane * s.H1 * s.W1, static_cast<unsigned>(D) * s.H1 * s.W1 * 4u);
  const __amdgpu_buffer_rsrc_t p2rs = make_rsrc(P2 + pplane * s.H2 * s.W2, static_cast<unsigned>(D) * s.H2 * s.W2 * 4u);
  const int py = 2 * by + (r >> 1);
  const bool w1 = bx < s.nbx && (r & 1) == 0 && s.scales > 1 && py < s.H1;
  const bool w2 = bx < s.nbx && r == 0 && s.scales > 2 && by < s.H2 && bx < s.W2;
  unsigned p1o = static_cast<unsigned>(py) * s.W1 + 2u * bx, p2o = static_cast<unsigned>(by) * s.W2 + bx;
  unsigned goff = static_cast<unsigned>(s.mainC + g) * dHW + poff;
  const float4* row0 = ldsR4 + r * 4 * Wqp;
  const float4* row1 = ldsR4 + (TR + r) * 4 * Wqp;
  const float m = live ? 1.f : 0.f;

#pragma unroll 1
  for (int d = 0; d < D; ++d, goff += HW, p1o += s.H1 * s.W1, p2o += s.H2 * s.W2) {
    float dv[4];
    unpack(dnext, dv);
    if (d + 1 < D) dnext = bld4<true>(drs, live ? static_cast<unsigned>(d + 1) * HW + poff : OOR, x4, W);
    unsigned op[4];
    float fr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tap4<true>(x4 + k, d, dv[k], W, Wm1, Wq, Wqp, op[k], fr[k]);
    float g0[4] = {0.f, 0.f, 0.f, 0.f};
    float sa[GRP], sb[GRP];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4* rrow = h ? row1 : row0;
      float4 ta[4], tb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ta[k] = rrow[op[k] & 0xffffu];
        tb[k] = rrow[op[k] >> 16];
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = h * 4 + cc;
        float ev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float t = (1.f - fr[k]) * comp(ta[k], cc) + fr[k] * comp(tb[k], cc);
          ev[k] = (lv[c][k] - t) * m;
          g0[k] += ev[k] * ev[k];
        }
        sa[c] = ev[0] + ev[1];
        sb[c] = ev[2] + ev[3];
      }
    }
    if (live) bst4<true>(orsrc, goff, 0u, x4, W, make_float4(-g0[0], -g0[1], -g0[2], -g0[3]));
    // the other row of the pair (lane ^ 1), then the other pair (lane ^ 2): quad permutes
    float a0 = 0.f, a1 = 0.f, acc = 0.f;
#pragma unroll
    for (int c = 0; c < GRP; ++c) {
      const float pa = sa[c] + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(sa[c]), 0xB1, 0xF, 0xF, true));
      const float pb = sb[c] + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(sb[c]), 0xB1, 0xF, 0xF, true));
      const float a = pa * 0.25f, e = pb * 0.25f;
      a0 += a * a;
      a1 += e * e;
      const float q = pa + pb;
      const float s2 = q + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(q), 0x4E, 0xF, 0xF, true));
      const float mm = s2 * 0.0625f;
      acc += mm * mm;
    }
    if (w1) {
      if (2 * bx < s.W1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-a0), p1rs, p1o * 4u, 0, 0);
      if (2 * bx + 1 < s.W1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-a1), p1rs, p1o * 4u + 4u, 0, 0);
    }
    if (w2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(-acc), p2rs, p2o * 4u, 0, 0);
  }
}

// trilinear(align_corners=True) expansion of the pooled maps (block_cost.py:74); the D axis maps
// to itself, so it is a per-candidate bilinear interpolation, done separably: a workgroup owns a
// band of RB output rows of one (b, g, d) plane, first interpolates the few pooled rows the band
// touches along W into LDS (coalesced reads of the tiny pooled maps), then blends pairs of those
// rows along H with ds_read_b128 and streams 16-byte stores.
template <bool VEC, int RB>
__global__ void __launch_bounds__(256)
block_cost_upsample(const float* __restrict__ P1, const float* __restrict__ P2,
                    float* __restrict__ out, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NR1 = RB / 2 + 2, NR2 = RB / 4 + 2;     // pooled rows a band can touch per level
  const int Wl = 4 * s.nbx;
  const int band = blockIdx.x, plane = blockIdx.y + blockIdx.z * gridDim.y;
  if (plane >= s.B * s.G * s.D) return;
  const int d = plane % s.D;
  const int bg = plane / s.D;
  const int g = bg % s.G, b = bg / s.G;
  const int yfirst = band * RB;
  const int ylast = min(yfirst + RB, s.H) - 1;
  const size_t HW = static_cast<size_t>(s.H) * s.W;

  int lo[3];
#pragma unroll
  for (int lvl = 1; lvl <= 2; ++lvl) {
    if (lvl >= s.scales) break;
    const float* P = (lvl == 1) ? P1 : P2;
    const int Hs = (lvl == 1) ? s.H1 : s.H2, Ws = (lvl == 1) ? s.W1 : s.W2;
    const float rh = (lvl == 1) ? s.rh1 : s.rh2, rw = (lvl == 1) ? s.rw1 : s.rw2;
    const int nr = (lvl == 1) ? NR1 : NR2;
    float* buf = lds + (lvl == 1 ? 0 : NR1 * Wl);
    lo[lvl] = static_cast<int>(rh * static_cast<float>(yfirst));
    const int hi = min(static_cast<int>(rh * static_cast<float>(ylast)) + 1, Hs - 1);
    const float* Pp = P + static_cast<size_t>(plane) * Hs * Ws;
    for (int i = threadIdx.x; i < nr * Wl; i += blockDim.x) {
      const int rr = i / Wl, x = i - rr * Wl;
      const int srow = lo[lvl] + rr;
      float v = 0.f;
      if (srow <= hi && x < s.W) {
        const float wr = rw * static_cast<float>(x);
        const int w1 = static_cast<int>(wr);
        const int wp = (w1 < Ws - 1) ? 1 : 0;
        const float wl = wr - static_cast<float>(w1);
        const float* row = Pp + static_cast<size_t>(srow) * Ws;
        v = (1.f - wl) * row[w1] + wl * row[w1 + wp];
      }
      buf[i] = v;
    }
  }
  __syncthreads();

  const int nrows = ylast - yfirst + 1;
  for (int item = threadIdx.x; item < nrows * s.nbx; item += blockDim.x) {
    const int r = item / s.nbx;
    const int x4 = (item - r * s.nbx) * 4;
    const int y = yfirst + r;
#pragma unroll
    for (int lvl = 1; lvl <= 2; ++lvl) {
      if (lvl >= s.scales) break;
      const int Hs = (lvl == 1) ? s.H1 : s.H2;
      const float rh = (lvl == 1) ? s.rh1 : s.rh2;
      const float* buf = lds + (lvl == 1 ? 0 : NR1 * Wl);
      const float hr = rh * static_cast<float>(y);
      const int h1 = static_cast<int>(hr);
      const int hp = (h1 < Hs - 1) ? 1 : 0;
      const float hl = hr - static_cast<float>(h1);
      const float4 a = *reinterpret_cast<const float4*>(buf + (h1 - lo[lvl]) * Wl + x4);
      const float4 c = *reinterpret_cast<const float4*>(buf + (h1 + hp - lo[lvl]) * Wl + x4);
      const float4 v = make_float4((1.f - hl) * a.x + hl * c.x, (1.f - hl) * a.y + hl * c.y,
                                   (1.f - hl) * a.z + hl * c.z, (1.f - hl) * a.w + hl * c.w);
      float* pl = out + ((static_cast<size_t>(b) * s.Ctot + s.mainC + lvl * s.G + g) * s.D + d) * HW +
                  static_cast<size_t>(y) * s.W;
      st4<VEC>(pl, x4, s.W, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Dense siblings of block_cost: cat_fms / dif_fms over ANY number of candidates
// (the literal shift-and-correlate over D = 48..192).  Same staging as the fast path -- TRD right rows of an
// 8-channel group in the channel-packed R4 layout, the left rows beside them -- but a workgroup walks the
// (candidate, 4-pixel block) items of its rows in strides of the block size instead of owning one each, and
// there is no pooling.  MODE 0: cat (left repeat | warped right);  1: max |left - warped| only (first pass of
// dif_fms: the fill value is the maximum over the WHOLE tensor, dif_fms.py:38);  2: dif with the fill applied;
// 3: the warp alone, sampled at x + disp (inverse_warp_3d itself, layers/inverse_warp_3d.py:4-58: its callers pass -disp).
// ------------------------------------------------------------------------------------------------
constexpr int TRD = 2;

template <int MODE, bool VEC>
__global__ void __launch_bounds__(256)
dense_warp_kernel(const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
                  float* __restrict__ out, unsigned* __restrict__ maxbits, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TRD;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const unsigned HW = static_cast<unsigned>(H) * W;
  const float* Lg = L + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Rg = R + (static_cast<size_t>(b) * C + g * GRP) * HW;
  const int Wq = s.Wq, Wqp = s.Wqp, Wl = 4 * s.Wq;
  float4* ldsR4 = reinterpret_cast<float4*>(lds);                       // [2][TRD][4][Wqp]
  float* ldsL = lds + static_cast<size_t>(2) * TRD * 4 * Wqp * 4;       // [GRP][TRD][Wl]
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < 2 * TRD * Wq; i += nthr) {
    const int jx = i % Wq, hr = i / Wq;           // hr = h*TRD + r
    const int r = hr % TRD, h = hr / TRD;
    const int y = min(y0 + r, H - 1);
    float v[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) unpack(ld4<VEC>(Rg + static_cast<size_t>(h * 4 + cc) * HW + static_cast<size_t>(y) * W, 4 * jx, W), v[cc]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      ldsR4[(hr * 4 + k) * Wqp + jx] = make_float4(v[0][k], v[1][k], v[2][k], v[3][k]);
  }
  if (tid < 2 * TRD) ldsR4[(tid * 4) * Wqp + Wq] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero slot of each row
  if constexpr (MODE != 3)
  for (int i = tid; i < GRP * TRD * Wq; i += nthr) {
    const int jx = i % Wq, cr = i / Wq;           // cr = c*TRD + r
    const int y = min(y0 + cr % TRD, H - 1);
    *reinterpret_cast<float4*>(ldsL + cr * Wl + 4 * jx) = ld4<VEC>(Lg + static_cast<size_t>(cr / TRD) * HW + static_cast<size_t>(y) * W, 4 * jx, W);
  }
  __syncthreads();

  const float Wm1 = static_cast<float>(W - 1);
  const unsigned dHW = static_cast<unsigned>(D) * HW;
  const int CO = (MODE == 0) ? 2 * C : C;
  const __amdgpu_buffer_rsrc_t orsrc = make_rsrc(out + static_cast<size_t>(b) * CO * dHW, static_cast<unsigned>(CO) * dHW * 4u);
  const __amdgpu_buffer_rsrc_t drsrc = make_rsrc(disp + static_cast<size_t>(b) * dHW, dHW * 4u);
  const float fillv = (MODE == 2) ? __uint_as_float(maxbits[0]) : 0.f;
  float vmax = 0.f;
  // one (candidate, 4-pixel block) item: taps, the 8 channels, stores
  auto do_item = [&](int r, int d, int x4, unsigned loff, const float4 dq) {
    float dv[4];
    unpack(dq, dv);
    if constexpr (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dv[k] = -dv[k];      // tap4 samples at x - disp (what block_cost / cat_fms / dif_fms ask of the warp)
    }
    unsigned op[4];
    float fr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tap4<true>(x4 + k, d, dv[k], W, Wm1, Wq, Wqp, op[k], fr[k]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4* rrow = ldsR4 + (h * TRD + r) * 4 * Wqp;
      float4 ta[4], tb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { ta[k] = rrow[op[k] & 0xffffu]; tb[k] = rrow[op[k] >> 16]; }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = h * 4 + cc;
        const float4 lv4 = MODE == 3 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(ldsL + (c * TRD + r) * Wl + x4);
        float lv[4], tv[4];
        unpack(lv4, lv);
#pragma unroll
        for (int k = 0; k < 4; ++k) tv[k] = (1.f - fr[k]) * comp(ta[k], cc) + fr[k] * comp(tb[k], cc);
        const unsigned plane = static_cast<unsigned>(g * GRP + c) * dHW;        // uniform (SGPR)
        if constexpr (MODE == 0) {
          bst4<VEC>(orsrc, loff, plane, x4, W, lv4);
          bst4<VEC>(orsrc, loff, plane + static_cast<unsigned>(C) * dHW, x4, W, pack(tv));
        } else if constexpr (MODE == 3) {
          bst4<VEC>(orsrc, loff, plane, x4, W, pack(tv));
        } else if constexpr (MODE == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (VEC || x4 + k < W) vmax = fmaxf(vmax, fabsf(lv[k] - tv[k]));
        } else {
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = (tv[k] > 0.f) ? fabsf(lv[k] - tv[k]) : fillv;   // dif_fms.py:40-42
          bst4<VEC>(orsrc, loff, plane, x4, W, pack(o));
        }
      }
    }
  };
  if constexpr (VEC) {
    // Run order (round 4).  The rows y0, y0 + 1 of one (plane, candidate) are ONE contiguous run of 2 W floats, and with W = 240 a row is 7.5
    // cache lines: walking row by row (below) writes the shared line of every run twice, at different times -- stores alone in that
    // order reach 3.9-4.3 TB/s on 1.66 GB where the same bytes written run by run reach 5.9 (tools/exp/dense_store_patterns.py).  So an
    // item is (candidate, float4 of the RUN), consecutive lanes = consecutive float4 of a run.  The candidates' disparities come through LDS
    // in chunks of DC candidates, the next chunk's loads issued BEFORE this chunk's stores: vmcnt counts loads and stores in order, and a
    // load requested per item (the row-order loop) makes every iteration wait for the previous iteration's stores to complete (-25 %:
    // the row-order kernel with its taps and LDS reads removed ran exactly as long as with them).
    const int rows = min(TRD, H - y0), run4 = rows * s.nbx;
    const int DC = max(1, min(8, (4 * 256) / run4));
    float4* ldsD = reinterpret_cast<float4*>(ldsL + GRP * TRD * Wl);          // [DC * run4 <= 1024]
    const unsigned runbase = static_cast<unsigned>(y0) * W;
    float4 dreg[4];
    auto fetch = [&](int d0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = tid + 256 * q;
        const int dd = i / run4, e = i - dd * run4;
        const bool ok = dd < DC && d0 + dd < D;
        dreg[q] = bld4<true>(drsrc, ok ? static_cast<unsigned>(d0 + dd) * HW + runbase + 4u * e : 0x3ffffff0u, 0, W);
      }
    };
    fetch(0);
    for (int d0 = 0; d0 < D; d0 += DC) {
      __syncthreads();                       // the previous chunk's candidates are consumed
#pragma unroll
      for (int q = 0; q < 4; ++q) ldsD[tid + 256 * q] = dreg[q];
      __syncthreads();
      if (d0 + DC < D) fetch(d0 + DC);
      const int nd = min(DC, D - d0);
      for (int i = tid; i < nd * run4; i += 256) {
        const int dd = i / run4, e = i - dd * run4;
        const int r = e / s.nbx, x4 = (e - r * s.nbx) * 4;
        const int d = d0 + dd;
        do_item(r, d, x4, static_cast<unsigned>(d) * HW + runbase + 4u * e, ldsD[i]);
      }
    }
  } else {
  const int nitems = s.nbx * D;
  for (int r = 0; r < TRD; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    // the candidate row of the NEXT item is requested before this item's taps and stores
    auto item_off = [&](int item, int& d, int& x4) {
      d = item / s.nbx;
      x4 = (item - d * s.nbx) * 4;
      return static_cast<unsigned>(d) * HW + static_cast<unsigned>(y) * W + x4;
    };
    int dn = 0, xn = 0;
    unsigned offn = tid < nitems ? item_off(tid, dn, xn) : 0u;
    float4 dnext = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < nitems) dnext = bld4<VEC>(drsrc, offn, xn, W);
    for (int item = tid; item < nitems; item += nthr) {
      const int d = dn, x4 = xn;
      const unsigned loff = offn;
      const float4 dq = dnext;
      if (item + nthr < nitems) {
        offn = item_off(item + nthr, dn, xn);
        dnext = bld4<VEC>(drsrc, offn, xn, W);
      }
      do_item(r, d, x4, loff, dq);
    }
  }
  }
  if constexpr (MODE == 1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((tid & 63) == 0) atomicMax(maxbits, __float_as_uint(vmax));      // non-negative floats order like their bits
  }
}

// Direct form of the same expansion: one lane per float4 of output of BOTH levels, no LDS, no barrier.
// The pooled maps are tiny (a plane of level 1 is H/2 x W/2) and live in L2, four output pixels touch at
// most 4 (level 1) / 3 (level 2) pooled cells per row, so a lane issues 14 independent clamped loads up
// front and two 16-byte stores: a single memory round trip instead of load -> LDS -> barrier -> blend.
template <bool VEC>
__global__ void __launch_bounds__(256)
block_cost_upsample_direct(const float* __restrict__ P1, const float* __restrict__ P2,
                           float* __restrict__ out, const Shape s) {
  constexpr int RPT = 2;                      // output rows per lane: halves the wave count (one resident round at config-2 sizes)
  const int plane = blockIdx.y;
  const int d = plane % s.D;
  const int bg = plane / s.D;
  const int g = bg % s.G, b = bg / s.G;
  const size_t HW = static_cast<size_t>(s.H) * s.W;
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = (s.H + RPT - 1) / RPT;
  if (item >= rows * s.nbx) return;
  const int yb = (item / s.nbx) * RPT;
  const int x4 = (item - (item / s.nbx) * s.nbx) * 4;
  float cell[RPT][3][2][4];                   // [row][level][pooled row][cell]
  int c0[3], hpv[RPT][3];
  float hlv[RPT][3];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int y = min(yb + q, s.H - 1);
#pragma unroll
    for (int lvl = 1; lvl <= 2; ++lvl) {
      const bool on = lvl < s.scales;
      const float* P = (lvl == 1) ? P1 : P2;
      const int Hs = (lvl == 1) ? s.H1 : s.H2, Ws = (lvl == 1) ? s.W1 : s.W2;
      const float rh = (lvl == 1) ? s.rh1 : s.rh2, rw = (lvl == 1) ? s.rw1 : s.rw2;
      const float hr = rh * static_cast<float>(y);
      const int h1 = min(static_cast<int>(hr), Hs - 1);
      hpv[q][lvl] = (h1 < Hs - 1) ? 1 : 0;
      hlv[q][lvl] = hr - static_cast<float>(h1);
      c0[lvl] = min(static_cast<int>(rw * static_cast<float>(x4)), Ws - 1);
      const float* Pp = (on ? P : P1) + static_cast<size_t>(on ? plane : 0) * Hs * Ws;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          cell[q][lvl][r][c] = Pp[static_cast<size_t>(h1 + (r ? hpv[q][lvl] : 0)) * Ws + min(c0[lvl] + c, Ws - 1)];
    }
  }
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int y = yb + q;
    if (y >= s.H) break;
#pragma unroll
    for (int lvl = 1; lvl <= 2; ++lvl) {
      if (lvl >= s.scales) break;
      const int Ws = (lvl == 1) ? s.W1 : s.W2;
      const float rw = (lvl == 1) ? s.rw1 : s.rw2;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float wr = rw * static_cast<float>(min(x4 + k, s.W - 1));
        const int w1 = static_cast<int>(wr);
        const int wp = (w1 < Ws - 1) ? 1 : 0;
        const float wl = wr - static_cast<float>(w1);
        const int i0 = w1 - c0[lvl], i1 = i0 + wp;        // 0..3 (pooled width is at most half the output's)
        float a0 = cell[q][lvl][0][0], a1 = a0, b0 = cell[q][lvl][1][0], b1 = b0;
#pragma unroll
        for (int c = 1; c < 4; ++c) {
          a0 = (i0 == c) ? cell[q][lvl][0][c] : a0; a1 = (i1 == c) ? cell[q][lvl][0][c] : a1;
          b0 = (i0 == c) ? cell[q][lvl][1][c] : b0; b1 = (i1 == c) ? cell[q][lvl][1][c] : b1;
        }
        const float top = (1.f - wl) * a0 + wl * a1;      // along W first, then along H (the staged kernel's order)
        const float bot = (1.f - wl) * b0 + wl * b1;
        v[k] = (1.f - hlv[q][lvl]) * top + hlv[q][lvl] * bot;
      }
      float* pl = out + ((static_cast<size_t>(b) * s.Ctot + s.mainC + lvl * s.G + g) * s.D + d) * HW +
                  static_cast<size_t>(y) * s.W;
      st4<VEC>(pl, x4, s.W, make_float4(v[0], v[1], v[2], v[3]));
    }
  }
}

// The same expansion with RB output rows per lane: the pooled rows a run of RB output rows touches (at most RB/2+2 of
// level 1, RB/4+2 of level 2) are fetched ONCE and interpolated along W once per pooled row instead of twice per output row
// (an eighth of the loads and of the W-lerps per stored float4 at RB = 8); per output row what is left is the choice of its two
// pooled rows (compare-selects: no dynamic register indexing) and the H-lerp.  Same arithmetic order, bit-identical values.
template <bool VEC, int RB>
__global__ void __launch_bounds__(256)
block_cost_upsample_rows(const float* __restrict__ P1, const float* __restrict__ P2, float* __restrict__ out, const Shape s) {
  constexpr int NR1 = RB / 2 + 2, NR2 = RB / 4 + 2;
  const int plane = blockIdx.y;
  const int d = plane % s.D;
  const int bg = plane / s.D;
  const int g = bg % s.G, b = bg / s.G;
  const size_t HW = static_cast<size_t>(s.H) * s.W;
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  const int runs = (s.H + RB - 1) / RB;
  if (item >= runs * s.nbx) return;
  const int yb = (item / s.nbx) * RB;
  const int x4 = (item - (item / s.nbx) * s.nbx) * 4;
#pragma unroll
  for (int lvl = 1; lvl <= 2; ++lvl) {
    if (lvl >= s.scales) break;
    constexpr int NRmax = NR1;
    const int NR = (lvl == 1) ? NR1 : NR2;
    const float* P = (lvl == 1) ? P1 : P2;
    const int Hs = (lvl == 1) ? s.H1 : s.H2, Ws = (lvl == 1) ? s.W1 : s.W2;
    const float rh = (lvl == 1) ? s.rh1 : s.rh2, rw = (lvl == 1) ? s.rw1 : s.rw2;
    const float* Pp = P + static_cast<size_t>(plane) * Hs * Ws;
    const int hlo = min(static_cast<int>(rh * static_cast<float>(yb)), Hs - 1);
    // the four pooled cells a float4 of output can touch are contiguous: ONE 16-byte load per pooled row (dword-aligned buffer
    // load; the window is shifted left at the right border instead of clamping cell by cell) -- 7 loads per lane instead of 28
    // single cells, which at ~29 cycles of the texture addresser each were 3 of the kernel's 7.6 us
    const int c0 = max(min(static_cast<int>(rw * static_cast<float>(x4)), Ws - 4), 0);
    float cell[NRmax][4];
    if (Ws >= 4) {
      const __amdgpu_buffer_rsrc_t prs = make_rsrc(Pp, static_cast<unsigned>(Hs) * Ws * 4u);
#pragma unroll
      for (int r = 0; r < NRmax; ++r) {
        if (r < NR) {
          const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(prs, (static_cast<unsigned>(min(hlo + r, Hs - 1)) * Ws + c0) * 4u, 0, 0);
          cell[r][0] = __uint_as_float(u.x); cell[r][1] = __uint_as_float(u.y); cell[r][2] = __uint_as_float(u.z); cell[r][3] = __uint_as_float(u.w);
        } else {
          cell[r][0] = cell[r][1] = cell[r][2] = cell[r][3] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < NRmax; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          cell[r][c] = (r < NR) ? Pp[static_cast<size_t>(min(hlo + r, Hs - 1)) * Ws + min(c0 + c, Ws - 1)] : 0.f;
    }
    // along W, once per pooled row
    float rowv[NRmax][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float wr = rw * static_cast<float>(min(x4 + k, s.W - 1));
      const int w1 = static_cast<int>(wr);
      const int wp = (w1 < Ws - 1) ? 1 : 0;
      const float wl = wr - static_cast<float>(w1);
      const int i0 = w1 - c0, i1 = i0 + wp;
#pragma unroll
      for (int r = 0; r < NRmax; ++r) {
        float a0 = cell[r][0], a1 = a0;
#pragma unroll
        for (int c = 1; c < 4; ++c) { a0 = (i0 == c) ? cell[r][c] : a0; a1 = (i1 == c) ? cell[r][c] : a1; }
        rowv[r][k] = (1.f - wl) * a0 + wl * a1;
      }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int y = yb + q;
      if (y >= s.H) break;
      const float hr = rh * static_cast<float>(y);
      const int h1 = min(static_cast<int>(hr), Hs - 1);
      const int hp = (h1 < Hs - 1) ? 1 : 0;
      const float hl = hr - static_cast<float>(h1);
      const int i0 = h1 - hlo, i1 = i0 + hp;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float top = rowv[0][k], bot = top;
#pragma unroll
        for (int r = 1; r < NRmax; ++r) { top = (i0 == r) ? rowv[r][k] : top; bot = (i1 == r) ? rowv[r][k] : bot; }
        v[k] = (1.f - hl) * top + hl * bot;
      }
      float* pl = out + ((static_cast<size_t>(b) * s.Ctot + s.mainC + lvl * s.G + g) * s.D + d) * HW + static_cast<size_t>(y) * s.W;
      st4<VEC>(pl, x4, s.W, make_float4(v[0], v[1], v[2], v[3]));
    }
  }
}

int make_shape(Shape& s, bool sampled, int B, int C, int H, int W, int D, int scales, int omit_ref = 0, int min_hw = 4) {
  TS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && D > 0, TS_ERR_SHAPE, "block_cost: non-positive size");
  TS_REQUIRE(C % GRP == 0, TS_ERR_SHAPE, "block_cost: C=%d is not a multiple of 8 (block_cost.py:9)", C);
  TS_REQUIRE(scales >= 1 && scales <= 3, TS_ERR_UNSUPPORTED, "block_cost: scales=%d outside 1..3", scales);
  TS_REQUIRE(H >= min_hw && W >= min_hw, TS_ERR_UNSUPPORTED, "block_cost: H,W must be >= %d (got %dx%d)", min_hw, H, W);
  // D == 1 (and H == 1, W == 1) divide by zero in the reference's coordinate normalisation (inverse_warp_3d.py:45-47): its output is
  // whatever grid_sample makes of NaN coordinates
  TS_REQUIRE(!sampled || D >= 2, TS_ERR_UNSUPPORTED, "block_cost: sampled path needs D >= 2");
  TS_REQUIRE(!sampled || (H >= 2 && W >= 2), TS_ERR_UNSUPPORTED, "block_cost: sampled path needs H, W >= 2 (got %dx%d)", H, W);
  TS_REQUIRE(B <= 65535 && C / GRP <= 65535, TS_ERR_UNSUPPORTED, "block_cost: grid too large");
  s.B = B; s.C = C; s.H = H; s.W = W; s.D = D; s.scales = scales;
  s.G = C / GRP;
  s.omit_ref = sampled ? omit_ref : 0;
  s.tch = s.omit_ref ? 0 : C;
  s.mainC = (sampled && !omit_ref) ? 2 * C : (s.omit_ref == 2 ? 0 : C);
  s.Ctot = s.mainC + scales * s.G;
  s.H1 = H / 2; s.W1 = W / 2; s.H2 = H / 4; s.W2 = W / 4;
  s.nbx = (W + 3) / 4; s.nby = (H + 3) / 4;
  s.Wq = s.nbx; s.Wqp = (s.Wq + 1) | 1;    // >= Wq + 1: slot Wq of quarter-row 0 is the zero slot
  // lanes of one wave should store into one plane: pad the per-candidate item count to a divisor
  // (or multiple) of 64 when that idles at most 1/8 of the lanes
  s.nbxp = s.nbx;
  for (int q : {8, 16, 32, 64, 128, 192, 256, 320, 384, 448, 512}) {
    if (q >= s.nbx) {
      if ((q - s.nbx) * 8 <= q) s.nbxp = q;
      break;
    }
  }
  auto scale = [](int in, int outn) { return outn > 1 ? static_cast<float>(in - 1) / static_cast<float>(outn - 1) : 0.f; };
  s.rh1 = scale(s.H1, H); s.rw1 = scale(s.W1, W);
  s.rh2 = scale(s.H2, H); s.rw2 = scale(s.W2, W);
  return TS_OK;
}

size_t pooled_bytes(const Shape& s, int lvl) {
  const size_t n = static_cast<size_t>(s.B) * s.G * s.D * (lvl == 1 ? s.H1 * s.W1 : s.H2 * s.W2);
  return ts::round_up(n * sizeof(float), 256);
}

template <bool SAMPLED>
int launch_fwd(const float* left, const float* right, const float* disp, float* out, void* workspace,
               int B, int C, int H, int W, int D, int scales, void* stream, int omit_ref = 0) {
  Shape s;
  if (int rc = make_shape(s, SAMPLED, B, C, H, W, D, scales, omit_ref)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(out);
  if (SAMPLED) TS_REQUIRE_PTR(disp);
  if (scales > 1) TS_REQUIRE_PTR(workspace);
  float* P1 = reinterpret_cast<float*>(workspace);
  float* P2 = scales > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pooled_bytes(s, 1)) : nullptr;

  const bool vec = (W % 4 == 0) && ts::aligned16(left) && ts::aligned16(right) && ts::aligned16(out) &&
                   (!SAMPLED || ts::aligned16(disp));
  const int nitems = s.nbxp * D;
  // one pass when the row of items fits a workgroup, otherwise an even split over the fewest passes
  const int passes = (nitems + 511) / 512;
  int threads = static_cast<int>(ts::round_up(static_cast<size_t>((nitems + passes - 1) / passes), ts::kWave));
  if (threads > 512) threads = 512;
  // LDS: 4 right rows (de-interleaved, padded) + 2 left rows, per channel of the group
  const int np = (GRP * 2 * s.Wq + threads - 1) / threads;     // left float4s per thread and row pair
  // the NP the dispatch below INSTANTIATES (ragged widths have no NP = 3 form: np = 3 runs as NP = 4, which keeps all four left rows in
  // LDS -- sizing that case for two rows put rows 2, 3 outside the allocation: every correlation plane wrong for e.g. W = 38, D = 5)
  const int np_inst = np <= 2 ? 2 : ((vec && np == 3) ? 3 : (np <= 4 ? 4 : 8));
  const size_t lds_bytes = static_cast<size_t>(GRP) * (TR * 4 * s.Wqp + (np_inst <= 3 ? 2 : 4) * 4 * s.Wq) * sizeof(float);   // R4 + Lx
  // the fast path addresses one batch item's output slab through a 32-bit buffer descriptor
  const bool stage = lds_bytes <= 64 * 1024 && passes == 1 && np <= 8 &&
                     static_cast<unsigned long long>(s.Ctot) * D * H * W * 4ull < 0xffffff00ull;
  const dim3 grid(s.nby, s.G, B);
  hipStream_t st = ts::as_stream(stream);

  // correlation blocks alone on aligned maps: one lane per block row, candidates as a loop (block_cost_corr_rows)
  static const bool corr_rows = [] { const char* e = getenv("TS_K1_CORR_ROWS"); return !e || atoi(e) != 0; }();
  bool done = false;
  if (SAMPLED && omit_ref == 2 && vec && corr_rows && static_cast<unsigned long long>(s.Ctot) * D * H * W * 4ull < 0xffffff00ull) {
    const int cthreads = static_cast<int>(ts::round_up(static_cast<size_t>(s.nbx) * 4, ts::kWave));
    const int nrp = cthreads <= 1024 ? (2 * TR * s.Wq + cthreads - 1) / cthreads : 99;
    const size_t clds = static_cast<size_t>(2) * TR * 4 * s.Wqp * 4 * sizeof(float);
    if (cthreads <= 256 && nrp <= 2 && clds <= 64 * 1024) {
      if (nrp == 1) hipLaunchKernelGGL((block_cost_corr_rows<1>), grid, dim3(cthreads), clds, st, left, right, disp, out, P1, P2, s);
      else hipLaunchKernelGGL((block_cost_corr_rows<2>), grid, dim3(cthreads), clds, st, left, right, disp, out, P1, P2, s);
      done = true;
    } else if (cthreads <= 512 && nrp <= 2 && clds <= 64 * 1024) {
      if (nrp == 1) hipLaunchKernelGGL((block_cost_corr_rows<1, 512>), grid, dim3(cthreads), clds, st, left, right, disp, out, P1, P2, s);
      else hipLaunchKernelGGL((block_cost_corr_rows<2, 512>), grid, dim3(cthreads), clds, st, left, right, disp, out, P1, P2, s);
      done = true;
    }
  }
  if (done) {
  } else
#define TS_LAUNCH_FAST(V, N)                                                                        \
  do {                                                                                               \
    if (SAMPLED && omit_ref == 2)                                                                    \
      hipLaunchKernelGGL((block_cost_fast<SAMPLED, V, N, false, false>), grid, dim3(threads), lds_bytes, st, \
                         left, right, disp, out, P1, P2, s);                                         \
    else if (SAMPLED && omit_ref)                                                                    \
      hipLaunchKernelGGL((block_cost_fast<SAMPLED, V, N, false>), grid, dim3(threads), lds_bytes, st, \
                         left, right, disp, out, P1, P2, s);                                         \
    else                                                                                             \
      hipLaunchKernelGGL((block_cost_fast<SAMPLED, V, N, true>), grid, dim3(threads), lds_bytes, st, \
                         left, right, disp, out, P1, P2, s);                                         \
  } while (0)
#define TS_LAUNCH_WIDE(V)                                                                        \
  hipLaunchKernelGGL((block_cost_main<SAMPLED, V, false, 1>), grid, dim3(threads), 0, st,        \
                     left, right, disp, out, P1, P2, s)
  if (stage) {
    if (vec) { if (np <= 2) TS_LAUNCH_FAST(true, 2); else if (np == 3) TS_LAUNCH_FAST(true, 3); else if (np == 4) TS_LAUNCH_FAST(true, 4); else TS_LAUNCH_FAST(true, 8); }
    else { if (np <= 2) TS_LAUNCH_FAST(false, 2); else if (np <= 4) TS_LAUNCH_FAST(false, 4); else TS_LAUNCH_FAST(false, 8); }
  } else {
    if (vec) TS_LAUNCH_WIDE(true);
    else TS_LAUNCH_WIDE(false);
  }
#undef TS_LAUNCH_FAST
#undef TS_LAUNCH_WIDE
  if (int rc = ts::launched("block_cost_main")) return rc;

  // output rows per lane of the expansion: 4 (measured at the 1/4 level: 2 rows 9.9 us, 4 rows 7.6 us, 8 rows 8.9 us -- too few
  // lanes); TS_K1_UPSAMPLE_ROWS=2|8 for the A/B
  static const int up_rows = [] { const char* e = getenv("TS_K1_UPSAMPLE_ROWS"); return e ? atoi(e) : 4; }();
  if (scales > 1 && static_cast<long long>(B) * s.G * D <= 65535 && (up_rows == 4 || up_rows == 8)) {
    const int rb = up_rows;
    const dim3 dgrid((((H + rb - 1) / rb) * s.nbx + 255) / 256, B * s.G * D);
    if (rb == 8) {
      if (vec) hipLaunchKernelGGL((block_cost_upsample_rows<true, 8>), dgrid, dim3(256), 0, st, P1, P2, out, s);
      else hipLaunchKernelGGL((block_cost_upsample_rows<false, 8>), dgrid, dim3(256), 0, st, P1, P2, out, s);
    } else {
      if (vec) hipLaunchKernelGGL((block_cost_upsample_rows<true, 4>), dgrid, dim3(256), 0, st, P1, P2, out, s);
      else hipLaunchKernelGGL((block_cost_upsample_rows<false, 4>), dgrid, dim3(256), 0, st, P1, P2, out, s);
    }
    if (int rc = ts::launched("block_cost_upsample_rows")) return rc;
  } else if (scales > 1 && static_cast<long long>(B) * s.G * D <= 65535) {
    const dim3 dgrid((((H + 1) / 2) * s.nbx + 255) / 256, B * s.G * D);
    if (vec) hipLaunchKernelGGL(block_cost_upsample_direct<true>, dgrid, dim3(256), 0, st, P1, P2, out, s);
    else hipLaunchKernelGGL(block_cost_upsample_direct<false>, dgrid, dim3(256), 0, st, P1, P2, out, s);
    if (int rc = ts::launched("block_cost_upsample_direct")) return rc;
  } else if (scales > 1) {
    const int nplanes = B * s.G * D;
    const int gy = nplanes < 32768 ? nplanes : 32768;
    const int gz = (nplanes + gy - 1) / gy;
    const int Wl = 4 * s.nbx;
    // largest row band whose separable intermediate fits 64 KiB of LDS
    int rb = 16;
    while (rb > 4 && static_cast<size_t>(rb / 2 + 2 + rb / 4 + 2) * Wl * sizeof(float) > 64 * 1024) rb /= 2;
    const size_t ulds = static_cast<size_t>(rb / 2 + 2 + rb / 4 + 2) * Wl * sizeof(float);
    TS_REQUIRE(ulds <= 64 * 1024, TS_ERR_UNSUPPORTED, "block_cost: W=%d too wide for the upsample stage", W);
    const dim3 ugrid((H + rb - 1) / rb, gy, gz);
#define TS_LAUNCH_UP(V, RBV) \
  hipLaunchKernelGGL((block_cost_upsample<V, RBV>), ugrid, dim3(256), ulds, st, P1, P2, out, s)
    if (vec) { if (rb == 16) TS_LAUNCH_UP(true, 16); else if (rb == 8) TS_LAUNCH_UP(true, 8); else TS_LAUNCH_UP(true, 4); }
    else { if (rb == 16) TS_LAUNCH_UP(false, 16); else if (rb == 8) TS_LAUNCH_UP(false, 8); else TS_LAUNCH_UP(false, 4); }
#undef TS_LAUNCH_UP
    if (int rc = ts::launched("block_cost_upsample")) return rc;
  }
  return TS_OK;
}

// ------------------------------------------------------------------------------------------------
// Backward.
//   step 1  block_cost_upsample_adjoint: dP_l = U_l^T dOut(scale-l block)   (gather, deterministic)
//   step 2  block_cost_bwd_main: same decomposition as the forward; each lane owns a 4x4 block of
//           one candidate, recomputes the differences e = L - t, and pushes
//             ge = dLoss/de = -2 e dG0 - 2 mean2x2(e) dP1/4 - 2 mean4x4(e) dP2/16  (- 2 e dCost, int path)
//             gL += ge (+ dOut_L);  gt = -ge (+ dOut_T);  gR[x0], gR[x0+1] += w gt;
//             gDisp -= gt (R[x0+1] - R[x0])            (zeros padding: out-of-row taps read as 0)
//           with fp32 hardware atomics; gL/gR/gDisp are zero-filled by the host entry first.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
block_cost_upsample_adjoint(const float* __restrict__ dout, float* __restrict__ dP1, float* __restrict__ dP2,
                            const Shape s) {
  const long long n1 = static_cast<long long>(s.B) * s.G * s.D * s.H1 * s.W1;
  const long long n2 = (s.scales > 2) ? static_cast<long long>(s.B) * s.G * s.D * s.H2 * s.W2 : 0;
  const size_t HW = static_cast<size_t>(s.H) * s.W;
  for (long long idx = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; idx < n1 + n2;
       idx += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int lvl = (idx < n1) ? 1 : 2;
    long long t = (lvl == 1) ? idx : idx - n1;
    const int Hs = (lvl == 1) ? s.H1 : s.H2, Ws = (lvl == 1) ? s.W1 : s.W2;
    const float rh = (lvl == 1) ? s.rh1 : s.rh2, rw = (lvl == 1) ? s.rw1 : s.rw2;
    const int px = static_cast<int>(t % Ws); t /= Ws;
    const int py = static_cast<int>(t % Hs); t /= Hs;
    const int d = static_cast<int>(t % s.D);
    const long long bg = t / s.D;
    const int g = static_cast<int>(bg % s.G), b = static_cast<int>(bg / s.G);
    const float* plane = dout + ((static_cast<size_t>(b) * s.Ctot + s.mainC + lvl * s.G + g) * s.D + d) * HW;
    int ylo = 0, yhi = s.H - 1, xlo = 0, xhi = s.W - 1;
    if (rh > 0.f) {
      ylo = max(0, static_cast<int>(floorf((py - 1) / rh)) - 1);
      yhi = min(s.H - 1, static_cast<int>(ceilf((py + 1) / rh)) + 1);
    }
    if (rw > 0.f) {
      xlo = max(0, static_cast<int>(floorf((px - 1) / rw)) - 1);
      xhi = min(s.W - 1, static_cast<int>(ceilf((px + 1) / rw)) + 1);
    }
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      const float hr = rh * static_cast<float>(y);
      const int h1 = static_cast<int>(hr);
      const int hp = (h1 < Hs - 1) ? 1 : 0;
      const float hl = hr - static_cast<float>(h1);
      const float wy = (h1 == py ? 1.f - hl : 0.f) + (h1 + hp == py ? hl : 0.f);
      if (wy == 0.f) continue;
      float racc = 0.f;
      for (int x = xlo; x <= xhi; ++x) {
        const float wr = rw * static_cast<float>(x);
        const int w1 = static_cast<int>(wr);
        const int wp = (w1 < Ws - 1) ? 1 : 0;
        const float wl = wr - static_cast<float>(w1);
        const float wx = (w1 == px ? 1.f - wl : 0.f) + (w1 + wp == px ? wl : 0.f);
        racc += wx * plane[static_cast<size_t>(y) * s.W + x];
      }
      acc += wy * racc;
    }
    if (lvl == 1) dP1[idx] = acc;
    else dP2[idx - n1] = acc;
  }
}

// source column (clamped to [-2, W+1]) and fraction of one output pixel
template <bool SAMPLED>
__device__ __forceinline__ void source_column(int x, int d, float dispv, int W, float Wm1, int& xi, float& f) {
  if constexpr (SAMPLED) {
    const float xs = static_cast<float>(x) + (-dispv);
    const float gx = (xs / Wm1 * 2.f) - 1.f;
    float ix = ((gx + 1.f) / 2.f) * Wm1;
    ix = fminf(fmaxf(ix, -2.f), static_cast<float>(W) + 1.f);
    const float fl = floorf(ix);
    f = ix - fl;
    xi = static_cast<int>(fl);
  } else {
    xi = x - d;
    f = 0.f;
  }
}

// The same in two steps (round 5): the source POSITION of a pixel (the reference's normalise / un-normalise float sequence with its
// one true division) depends on (candidate, pixel) only -- block_cost_bwd_tile computes it once per item instead of twice per channel
// (32 divisions per thread and channel) and keeps it where it kept the candidate's disparity; column and fraction are a floor away.
__device__ __forceinline__ float source_position(int x, float dispv, int W, float Wm1) {
  const float xs = static_cast<float>(x) + (-dispv);
  const float gx = (xs / Wm1 * 2.f) - 1.f;
  const float ix = ((gx + 1.f) / 2.f) * Wm1;
  return fminf(fmaxf(ix, -2.f), static_cast<float>(W) + 1.f);
}
template <bool SAMPLED>
__device__ __forceinline__ void column_of(int x, int d, float pos, int& xi, float& f) {
  if constexpr (SAMPLED) {
    const float fl = floorf(pos);
    f = pos - fl;
    xi = static_cast<int>(fl);
  } else {
    xi = x - d;
    f = 0.f;
  }
}

template <bool SAMPLED, bool VEC, bool STAGE>
__global__ void __launch_bounds__(256)
block_cost_bwd_main(const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
                    const float* __restrict__ dout, const float* __restrict__ dP1, const float* __restrict__ dP2,
                    float* __restrict__ gL, float* __restrict__ gR, float* __restrict__ gD, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const size_t HW = static_cast<size_t>(H) * W;
  const size_t goff = (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Lg = L + goff;
  const float* Rg = R + goff;
  if constexpr (STAGE) stage_right_rows<VEC>(lds, Rg, y0, H, W, HW, s.Wq, s.Wqp);

  const float Wm1 = static_cast<float>(W - 1);
  const int nitems = s.nbx * D;
  for (int item = threadIdx.x; item < nitems; item += blockDim.x) {
    const int d = item / s.nbx;
    const int bx = item - d * s.nbx;
    const int x4 = bx * 4;
    const float* dplane0 = dout + (static_cast<size_t>(b) * s.Ctot * D + d) * HW;
    const size_t cstride = static_cast<size_t>(D) * HW;
    const size_t pbase = ((static_cast<size_t>(b) * s.G + g) * D + d);

    int xi[TR][4];
    float fr[TR][4], dg0[TR][4], gdacc[TR][4];
    float dp1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float dp2 = 0.f;
#pragma unroll
    for (int r = 0; r < TR; ++r) {
      const int y = min(y0 + r, H - 1);
      float dv[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SAMPLED)
        unpack(ld4<VEC>(disp + (static_cast<size_t>(b) * D + d) * HW + static_cast<size_t>(y) * W, x4, W), dv);
      unpack(ld4<VEC>(dplane0 + static_cast<size_t>(s.mainC + g) * cstride + static_cast<size_t>(y) * W, x4, W), dg0[r]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        source_column<SAMPLED>(x4 + k, d, dv[k], W, Wm1, xi[r][k], fr[r][k]);
        gdacc[r][k] = 0.f;
      }
    }
    if (s.scales > 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int py = 2 * by + i, px = 2 * bx + j;
          if (py < s.H1 && px < s.W1) dp1[i][j] = dP1[(pbase * s.H1 + py) * s.W1 + px];
        }
      if (s.scales > 2 && by < s.H2 && bx < s.W2) dp2 = dP2[(pbase * s.H2 + by) * s.W2 + bx];
    }

#pragma unroll 1
    for (int c = 0; c < GRP; ++c) {
      float e[TR][4], slope[TR][4];
      float m1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int y = y0 + r;
        const bool rowok = y < H;
        const size_t rowoff = static_cast<size_t>(min(y, H - 1)) * W;
        float lv[4];
        unpack(ld4<VEC>(Lg + c * HW + rowoff, x4, W), lv);
        const float* src;
        if constexpr (STAGE) src = lds + static_cast<size_t>(c * TR + r) * 4 * s.Wqp;
        else src = Rg + c * HW + rowoff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int a = xi[r][k];
          const int i0 = min(max(a, 0), W - 1), i1 = min(max(a + 1, 0), W - 1);
          const int q0 = STAGE ? (i0 & 3) * s.Wqp + (i0 >> 2) : i0;
          const int q1 = STAGE ? (i1 & 3) * s.Wqp + (i1 >> 2) : i1;
          const float r0 = (a >= 0 && a < W) ? src[q0] : 0.f;
          const float r1 = (SAMPLED && a + 1 >= 0 && a + 1 < W) ? src[q1] : 0.f;
          const float t = (1.f - fr[r][k]) * r0 + fr[r][k] * r1;
          const bool ok = rowok && (x4 + k < W);
          e[r][k] = ok ? (lv[k] - t) : 0.f;
          slope[r][k] = r1 - r0;
        }
        m1[r >> 1][0] += e[r][0] + e[r][1];
        m1[r >> 1][1] += e[r][2] + e[r][3];
      }
      const float m2 = (m1[0][0] + m1[0][1] + m1[1][0] + m1[1][1]) * 0.0625f;
      const size_t chan = static_cast<size_t>(g * GRP + c);
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int y = y0 + r;
        if (y < H) {
          const size_t rowoff = static_cast<size_t>(y) * W;
          float dmain[4], dwarp[4] = {0.f, 0.f, 0.f, 0.f};
          // omit_ref (the volume without its reference half, ts_block_cost_sampled_warped_*): no gradient arrives for it, the warped
          // half starts at channel 0 (s.tch)
          if (!s.omit_ref) unpack(ld4<VEC>(dplane0 + chan * cstride + rowoff, x4, W), dmain);
          else { dmain[0] = dmain[1] = dmain[2] = dmain[3] = 0.f; }
          if constexpr (SAMPLED) unpack(ld4<VEC>(dplane0 + (chan + s.tch) * cstride + rowoff, x4, W), dwarp);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = x4 + k;
            if (x < W) {
              float ge = -2.f * e[r][k] * dg0[r][k];
              ge -= 0.125f * m1[r >> 1][k >> 1] * dp1[r >> 1][k >> 1];   // 2 * (sum/4) / 4
              ge -= 0.125f * m2 * dp2;                                    // 2 * mean / 16
              float gl, gt;
              if constexpr (SAMPLED) {
                gl = ge + dmain[k];
                gt = dwarp[k] - ge;
              } else {
                ge -= 2.f * e[r][k] * dmain[k];
                gl = ge;
                gt = -ge;
              }
              float* grow = gR + goff + c * HW + rowoff;
              if (gL) unsafeAtomicAdd(gL + goff + c * HW + rowoff + x, gl);
              const int a = xi[r][k];
              if (gR) {
                if (a >= 0 && a < W) unsafeAtomicAdd(grow + a, (1.f - fr[r][k]) * gt);
                if (SAMPLED && a + 1 >= 0 && a + 1 < W) unsafeAtomicAdd(grow + a + 1, fr[r][k] * gt);
              }
              if constexpr (SAMPLED) gdacc[r][k] -= gt * slope[r][k];
            }
          }
        }
      }
    }
    if constexpr (SAMPLED) {
      if (gD) {
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (y0 + r < H && x4 + k < W)
              unsafeAtomicAdd(gD + (static_cast<size_t>(b) * D + d) * HW + static_cast<size_t>(y0 + r) * W + x4 + k,
                              gdacc[r][k]);
      }
    }
  }
}

// The same backward without global atomics on the feature gradients: a workgroup owns rows y0..y0+3 of its 8
// channels of gL AND of gR outright (the warp moves samples along x only), so the D candidates' contributions are
// summed in two LDS row tiles (ds_add_f32), one channel at a time, and leave as plain coalesced stores.  The
// atomic version issues 3 * C * D * H * W global atomics (62.7 M at the 1/4 level: 1.40 ms); this one only the
// 16-way sum of gDisp over the channel groups.  The LDS atomic rate (~3 cycles per lane) is what bounds it: with
// 48 atomics per thread and channel 0.52 ms; gL is therefore summed over the candidates by lane shuffles (the D
// items of a block sit in adjacent lanes) and stored plainly -- 32 atomics, 0.40 ms.
// One item (candidate, 4x4 block) per thread.
template <bool SAMPLED, bool VEC>
__global__ void __launch_bounds__(512, VEC ? 4 : 3)   // VEC: <= 128 VGPRs, three 5-wave workgroups per CU
block_cost_bwd_tile(const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
                    const float* __restrict__ dout, const float* __restrict__ dP1, const float* __restrict__ dP2,
                    float* __restrict__ gL, float* __restrict__ gR, float* __restrict__ gD, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const size_t HW = static_cast<size_t>(H) * W;
  const size_t goff = (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Lg = L + goff;
  const float* Rg = R + goff;
  const int Wl = 4 * s.Wq;
  float* accL = lds + static_cast<size_t>(GRP) * TR * 4 * s.Wqp;       // [TR][Wl]
  // gR in DOUBLE: an LDS float atomic add is 22x the cost of ds_add_f64 on gfx950 (block_cost_bwd_rows, tools/exp/lds_atomic_rate.hip)
  double* accR = reinterpret_cast<double*>(accL + TR * Wl);            // [TR][Wl] (8-byte aligned: every term above is a multiple of 4 floats)
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < TR * Wl; i += nthr) { accL[i] = 0.f; accR[i] = 0.0; }
  stage_right_rows<VEC>(lds, Rg, y0, H, W, HW, s.Wq, s.Wqp);           // ends with a barrier

  const float Wm1 = static_cast<float>(W - 1);
  // The D candidates of one 4x4 block sit in ADJACENT lanes (64 / D blocks per wave): their gL contributions are
  // summed by lane shuffles and written without atomics (the LDS atomic rate, ~3 cycles per lane, is what bounds
  // this kernel: 48 atomics per thread and channel cost 390 of its 518 us, these 16 of them 130 us).
  const int per_wave = 64 / D;
  const int lane = tid & 63, bxl = lane / D;
  const int dl = lane - bxl * D;
  const int bxw = (tid >> 6) * per_wave + bxl;
  const bool live = bxl < per_wave && bxw < s.nbx;
  const int d = live ? dl : 0;
  const int bx = live ? bxw : 0;
  const int x4 = bx * 4;
  const float* dplane0 = dout + (static_cast<size_t>(b) * s.Ctot * D + d) * HW;
  const size_t cstride = static_cast<size_t>(D) * HW;
  const size_t pbase = ((static_cast<size_t>(b) * s.G + g) * D + d);

  float dvs[TR][4], gdacc[TR][4];       // the items' tap positions (source_position), column / fraction re-derived per use (registers)
  float dp1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  float dp2 = 0.f;
#pragma unroll
  for (int r = 0; r < TR; ++r) {
    const int y = min(y0 + r, H - 1);
    float dv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SAMPLED)
      unpack(ld4<VEC>(disp + (static_cast<size_t>(b) * D + d) * HW + static_cast<size_t>(y) * W, x4, W), dv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dvs[r][k] = SAMPLED ? source_position(x4 + k, dv[k], W, Wm1) : 0.f;        // the tap POSITION from here on
      gdacc[r][k] = 0.f;
    }
  }
  if (live && s.scales > 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int py = 2 * by + i, px = 2 * bx + j;
        if (py < s.H1 && px < s.W1) dp1[i][j] = dP1[(pbase * s.H1 + py) * s.W1 + px];
      }
    if (s.scales > 2 && by < s.H2 && bx < s.W2) dp2 = dP2[(pbase * s.H2 + by) * s.W2 + bx];
  }
  auto lds_add = [](double* p, float v) { __hip_atomic_fetch_add(p, static_cast<double>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

#pragma unroll 1
  for (int c = 0; c < GRP; ++c) {
    {   // every lane runs this block (the shuffles below need whole waves); lanes without an item contribute zeros
      if constexpr (SAMPLED) {     // opaque to the optimiser: keeps the tap arithmetic inside the loop instead of
#pragma unroll                    // 32 loop-invariant registers (which spilled to scratch: 133 synchronous reloads)
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(dvs[r][q]));
      }
      // one pixel's difference e = L - warped R and the slope of R between its two taps; every LDS read is
      // unconditional at a clamped index and masked afterwards (a read behind a condition becomes a branch
      // around a wait: 118 of them in the first version of this kernel)
      auto diff = [&](int r, int kk, float lval, bool ok, float& slope, int& a, float& f) {
        column_of<SAMPLED>(x4 + kk, d, dvs[r][kk], a, f);
        const float* src = lds + static_cast<size_t>(c * TR + r) * 4 * s.Wqp;
        const int i0 = min(max(a, 0), W - 1), i1 = min(max(a + 1, 0), W - 1);
        // masked by multiplication: a select would be turned back into a load behind a branch
        const float r0 = src[(i0 & 3) * s.Wqp + (i0 >> 2)] * ((a >= 0 && a < W) ? 1.f : 0.f);
        const float r1 = SAMPLED ? src[(i1 & 3) * s.Wqp + (i1 >> 2)] * ((a + 1 >= 0 && a + 1 < W) ? 1.f : 0.f) : 0.f;
        slope = r1 - r0;
        const float t = (1.f - f) * r0 + f * r1;
        return ok ? (lval - t) : 0.f;
      };
      // pass A: the 2x2 / 4x4 block sums of e
      float m1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int y = y0 + r;
        float lv[4];
        unpack(ld4<VEC>(Lg + c * HW + static_cast<size_t>(min(y, H - 1)) * W, x4, W), lv);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          float slope, f;
          int a;
          m1[r >> 1][kk >> 1] += diff(r, kk, lv[kk], live && (y < H) && (VEC || x4 + kk < W), slope, a, f);
        }
      }
      const float m2 = (m1[0][0] + m1[0][1] + m1[1][0] + m1[1][1]) * 0.0625f;
      const size_t chan = static_cast<size_t>(g * GRP + c);
      if constexpr (SAMPLED) {     // ... and keeps pass B from inheriting 64 registers of pass A's intermediates
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(dvs[r][q]));
      }
      // pass B: e again (keeping e and the slope from pass A costs 32 registers: 68 B of scratch at 4 waves, 350 us;
      // 146 VGPRs at 3 waves, 395 us; recomputing: 337 us), the gradients, LDS accumulation
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int y = y0 + r;
        if (y < H) {                                                     // wave-uniform
          const size_t rowoff = static_cast<size_t>(y) * W;
          float lv[4], dg0r[4], dmain[4], dwarp[4] = {0.f, 0.f, 0.f, 0.f};
          unpack(ld4<VEC>(Lg + c * HW + rowoff, x4, W), lv);
          unpack(ld4<VEC>(dplane0 + static_cast<size_t>(s.mainC + g) * cstride + rowoff, x4, W), dg0r);
          // omit_ref (the volume without its reference half, ts_block_cost_sampled_warped_*): no gradient arrives for it, the warped
          // half starts at channel 0 (s.tch)
          if (!s.omit_ref) unpack(ld4<VEC>(dplane0 + chan * cstride + rowoff, x4, W), dmain);
          else { dmain[0] = dmain[1] = dmain[2] = dmain[3] = 0.f; }
          if constexpr (SAMPLED) unpack(ld4<VEC>(dplane0 + (chan + s.tch) * cstride + rowoff, x4, W), dwarp);
          int pend_a = -1;
          float pend_v = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const int x = x4 + kk;
            const bool inb = live && (VEC || x < W);
            float slope, f;
            int a;
            const float ev = diff(r, kk, lv[kk], inb, slope, a, f);
            float ge = -2.f * ev * dg0r[kk];
            ge -= 0.125f * m1[r >> 1][kk >> 1] * dp1[r >> 1][kk >> 1];   // 2 * (sum/4) / 4
            ge -= 0.125f * m2 * dp2;                                      // 2 * mean / 16
            float gl, gt;
            if constexpr (SAMPLED) {
              gl = ge + dmain[kk];
              gt = dwarp[kk] - ge;
            } else {
              ge -= 2.f * ev * dmain[kk];
              gl = ge;
              gt = -ge;
            }
            // gL: sum over the block's candidates in the neighbouring lanes, one plain store by the first of them
            float gsum = inb ? gl : 0.f;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
              const float up = __shfl_down(gsum, off, 64);
              gsum += (dl + off < D) ? up : 0.f;
            }
            if (inb && dl == 0) accL[r * Wl + x] = gsum;
            // gR: two taps per pixel; the +1 tap is carried to the next pixel and merged into its tap when they share
            // a column (neighbours of a smooth candidate map do).  Predicated atomics, not a dump slot: the cost of an
            // LDS atomic is per ACTIVE lane, and lanes sent to one dump address serialise (0.44 ms vs 0.40 ms).
            const bool t0 = inb && a >= 0 && a < W, t1 = SAMPLED && inb && a + 1 >= 0 && a + 1 < W;
            const bool merge = t0 && pend_a == a;
            if (pend_a >= 0 && !merge) lds_add(accR + r * Wl + pend_a, pend_v);
            if (t0) lds_add(accR + r * Wl + a, (1.f - f) * gt + (merge ? pend_v : 0.f));
            pend_a = t1 ? a + 1 : -1;
            pend_v = f * gt;
            if constexpr (SAMPLED) gdacc[r][kk] -= inb ? gt * slope : 0.f;
          }
          if (pend_a >= 0) lds_add(accR + r * Wl + pend_a, pend_v);
        }
      }
    }
    __syncthreads();
    // the channel's rows leave as plain stores; the tiles are cleared for the next channel
    for (int i = tid; i < 2 * TR * s.Wq; i += nthr) {
      const int which = i / (TR * s.Wq);                  // 0: gL, 1: gR
      const int rj = i - which * TR * s.Wq;
      const int r = rj / s.Wq, j = rj - r * s.Wq;
      float4 v;
      if (which) {
        double* acc = accR + r * Wl + 4 * j;
        v = make_float4(static_cast<float>(acc[0]), static_cast<float>(acc[1]), static_cast<float>(acc[2]), static_cast<float>(acc[3]));
        acc[0] = acc[1] = acc[2] = acc[3] = 0.0;
      } else {
        float* acc = accL + r * Wl + 4 * j;
        v = *reinterpret_cast<const float4*>(acc);
        *reinterpret_cast<float4*>(acc) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float* dst = which ? gR : gL;
      const int y = y0 + r;
      if (dst != nullptr && y < H) st4<VEC>(dst + goff + c * HW + static_cast<size_t>(y) * W, 4 * j, W, v);
    }
    __syncthreads();
  }
  if constexpr (SAMPLED) {
    if (gD && live) {
#pragma unroll
      for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (y0 + r < H && x4 + k < W)
            unsafeAtomicAdd(gD + (static_cast<size_t>(b) * D + d) * HW + static_cast<size_t>(y0 + r) * W + x4 + k,
                            gdacc[r][k]);
    }
  }
}

// Backward of the sampled path with one lane per ROW of a 4x4 block and the candidates as the OUTER loop (round 5; the forward
// counterpart is block_cost_corr_rows).  block_cost_bwd_tile gives a lane one (candidate, block) and all four rows: the block means
// need all rows before any gradient can be formed, so it evaluates every difference twice (two passes over the rows per channel),
// sums gL over the candidate lanes with four shuffles per pixel and stages it through LDS.  Here the four rows of a block are the four
// lanes of a quad: e is evaluated ONCE, the 2x2 / 4x4 sums are two quad permutes, gL accumulates over the candidates in registers
// and leaves as plain stores, the tap position is one division per (candidate, pixel) instead of one per channel.  What stays is the
// scatter of gR: LDS float atomics into one row tile per channel (all eight resident: the candidate loop is outside).
template <int NT = 256>               // NT: workgroup size bound (512: maps of 260-512 columns)
__global__ void __launch_bounds__(NT)
block_cost_bwd_rows(const float* __restrict__ L, const float* __restrict__ R, const float* __restrict__ disp,
                    const float* __restrict__ dout, const float* __restrict__ dP1, const float* __restrict__ dP2,
                    float* __restrict__ gL, float* __restrict__ gR, float* __restrict__ gD, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int by = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int y0 = by * TR;
  const int H = s.H, W = s.W, D = s.D, C = s.C;
  const size_t HW = static_cast<size_t>(H) * W;
  const size_t goff = (static_cast<size_t>(b) * C + g * GRP) * HW;
  const float* Lg = L + goff;
  const int Wl = 4 * s.Wq;
  // gR accumulates in DOUBLE: on gfx950 an LDS float atomic add costs 192 cycles per wave instruction (three per lane, serialised),
  // ds_add_f64 8.6 (tools/exp/lds_atomic_rate.hip: ds_add_u32 6.5, ds_add_u64 8.1, a plain write 7.2).  Four channels at a time so
  // that two workgroups still share a CU; the candidate loop runs once per half.
  constexpr int CH = GRP / 2;
  double* accR = reinterpret_cast<double*>(lds + static_cast<size_t>(GRP) * TR * 4 * s.Wqp);       // [CH][TR][Wl]
  const int tid = threadIdx.x, nthr = blockDim.x;
  stage_right_rows<true>(lds, R + goff, y0, H, W, HW, s.Wq, s.Wqp);   // ends with a barrier

  const int r = tid & 3, bx = tid >> 2;
  const int y = y0 + r, x4 = bx * 4;
  const bool live = bx < s.nbx && y < H;
  const float m = live ? 1.f : 0.f;
  const size_t rowoff = static_cast<size_t>(min(y, H - 1)) * W + (bx < s.nbx ? x4 : 0);
  const float Wm1 = static_cast<float>(W - 1);
  const size_t cstride = static_cast<size_t>(D) * HW;

  auto lds_add = [](double* p, float v) {
    __hip_atomic_fetch_add(p, static_cast<double>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  const int py = 2 * by + (r >> 1);

#pragma unroll 1
 for (int c0 = 0; c0 < GRP; c0 += CH) {
  for (int i = tid; i < CH * TR * Wl; i += nthr) accR[i] = 0.0;
  float lv[CH][4], gl[CH][4];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    unpack(*reinterpret_cast<const float4*>(Lg + (c0 + c) * HW + rowoff), lv[c]);
#pragma unroll
    for (int k = 0; k < 4; ++k) gl[c][k] = 0.f;
  }
  __syncthreads();
#pragma unroll 1
  for (int d = 0; d < D; ++d) {
    const float* dplane = dout + (static_cast<size_t>(b) * s.Ctot * D + d) * HW + rowoff;
    const size_t pbase = (static_cast<size_t>(b) * s.G + g) * D + d;
    float dv[4], dg0[4];
    unpack(*reinterpret_cast<const float4*>(disp + (static_cast<size_t>(b) * D + d) * HW + rowoff), dv);
    unpack(*reinterpret_cast<const float4*>(dplane + static_cast<size_t>(s.mainC + g) * cstride), dg0);
    float dp1a = 0.f, dp1b = 0.f, dp2 = 0.f;                 // this row pair's two 2x2 cells, the block's 4x4 cell
    if (live && s.scales > 1 && py < s.H1) {
      if (2 * bx < s.W1) dp1a = dP1[(pbase * s.H1 + py) * s.W1 + 2 * bx];
      if (2 * bx + 1 < s.W1) dp1b = dP1[(pbase * s.H1 + py) * s.W1 + 2 * bx + 1];
    }
    if (live && s.scales > 2 && by < s.H2 && bx < s.W2) dp2 = dP2[(pbase * s.H2 + by) * s.W2 + bx];
    int a[4];
    float f[4], w0[4], w1[4], gd[4] = {0.f, 0.f, 0.f, 0.f};
    int o0[4], o1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      column_of<true>(x4 + k, d, source_position(x4 + k, dv[k], W, Wm1), a[k], f[k]);
      const int i0 = min(max(a[k], 0), W - 1), i1 = min(max(a[k] + 1, 0), W - 1);
      o0[k] = (i0 & 3) * s.Wqp + (i0 >> 2);
      o1[k] = (i1 & 3) * s.Wqp + (i1 >> 2);
      w0[k] = (a[k] >= 0 && a[k] < W) ? 1.f : 0.f;            // masks by multiplication (a select becomes a load behind a branch)
      w1[k] = (a[k] + 1 >= 0 && a[k] + 1 < W) ? 1.f : 0.f;
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const float* src = lds + static_cast<size_t>((c0 + c) * TR + r) * 4 * s.Wqp;
      float e[4], slope[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float r0 = src[o0[k]] * w0[k], r1 = src[o1[k]] * w1[k];
        slope[k] = r1 - r0;
        e[k] = (lv[c][k] - ((1.f - f[k]) * r0 + f[k] * r1)) * m;
      }
      const float sa = e[0] + e[1], sb = e[2] + e[3];
      const float pa = sa + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(sa), 0xB1, 0xF, 0xF, true));
      const float pb = sb + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(sb), 0xB1, 0xF, 0xF, true));
      const float q = pa + pb;
      const float m2 = (q + __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(q), 0x4E, 0xF, 0xF, true))) * 0.0625f;
      const size_t chan = static_cast<size_t>(g * GRP + c0 + c);
      float dmain[4] = {0.f, 0.f, 0.f, 0.f}, dwarp[4];
      if (!s.omit_ref) unpack(*reinterpret_cast<const float4*>(dplane + chan * cstride), dmain);
      unpack(*reinterpret_cast<const float4*>(dplane + (chan + s.tch) * cstride), dwarp);
      double* acc = accR + static_cast<size_t>(c * TR + r) * Wl;
      int pend_a = -1;
      float pend_v = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float ge = -2.f * e[k] * dg0[k];
        ge -= 0.125f * (k < 2 ? pa * dp1a : pb * dp1b);         // 2 * (sum/4) / 4
        ge -= 0.125f * m2 * dp2;                                 // 2 * mean / 16
        gl[c][k] += live ? ge + dmain[k] : 0.f;
        const float gt = dwarp[k] - ge;
        // gR: two taps per pixel; the +1 tap is carried to the next pixel and merged into its tap when they share a column
        const bool t0 = live && a[k] >= 0 && a[k] < W, t1 = live && a[k] + 1 >= 0 && a[k] + 1 < W;
        const bool merge = t0 && pend_a == a[k];
        if (pend_a >= 0 && !merge) lds_add(acc + pend_a, pend_v);
        if (t0) lds_add(acc + a[k], (1.f - f[k]) * gt + (merge ? pend_v : 0.f));
        pend_a = t1 ? a[k] + 1 : -1;
        pend_v = f[k] * gt;
        gd[k] -= live ? gt * slope[k] : 0.f;
      }
      if (pend_a >= 0) lds_add(acc + pend_a, pend_v);
    }
    if (gD && live) {
#pragma unroll
      for (int k = 0; k < 4; ++k) unsafeAtomicAdd(gD + (static_cast<size_t>(b) * D + d) * HW + rowoff + k, gd[k]);
    }
  }
  if (gL != nullptr && live) {
#pragma unroll
    for (int c = 0; c < CH; ++c) st4<true>(gL + goff + (c0 + c) * HW + static_cast<size_t>(y) * W, x4, W, pack(gl[c]));
  }
  __syncthreads();
  if (gR != nullptr) {
    for (int i = tid; i < CH * TR * s.Wq; i += nthr) {
      const int j = i % s.Wq, cr = i / s.Wq;
      const int rr = cr & (TR - 1), c = cr >> 2;
      const double* a4 = accR + static_cast<size_t>(cr) * Wl + 4 * j;
      if (y0 + rr < H)
        st4<true>(gR + goff + (c0 + c) * HW + static_cast<size_t>(y0 + rr) * W, 4 * j, W,
                  make_float4(static_cast<float>(a4[0]), static_cast<float>(a4[1]), static_cast<float>(a4[2]), static_cast<float>(a4[3])));
    }
  }
  __syncthreads();
 }
}

// The same adjoint as a SCATTER over coalesced fine rows (round 5): a workgroup owns R2 rows of the 1/4-pooled map and the 2 R2 rows of
// the 1/2-pooled map over them, of one (b, g, d) plane; it reads the fine rows that can reach them (lanes along x), and every fine
// value goes to its (up to) four cells per level by ds_add_f64 (8.6 cycles per wave instruction: tools/exp/lds_atomic_rate.hip) --
// cells of other owners are dropped, so nothing is shared between workgroups.  block_cost_upsample_adjoint gathers instead: one lane
// per pooled cell walking a 6 x 6 ... 10 x 10 window of strided loads (48 us for 21 MB at the 1/4 level).
constexpr int ADJ_R2 = 2;
__global__ void __launch_bounds__(256)
block_cost_upsample_adjoint_rows(const float* __restrict__ dout, float* __restrict__ dP1, float* __restrict__ dP2, const Shape s) {
  extern __shared__ __attribute__((aligned(16))) double adj_lds[];
  const int band = blockIdx.x, plane = blockIdx.y;                 // plane = (b * G + g) * D + d
  const int d = plane % s.D, bg = plane / s.D;
  const int g = bg % s.G, b = bg / s.G;
  const size_t HW = static_cast<size_t>(s.H) * s.W;
  double* acc1 = adj_lds;                                           // [2 ADJ_R2][W1]
  double* acc2 = adj_lds + 2 * ADJ_R2 * s.W1;                       // [ADJ_R2][W2]
  const int n1 = 2 * ADJ_R2 * s.W1, n2 = (s.scales > 2) ? ADJ_R2 * s.W2 : 0;
  for (int i = threadIdx.x; i < n1 + n2; i += blockDim.x) adj_lds[i] = 0.0;
  __syncthreads();
#pragma unroll
  for (int lvl = 1; lvl <= 2; ++lvl) {
    if (lvl >= s.scales) break;
    const int Hs = (lvl == 1) ? s.H1 : s.H2, Ws = (lvl == 1) ? s.W1 : s.W2;
    const float rh = (lvl == 1) ? s.rh1 : s.rh2, rw = (lvl == 1) ? s.rw1 : s.rw2;
    const int r0 = (lvl == 1) ? 2 * ADJ_R2 * band : ADJ_R2 * band;                 // first owned pooled row
    const int nr = min((lvl == 1) ? 2 * ADJ_R2 : ADJ_R2, Hs - r0);
    if (nr <= 0) continue;
    double* acc = (lvl == 1) ? acc1 : acc2;
    // fine rows that can touch rows [r0, r0 + nr): the same bounds the gather form uses, per edge
    int ylo = 0, yhi = s.H - 1;
    if (rh > 0.f) {
      ylo = max(0, static_cast<int>(floorf((r0 - 1) / rh)) - 1);
      yhi = min(s.H - 1, static_cast<int>(ceilf((r0 + nr) / rh)) + 1);
    }
    const float* src = dout + ((static_cast<size_t>(b) * s.Ctot + s.mainC + lvl * s.G + g) * s.D + d) * HW;
    const int nrows = yhi - ylo + 1;
    for (int i = threadIdx.x; i < nrows * s.W; i += blockDim.x) {
      const int yy = i / s.W, x = i - yy * s.W;
      const int y = ylo + yy;
      const float hr = rh * static_cast<float>(y);
      const int h1 = static_cast<int>(hr);
      const int hp = (h1 < Hs - 1) ? 1 : 0;
      const float hl = hr - static_cast<float>(h1);
      const int ra = h1 - r0, rb = h1 + hp - r0;
      const bool oa = ra >= 0 && ra < nr, ob = rb >= 0 && rb < nr;
      if (!oa && !ob) continue;
      const float v = src[static_cast<size_t>(y) * s.W + x];
      const float wr = rw * static_cast<float>(x);
      const int w1 = static_cast<int>(wr);
      const int wp = (w1 < Ws - 1) ? 1 : 0;
      const float wl = wr - static_cast<float>(w1);
      auto add = [](double* p, float t) { __hip_atomic_fetch_add(p, static_cast<double>(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
      if (oa) {
        add(acc + ra * Ws + w1, (1.f - hl) * (1.f - wl) * v);
        add(acc + ra * Ws + w1 + wp, (1.f - hl) * wl * v);
      }
      if (ob) {
        add(acc + rb * Ws + w1, hl * (1.f - wl) * v);
        add(acc + rb * Ws + w1 + wp, hl * wl * v);
      }
    }
  }
  __syncthreads();
  const size_t pbase = static_cast<size_t>(plane);
  for (int i = threadIdx.x; i < n1; i += blockDim.x) {
    const int rr = i / s.W1, c = i - rr * s.W1;
    const int row = 2 * ADJ_R2 * band + rr;
    if (row < s.H1) dP1[(pbase * s.H1 + row) * s.W1 + c] = static_cast<float>(acc1[i]);
  }
  for (int i = threadIdx.x; i < n2; i += blockDim.x) {
    const int rr = i / s.W2, c = i - rr * s.W2;
    const int row = ADJ_R2 * band + rr;
    if (row < s.H2) dP2[(pbase * s.H2 + row) * s.W2 + c] = static_cast<float>(acc2[i]);
  }
}

template <bool SAMPLED>
int launch_bwd(const float* left, const float* right, const float* disp, const float* grad_out,
               float* grad_left, float* grad_right, float* grad_disp, void* workspace,
               int B, int C, int H, int W, int D, int scales, void* stream, int omit_ref = 0) {
  Shape s;
  if (int rc = make_shape(s, SAMPLED, B, C, H, W, D, scales, omit_ref)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(grad_out);
  if (SAMPLED) TS_REQUIRE_PTR(disp);
  if (scales > 1) TS_REQUIRE_PTR(workspace);
  hipStream_t st = ts::as_stream(stream);
  const size_t nfeat = static_cast<size_t>(B) * C * H * W * sizeof(float);
  // tile path: the workgroup that owns a row tile writes every element of it (no zero fill, no global atomics)
  const size_t tile_lds = (static_cast<size_t>(GRP) * TR * 4 * s.Wqp + 3 * TR * 4 * s.Wq + 4) * sizeof(float);   // R rows, gL tile (float), gR tile (double)
  const int tile_threads = D <= 16 ? ((s.nbx + 64 / D - 1) / (64 / D)) * 64 : 1 << 30;   // 64 / D blocks per wave
  const bool tile = tile_lds <= 64 * 1024 && tile_threads <= 512;
  if (!tile) {
    if (grad_left) if (hipError_t e = hipMemsetAsync(grad_left, 0, nfeat, st)) return ts::fail(e, "memset grad_left");
    if (grad_right) if (hipError_t e = hipMemsetAsync(grad_right, 0, nfeat, st)) return ts::fail(e, "memset grad_right");
  }
  if (SAMPLED && grad_disp)
    if (hipError_t e = hipMemsetAsync(grad_disp, 0, static_cast<size_t>(B) * D * H * W * sizeof(float), st))
      return ts::fail(e, "memset grad_disp");

  float* dP1 = reinterpret_cast<float*>(workspace);
  float* dP2 = scales > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + pooled_bytes(s, 1)) : nullptr;
  if (scales > 1) {
    const long long total = static_cast<long long>(B) * s.G * D * (static_cast<long long>(s.H1) * s.W1 + (scales > 2 ? s.H2 * s.W2 : 0));
    long long blocks = (total + 255) / 256;
    if (blocks > ts::kNumCU * 16) blocks = ts::kNumCU * 16;
    static const bool adj_rows = [] { const char* e = getenv("TS_K1_ADJOINT_ROWS"); return !e || atoi(e) != 0; }();
    const long long planes = static_cast<long long>(B) * s.G * D;
    const int bands = scales > 2 ? (s.H2 + ADJ_R2 - 1) / ADJ_R2 : (s.H1 + 2 * ADJ_R2 - 1) / (2 * ADJ_R2);
    const size_t alds = (static_cast<size_t>(2) * ADJ_R2 * s.W1 + ADJ_R2 * s.W2) * sizeof(double);
    // (with three scales H1 >= 2 H2: the 1/2-pooled rows beyond 2 R2 bands of the 1/4 map, if any, need one more band)
    const int bands1 = (s.H1 + 2 * ADJ_R2 - 1) / (2 * ADJ_R2);
    if (adj_rows && planes <= 65535 && alds <= 64 * 1024) {
      hipLaunchKernelGGL(block_cost_upsample_adjoint_rows, dim3(static_cast<unsigned>(bands > bands1 ? bands : bands1), static_cast<unsigned>(planes)),
                         dim3(256), alds, st, grad_out, dP1, dP2, s);
      if (int rc = ts::launched("block_cost_upsample_adjoint_rows")) return rc;
    } else {
      hipLaunchKernelGGL(block_cost_upsample_adjoint, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                         grad_out, dP1, dP2, s);
      if (int rc = ts::launched("block_cost_upsample_adjoint")) return rc;
    }
  }
  const bool vec = (W % 4 == 0) && ts::aligned16(left) && ts::aligned16(right) && ts::aligned16(grad_out) &&
                   (!SAMPLED || ts::aligned16(disp));
  const size_t lds_bytes = static_cast<size_t>(GRP) * TR * 4 * s.Wqp * sizeof(float);
  const bool stage = lds_bytes <= 64 * 1024;
  const int nitems = s.nbx * D;
  const int passes = (nitems + 255) / 256;
  int threads = static_cast<int>(ts::round_up(static_cast<size_t>((nitems + passes - 1) / passes), ts::kWave));
  if (threads > 256) threads = 256;
  const dim3 grid(s.nby, s.G, B);
  {   // sampled path on aligned maps of up to 256 columns: one lane per block row, candidates outside (block_cost_bwd_rows)
    static const bool rows = [] { const char* e = getenv("TS_K1_BWD_ROWS"); return !e || atoi(e) != 0; }();
    const int rthreads = static_cast<int>(ts::round_up(static_cast<size_t>(s.nbx) * 4, ts::kWave));
    const size_t rlds = (static_cast<size_t>(GRP) * TR * 4 * s.Wqp + static_cast<size_t>(GRP) * TR * 4 * s.Wq) * sizeof(float);   // R rows + [GRP/2][TR][Wl] doubles
    if (SAMPLED && rows && vec && rthreads <= 512 && rlds <= 80 * 1024 && (!grad_left || ts::aligned16(grad_left)) &&
        (!grad_right || ts::aligned16(grad_right))) {
      if (rthreads <= 256) {
        auto kern = &block_cost_bwd_rows<256>;
        if (rlds > 64 * 1024)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(rlds));
        hipLaunchKernelGGL(kern, grid, dim3(rthreads), rlds, st, left, right, disp, grad_out, dP1, dP2, grad_left, grad_right, grad_disp, s);
      } else {
        auto kern = &block_cost_bwd_rows<512>;
        if (rlds > 64 * 1024)
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(rlds));
        hipLaunchKernelGGL(kern, grid, dim3(rthreads), rlds, st, left, right, disp, grad_out, dP1, dP2, grad_left, grad_right, grad_disp, s);
      }
      return ts::launched("block_cost_bwd_rows");
    }
  }
  if (tile) {
    const int tthreads = tile_threads;
    if (vec) hipLaunchKernelGGL((block_cost_bwd_tile<SAMPLED, true>), grid, dim3(tthreads), tile_lds, st, left, right, disp,
                                grad_out, dP1, dP2, grad_left, grad_right, grad_disp, s);
    else hipLaunchKernelGGL((block_cost_bwd_tile<SAMPLED, false>), grid, dim3(tthreads), tile_lds, st, left, right, disp,
                            grad_out, dP1, dP2, grad_left, grad_right, grad_disp, s);
    return ts::launched("block_cost_bwd_tile");
  }
#define TS_LAUNCH_BWD(V, S)                                                                          \
  hipLaunchKernelGGL((block_cost_bwd_main<SAMPLED, V, S>), grid, dim3(threads), (S) ? lds_bytes : 0, st, \
                     left, right, disp, grad_out, dP1, dP2, grad_left, grad_right, grad_disp, s)
  if (vec && stage) TS_LAUNCH_BWD(true, true);
  else if (vec) TS_LAUNCH_BWD(true, false);
  else if (stage) TS_LAUNCH_BWD(false, true);
  else TS_LAUNCH_BWD(false, false);
#undef TS_LAUNCH_BWD
  return ts::launched("block_cost_bwd_main");
}

}  // namespace

extern "C" size_t ts_block_cost_workspace_bytes(int B, int C, int H, int W, int D, int scales) {
  Shape s;
  if (make_shape(s, false, B, C, H, W, D, scales) != TS_OK) return 0;
  if (scales < 2) return 256;
  return pooled_bytes(s, 1) + pooled_bytes(s, 2);
}

extern "C" int ts_block_cost_int_fwd(const float* left, const float* right, float* out, void* workspace,
                                     int B, int C, int H, int W, int D, int scales, void* stream) {
  return launch_fwd<false>(left, right, nullptr, out, workspace, B, C, H, W, D, scales, stream);
}

extern "C" int ts_block_cost_sampled_fwd(const float* left, const float* right, const float* disp, float* out,
                                         void* workspace, int B, int C, int H, int W, int D, int scales,
                                         void* stream) {
  return launch_fwd<true>(left, right, disp, out, workspace, B, C, H, W, D, scales, stream);
}

extern "C" int ts_block_cost_sampled_warped_fwd(const float* left, const float* right, const float* disp, float* out,
                                                void* workspace, int B, int C, int H, int W, int D, int scales,
                                                void* stream) {
  return launch_fwd<true>(left, right, disp, out, workspace, B, C, H, W, D, scales, stream, 1);
}

extern "C" int ts_block_cost_sampled_corr_fwd(const float* left, const float* right, const float* disp, float* out,
                                              void* workspace, int B, int C, int H, int W, int D, int scales,
                                              void* stream) {
  return launch_fwd<true>(left, right, disp, out, workspace, B, C, H, W, D, scales, stream, 2);
}

namespace {
template <int MODE>
int launch_dense(const float* left, const float* right, const float* disp, float* out, unsigned* maxbits,
                 int B, int C, int H, int W, int D, void* stream) {
  Shape s;
  if (int rc = make_shape(s, true, B, C, H, W, D, 1, 0, 1)) return rc;       // no pooling here: any H, W
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(disp);
  if (MODE != 1) TS_REQUIRE_PTR(out);
  if (MODE == 1 || MODE == 2) TS_REQUIRE_PTR(maxbits);
  bool vec = (W % 4 == 0) && ts::aligned16(left) && ts::aligned16(right) && ts::aligned16(disp) && (MODE == 1 || ts::aligned16(out));
  const size_t base_bytes = (static_cast<size_t>(2) * TRD * 4 * s.Wqp * 4 + static_cast<size_t>(GRP) * TRD * 4 * s.Wq) * sizeof(float);
  // the run-order (16-byte) form keeps 16 KB of candidates in LDS on top of the row staging: where that no longer fits (aligned maps
  // of roughly 384 <= W <= 508) the row-order form serves, as it did before the run-order form existed
  if (vec && base_bytes + 1024 * 16 > 64 * 1024) vec = false;
  const size_t lds_bytes = base_bytes + (vec ? 1024 * 16 : 0);
  TS_REQUIRE(lds_bytes <= 64 * 1024, TS_ERR_UNSUPPORTED, "cat/dif_fms: W=%d too wide for the row staging", W);
  const int CO = (MODE == 0) ? 2 * C : C;
  TS_REQUIRE(static_cast<unsigned long long>(CO) * D * H * W * 4ull < (1ull << 32), TS_ERR_UNSUPPORTED,
             "cat/dif_fms: one batch element of the volume spans 4 GiB or more");
  const dim3 grid((H + TRD - 1) / TRD, s.G, B);
  hipStream_t st = ts::as_stream(stream);
  if (vec) hipLaunchKernelGGL((dense_warp_kernel<MODE, true>), grid, dim3(256), lds_bytes, st, left, right, disp, out, maxbits, s);
  else hipLaunchKernelGGL((dense_warp_kernel<MODE, false>), grid, dim3(256), lds_bytes, st, left, right, disp, out, maxbits, s);
  return ts::launched("dense_warp_kernel");
}
}  // namespace

// cat_fms (aggregation/utils/cat_fms.py:5-36): out [B,2C,D,H,W] = cat[left repeated over D, right warped by disp[:, d]]
extern "C" int ts_cat_fms_fwd(const float* left, const float* right, const float* disp, float* out, int B, int C, int H,
                              int W, int D, void* stream) {
  return launch_dense<0>(left, right, disp, out, nullptr, B, C, H, W, D, stream);
}

// inverse_warp_3d (layers/inverse_warp_3d.py:4-58) for a 4-D image, padding_mode 'zeros', no disp_Y:
// out [B,C,D,H,W][b,c,d,y,x] = lerp(img[b,c,y,:], x + disp[b,d,y,x]) with zeros outside [0, W-1] (grid_sample, align_corners=True)
extern "C" int ts_inverse_warp_3d_fwd(const float* img, const float* disp, float* out, int B, int C, int H, int W, int D,
                                      void* stream) {
  return launch_dense<3>(img, img, disp, out, nullptr, B, C, H, W, D, stream);
}

extern "C" size_t ts_dif_fms_workspace_bytes(void) { return 256; }

// dif_fms (aggregation/utils/dif_fms.py:5-44): out [B,C,D,H,W] = |left - warped|, elements whose warped value is not > 0
// replaced by the maximum difference of the whole tensor (two passes: the maximum first, into the workspace).
extern "C" int ts_dif_fms_fwd(const float* left, const float* right, const float* disp, float* out, void* workspace,
                              int B, int C, int H, int W, int D, void* stream) {
  TS_REQUIRE_PTR(workspace);
  hipError_t e = hipMemsetAsync(workspace, 0, sizeof(unsigned), ts::as_stream(stream));
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "dif_fms: %s", hipGetErrorString(e));
  if (int rc = launch_dense<1>(left, right, disp, nullptr, reinterpret_cast<unsigned*>(workspace), B, C, H, W, D, stream)) return rc;
  return launch_dense<2>(left, right, disp, out, reinterpret_cast<unsigned*>(workspace), B, C, H, W, D, stream);
}

extern "C" size_t ts_block_cost_bwd_workspace_bytes(int B, int C, int H, int W, int D, int scales) {
  return ts_block_cost_workspace_bytes(B, C, H, W, D, scales);
}

extern "C" int ts_block_cost_int_bwd(const float* left, const float* right, const float* grad_out,
                                     float* grad_left, float* grad_right, void* workspace,
                                     int B, int C, int H, int W, int D, int scales, void* stream) {
  return launch_bwd<false>(left, right, nullptr, grad_out, grad_left, grad_right, nullptr, workspace,
                           B, C, H, W, D, scales, stream);
}

// Backward of ts_block_cost_sampled_warped_fwd: grad_out [B, C + scales*C/8, D, H, W] (the volume without its reference half; round 5:
// the first layer of a sampled level takes the D-invariant left half as a per-pixel term in training too, functional.first_layer_split)
extern "C" int ts_block_cost_sampled_warped_bwd(const float* left, const float* right, const float* disp,
                                                const float* grad_out, float* grad_left, float* grad_right,
                                                float* grad_disp, void* workspace,
                                                int B, int C, int H, int W, int D, int scales, void* stream) {
  return launch_bwd<true>(left, right, disp, grad_out, grad_left, grad_right, grad_disp, workspace,
                          B, C, H, W, D, scales, stream, 1);
}

extern "C" int ts_block_cost_sampled_bwd(const float* left, const float* right, const float* disp,
                                         const float* grad_out, float* grad_left, float* grad_right,
                                         float* grad_disp, void* workspace,
                                         int B, int C, int H, int W, int D, int scales, void* stream) {
  return launch_bwd<true>(left, right, disp, grad_out, grad_left, grad_right, grad_disp, workspace,
                          B, C, H, W, D, scales, stream);
}
