// Device-side pieces shared by the convolution kernels (conv3d.hip, conv_x6s.hip): activation codes, the launch
// descriptor, buffer-resource helper and the fp32 -> three-bf16 split of the "x6" kernels.
#pragma once
#include <cstdlib>

#include "ts_common.hpp"

namespace {

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_TANH_OFFSET = 3, ACT_HEAD_PAIR = 4 };

// co: output channel (only ACT_HEAD_PAIR looks at it: channel 0 = the cost head, no activation; channel 1 =
// the offset head -- both prediction heads as one block-diagonal convolution)
// v / (1 + e^-v) with v_exp_f32 and v_rcp_f32 (1 ulp each) instead of the library's expf and an IEEE division: ~6 instructions per value
// (+ one Newton step on the reciprocal: 8) instead of ~30.  A convolution lane finishes 16-32 values with nothing left to hide them -- 4.5 k of a 7 k-cycle epilogue in the x6
// kernels (tools/exp/x6p_trace.py), 1-2 us of every launch-bound layer of the pyramid.  |error| ~1.5 ulp of the result, the class of the exact form (the unrefined reciprocal's extra ulp was enough to move one more near-tie of
// a temporal sequence in tests/test_fullsize_gpu.py's audit); the e^-v
// argument's rounding (|v| 2^-24 relative: |v| 6e-8 of e^-v) only meets results that are themselves ~e^-|v|.  Below v = -88.7 e^-v is inf, the
// reciprocal 0 and the Newton step would be inf * 0: the unrefined 0 is kept there (v * 0 = -0, as the exact form's v / inf; NaN stays NaN)
// (tests/test_conv_x6_gpu.py::test_silu_epilogue_at_the_ends_of_the_range).
// TS_EXACT_SILU (compile-time) restores the library form.
__device__ __forceinline__ float silu_fast(float v) {
#ifdef TS_EXACT_SILU
  return v / (1.f + expf(-v));
#else
  const float d = 1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
  const float r = __builtin_amdgcn_rcpf(d);
  const float rn = fmaf(fmaf(-d, r, 1.f), r, r);          // one Newton step: the quotient to ~0.5 ulp, as the IEEE division's
  return v * (d < 3.0e38f ? rn : r);
#endif
}

__device__ __forceinline__ float apply_act(float v, int act, float p, int co = 0) {
  if (act == ACT_HEAD_PAIR) act = co == 0 ? ACT_NONE : ACT_TANH_OFFSET;
  switch (act) {
    case ACT_SILU: return silu_fast(v);
    case ACT_RELU: return fmaxf(v, 0.f);
    // PredictionHeads.regress_offset (module.py:384-390): tanh(x/100).clamp(-1,1) * delta
    case ACT_TANH_OFFSET: return fminf(fmaxf(tanhf(v / 100.f), -1.f), 1.f) * p;
    default: return v;
  }
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
enum { MODE_HW = 0, MODE_HWT = 1, MODE_D = 2 };
constexpr unsigned kOOB = 0x80000000u;   // buffer offset past every num_records we allow: the load returns 0

struct IG {
  int Cin, Cout, coutp;         // coutp: padded channel count of the weight / scale / shift arrays
  int D, H, W;                  // input: planes per channel, plane geometry
  int Do, Ho, Wo;               // output
  int stride, dil, pad, k, transposed;
  int act;
  float act_param;
  long long in_bstride, in_cstride, out_bstride, out_cstride;
  unsigned in_bytes, w_bytes;   // extent of one batch element of x / of the weight array (buffer range checks)
  unsigned out_bytes, part_bytes;   // ... of one batch element of y / of one (slice, batch) block of the partials
  int tiles_x, co_groups;
  int ksplit, kspan;            // split-K: this many slices of `kspan` input channels each (partials -> workspace)
  float* partial;               // [ksplit][B][Cout][Do*Ho*Wo] raw sums when ksplit > 1
  int B;
  const float* addend;          // [B][Cout][Ho*Wo] added to every depth plane's sum before scale/shift (or null);
  long long add_bstride;        // add_dstride != 0: one term per depth plane, [B][Cout][D][Ho*Wo] (ts_conv3d_hw_warp_fwd)
  long long add_cstride, add_dstride;
  int xcd;                      // XCD-banded workgroup order (ig_conv_kernel)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ig_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
// (a, b) -> packed (hi, mid, lo) parts
__device__ __forceinline__ void split6(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = pack_bf16(a, b);
  a -= __uint_as_float(hi << 16); b -= __uint_as_float(hi & 0xffff0000u);
  mid = pack_bf16(a, b);
  a -= __uint_as_float(mid << 16); b -= __uint_as_float(mid & 0xffff0000u);
  lo = pack_bf16(a, b);
}

inline long long env_ll(const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; }
inline bool env_not_zero(const char* name) { const char* e = getenv(name); return !(e && e[0] == '0'); }

}  // namespace
