// Device-side pieces shared by the convolution kernels (conv3d.hip, conv_x6s.hip): activation codes, the launch
// descriptor, buffer-resource helper and the fp32 -> three-bf16 split of the "x6" kernels.
#pragma once
#include <cstdlib>

#include "ts_common.hpp"

namespace {

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_TANH_OFFSET = 3, ACT_HEAD_PAIR = 4 };

// co: output channel (only ACT_HEAD_PAIR looks at it: channel 0 = the cost head, no activation; channel 1 =
// the offset head -- both prediction heads as one block-diagonal convolution)
__device__ __forceinline__ float apply_act(float v, int act, float p, int co = 0) {
  if (act == ACT_HEAD_PAIR) act = co == 0 ? ACT_NONE : ACT_TANH_OFFSET;
  switch (act) {
    case ACT_SILU: return v / (1.f + expf(-v));
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
