// Shared host-side helpers for libts_hip.so (gfx950 only; no CUDA compatibility layer).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/ts_hip.h"

#define TS_ABI_VERSION 10   // 10: ts_inverse_warp_3d_fwd (the inverse_warp_3d seam); 9: ts_peer_status_async / set_timeout_ms / reset, layout + bf16 split of a framework-layout weight in one launch (ts_conv3d_hw_x6_weight_split_from), two-table weight layouts (ts_conv_weight_layout_many2); 8: ts_peer_* (SyncBatchNorm exchanges as kernels over peer-mapped memory); 7: ts_conv3d_hw_x6s_* (bf16-split stride-2 / transposed convolutions); 6: ts_block_cost_sampled_corr_fwd, ts_conv3d_hw_warp_fwd (first layer of a sampled level without the warped volume); 5: split-K workspace of ts_conv3d_hw_x6_fwd; 4: train_ops (candidates, offset head, deconv2d training forms, clip+RMSprop); 3: bwd_weight workspace, bn_stats counter, x6 / weight-layout / bn_train

namespace ts {

// thread-local message for ts_last_error_string()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// check the launch that was just issued; returns 0 or the positive hipError_t
inline int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(static_cast<int>(e), "%s: %s", what, hipGetErrorString(e));
  return TS_OK;
}

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kNumCU = 256;     // MI355X
constexpr int kNumXCD = 8;

}  // namespace ts

#define TS_REQUIRE_PTR(p)                                                     \
  do {                                                                        \
    if ((p) == nullptr) return ts::fail(TS_ERR_NULL, "%s is NULL", #p);       \
  } while (0)
#define TS_REQUIRE_ALIGNED(p)                                                          \
  do {                                                                                 \
    if (!ts::aligned16(p)) return ts::fail(TS_ERR_ALIGN, "%s not 16-byte aligned", #p); \
  } while (0)
#define TS_REQUIRE(cond, code, ...)                       \
  do {                                                    \
    if (!(cond)) return ts::fail((code), __VA_ARGS__);    \
  } while (0)
