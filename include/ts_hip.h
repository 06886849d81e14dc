/*
 * ts_hip.h -- C ABI of libts_hip.so: the MI355X (gfx950) cost-volume stereo hot path.
 *
 * Drop-in boundary for the hot path of youmi-zym/TemporalStereo.
 * Every entry point replaces one python-level function of the reference; the reference
 * file:line is cited on each declaration.  The reference-side binding a maintainer would add
 * (a ctypes stub inside architecture/modeling/...) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all tensors are dense fp32, NCHW / NCDHW contiguous, device pointers owned by the caller;
 *     the library never allocates, frees or retains a pointer and never synchronises.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - return value: 0 = ok; negative = argument error (TS_ERR_*); positive = hipError_t of the
 *     failed launch.  ts_last_error_string() describes the last failure on the calling thread.
 *   - kernels needing scratch take `workspace` + a ts_*_workspace_bytes() query; 256-byte aligned.
 *   - re-entrant: no global mutable state besides the thread-local error string.
 */
#ifndef TS_HIP_H
#define TS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS_OK 0
#define TS_ERR_NULL (-1)         /* a required pointer is NULL                         */
#define TS_ERR_SHAPE (-2)        /* sizes inconsistent / non-positive                  */
#define TS_ERR_UNSUPPORTED (-3)  /* valid for the reference but outside this build     */
#define TS_ERR_ALIGN (-4)        /* pointer not 16-byte aligned                        */

int ts_version(void);                     /* ABI version, bumps on any signature change */
const char* ts_last_error_string(void);   /* thread-local, never NULL                   */

/* ------------------------------------------------------------------------------------------
 * K1  cost-volume construction.
 * Replaces block_cost()  architecture/modeling/aggregation/utils/block_cost.py:16-83
 *   (+ groupwise_correlation :6-13, inverse_warp_3d  layers/inverse_warp_3d.py:4-58).
 * left,right [B,C,H,W]; C % 8 == 0; H,W >= 4; 1 <= scales <= 3.
 *   int path      (:34-45): out [B, C  + scales*C/8, D, H, W]
 *   sampled path  (:47-58): out [B, 2C + scales*C/8, D, H, W], disp [B,D,H,W], D >= 2
 * ---------------------------------------------------------------------------------------- */
size_t ts_block_cost_workspace_bytes(int B, int C, int H, int W, int D, int scales);

int ts_block_cost_int_fwd(const float* left, const float* right, float* out, void* workspace,
                          int B, int C, int H, int W, int D, int scales, void* stream);

int ts_block_cost_sampled_fwd(const float* left, const float* right, const float* disp, float* out,
                              void* workspace, int B, int C, int H, int W, int D, int scales,
                              void* stream);

/* Backward of the two paths (autograd of the reference's torch ops).  grad_out has the layout
 * of `out`.  grad_left/grad_right [B,C,H,W] and grad_disp [B,D,H,W] are OVERWRITTEN (any may be
 * NULL to skip).  grad_right / grad_disp accumulate with fp32 atomics (order not deterministic,
 * as in the reference's grid_sampler backward). */
size_t ts_block_cost_bwd_workspace_bytes(int B, int C, int H, int W, int D, int scales);

int ts_block_cost_int_bwd(const float* left, const float* right, const float* grad_out,
                          float* grad_left, float* grad_right, void* workspace,
                          int B, int C, int H, int W, int D, int scales, void* stream);

int ts_block_cost_sampled_bwd(const float* left, const float* right, const float* disp,
                              const float* grad_out, float* grad_left, float* grad_right,
                              float* grad_disp, void* workspace,
                              int B, int C, int H, int W, int D, int scales, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement support (no reference counterpart): float4 streams used by bench.py to calibrate
 * what this box sustains.  kind 0 = fill dst (write-only), 1 = copy src->dst, 2 = read src
 * (dst = 4-byte sink).  nbytes % 16 == 0.
 * ---------------------------------------------------------------------------------------- */
int ts_calib_stream(int kind, void* dst, const void* src, size_t nbytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TS_HIP_H */
