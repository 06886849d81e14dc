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

/* Inference form of the sampled path: the first C channels of its output are `left` repeated D times
 * (block_cost.py:51), and the layer that consumes the volume is a (1,3,3) convolution, so
 *   conv(volume) = conv_{W[:, :C]}(left) broadcast over D  +  conv_{W[:, C:]}(volume[:, C:]).
 * This entry writes only volume[:, C:]  ->  out [B, C + scales*C/8, D, H, W]  (warped half, then the
 * correlation blocks); ts_conv3d_hw_fwd takes the left term as its `addend`. */
int ts_block_cost_sampled_warped_fwd(const float* left, const float* right, const float* disp, float* out,
                                     void* workspace, int B, int C, int H, int W, int D, int scales,
                                     void* stream);

/* Correlation blocks only (ABI 6): out [B, scales*C/8, D, H, W] = channels [2C:] of ts_block_cost_sampled_fwd.  The consumer is
 * ts_conv3d_hw_warp_fwd, which needs neither half of the 2C main channels as a volume. */
int ts_block_cost_sampled_corr_fwd(const float* left, const float* right, const float* disp, float* out,
                                   void* workspace, int B, int C, int H, int W, int D, int scales,
                                   void* stream);

/* Dense siblings of block_cost (forward only): any number of candidates D >= 2, C % 8 == 0.
 *   cat_fms  aggregation/utils/cat_fms.py:5-36   out [B,2C,D,H,W] = cat[left repeated over D, warped right]
 *   dif_fms  aggregation/utils/dif_fms.py:5-44   out [B, C,D,H,W] = |left - warped right|, elements whose warped
 *            value is not > 0 replaced by the maximum difference over the whole tensor (workspace: 256 bytes) */
int ts_cat_fms_fwd(const float* left, const float* right, const float* disp, float* out, int B, int C, int H, int W, int D,
                   void* stream);
/* inverse_warp_3d(img, disp, padding_mode='zeros', disp_Y=None)  layers/inverse_warp_3d.py:4-58, the function-level seam itself
 * (block_cost / cat_fms / dif_fms call it with -disp): img [B,C,H,W] (C % 8 == 0), disp [B,D,H,W], D >= 2, H, W >= 2 ->
 * out [B,C,D,H,W][b,c,d,y,x] = img[b,c,y, x + disp[b,d,y,x]] linearly interpolated, zeros outside [0, W-1].  The wrapper
 * (temporalstereo_amd.inverse_warp_3d) serves 5-D images, other channel counts and the gradients (ts_block_cost_sampled_bwd). */
int ts_inverse_warp_3d_fwd(const float* img, const float* disp, float* out, int B, int C, int H, int W, int D, void* stream);
size_t ts_dif_fms_workspace_bytes(void);
int ts_dif_fms_fwd(const float* left, const float* right, const float* disp, float* out, void* workspace, int B, int C,
                   int H, int W, int D, void* stream);

/* Correlation volumes, the native counterpart of the third-party SpatialCorrelationSampler the
 * reference imports optionally (aggregation/utils/correlation.py:4-7; unvendored and unpinned, so parity is pinned by the
 * sampler's published definition -- sum over channels, zeros outside the image -- through oracle/correlation.py):
 *   out[b, k, y, x] = leaky_relu_0.1( sum_c left[b,c,y,x] * right[b,c, y + k/pW - pH/2, x + k%pW - pW/2] ),  k < keep
 *   correlation   (correlation.py:10-29): patch (p, p), keep = p*p
 *   correlation1d (correlation.py:32-57): patch (1, 2*max_disp-1), keep = max_disp  (plane k = disparity max_disp-1-k)
 * Backward OVERWRITES grad_left / grad_right [B,C,H,W] (either may be NULL); deterministic. */
int ts_correlation_fwd(const float* left, const float* right, float* out, int B, int C, int H, int W, int patch_h, int patch_w,
                       int keep, void* stream);
int ts_correlation_bwd(const float* left, const float* right, const float* out, const float* grad_out, float* grad_left,
                       float* grad_right, int B, int C, int H, int W, int patch_h, int patch_w, int keep, void* stream);

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
/* ... of ts_block_cost_sampled_warped_fwd (ABI 9): grad_out [B, C + scales*C/8, D, H, W], the volume without its reference half */
int ts_block_cost_sampled_warped_bwd(const float* left, const float* right, const float* disp,
                                     const float* grad_out, float* grad_left, float* grad_right,
                                     float* grad_disp, void* workspace,
                                     int B, int C, int H, int W, int D, int scales, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (+ activation) of the convolution wrappers in TRAIN mode (layers/basic_layers.py:194-235: conv -> norm ->
 * activation).  x / out / dy / dx are [B,C,N] with N = D*H*W contiguous and explicit batch / channel strides (elements).
 *   ts_bn_stats_fwd       mean[C], var[C] (biased) of x over (B,N); deterministic.  running_mean / running_var (may be NULL) are
 *                         updated with `momentum` (running_var with the unbiased estimate, as nn.BatchNorm);
 *                         *num_batches_tracked (device int64, may be NULL) is incremented in the same launch
 *   ts_bn_apply_act_fwd   out = act((x - mean) * rsqrt(var + eps) * gamma + beta);  act 0 none | 1 SiLU | 2 ReLU
 *   ts_bn_act_bwd_reduce  sum_dz[C] = sum dz, sum_dz_xhat[C] = sum dz * xhat with dz = dy * act'(z)  (= grad beta, grad gamma)
 *   ts_bn_act_bwd_apply   train: dx = (dz - sum_dz/count - xhat * sum_dz_xhat/count) * invstd * gamma; eval: dx = dz * invstd * gamma
 * mean / var / the two sums are arguments so that cross-rank statistics (SyncBatchNorm) can be exchanged in between.
 * ---------------------------------------------------------------------------------------- */
size_t ts_bn_workspace_bytes(int B, int C, long long N);
/* SyncBatchNorm merge: gathered = the ranks' [mean(C) | biased var(C) | count] records of ts_bn_stats_fwd (world x (2C+1) floats, one
 * all_gather) -> statistics of the whole batch (parallel-variance formula, double), running statistics updated with momentum
 * (NULL: not tracked), *inv_count = 1 / total count on the device (the backward sums are scaled with it: no host read). */
int ts_bn_sync_merge(const float* gathered, int world, int C, float* mean, float* var, float* running_mean,
                     float* running_var, float momentum, float* inv_count, void* stream);
/* Single-rank training forms (nothing to exchange between the statistics and their use): statistics + normalise/activate in two
 * launches, backward sums + input gradient in two (each a partial-sums kernel, then one whose workgroups finish their channel's
 * sums themselves in the fixed order of the three-launch forms -- same values).  ts_bn_train_bwd: count = B*N elements.
 * ABI 9: channels of at most ts_bn_set_small_elems() elements (B*N; default below, TS_BN_SMALL_ELEMS) take ONE launch each way -- a
 * workgroup per channel does both passes (per-thread fp32 partial sums combined in double: the same statistics to rounding). */
long long ts_bn_set_small_elems(long long n);   /* returns the previous bound; n < 0 only queries */
/* out[c] = sum over batch and pixels of x [B,C,N] (a convolution's bias gradient), deterministic; workspace: ts_bn_workspace_bytes(B, C, N) */
int ts_channel_sum_fwd(const float* x, float* out, void* workspace, int B, int C, long long N, long long bstride, long long cstride,
                       void* stream);
int ts_bn_train_fwd(const float* x, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                    long long* num_batches_tracked, const float* gamma, const float* beta, float* out, void* workspace,
                    int B, int C, long long N, long long x_bstride, long long x_cstride, long long out_bstride,
                    long long out_cstride, float eps, int act, void* stream);
int ts_bn_train_bwd(const float* x, const float* dy, const float* mean, const float* var, const float* gamma, const float* beta,
                    float* sum_dz, float* sum_dz_xhat, float* dx, void* workspace, int B, int C, long long N,
                    long long x_bstride, long long x_cstride, long long dy_bstride, long long dy_cstride, float eps, int act,
                    float count, void* stream);
int ts_bn_stats_fwd(const float* x, float* mean, float* var, float* running_mean, float* running_var, float momentum,
                    long long* num_batches_tracked, void* workspace, int B, int C, long long N, long long bstride,
                    long long cstride, void* stream);
int ts_bn_apply_act_fwd(const float* x, const float* mean, const float* var, const float* gamma, const float* beta, float* out,
                        int B, int C, long long N, long long x_bstride, long long x_cstride, long long out_bstride,
                        long long out_cstride, float eps, int act, void* stream);
int ts_bn_act_bwd_reduce(const float* x, const float* dy, const float* mean, const float* var, const float* gamma,
                         const float* beta, float* sum_dz, float* sum_dz_xhat, void* workspace, int B, int C, long long N,
                         long long x_bstride, long long x_cstride, long long dy_bstride, long long dy_cstride, float eps, int act,
                         void* stream);
int ts_bn_act_bwd_apply(const float* x, const float* dy, const float* mean, const float* var, const float* gamma,
                        const float* beta, const float* sum_dz, const float* sum_dz_xhat, float* dx, int B, int C, long long N,
                        long long x_bstride, long long x_cstride, long long dy_bstride, long long dy_cstride, float eps, int act,
                        int train, float count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss side of the path's outputs.
 * ts_wasserstein_loss_*: WarssersteinDistanceLoss.loss_per_level  architecture/modeling/losses/warsserstein_distance_loss.py:52-78
 *   cost/offset/sample [B,D,H,W], gt [B,1,Hg,Wg] (full resolution; pooled to (H,W) after / (Wg/W): avg, or max when `sparse`).
 *   loss[0] = mean_{b,y,x} sum_d (softmax_d(cost)+0.25) |offset+sample-gt'| [start < gt' < max_disp/scale]; gt_scaled [B,H,W] is
 *   kept for the backward, which OVERWRITES grad_cost and grad_offset (== grad of sample); either may be NULL.
 * ts_disp_smooth_l1_*: DispSmoothL1Loss.loss_per_level (losses/smooth_l1_loss.py:49-76) of the disparity est [B,1,h,w] after the
 *   wrapper's rescale to the ground truth's size, F.interpolate(est * Wg / w, (Hg,Wg), bilinear, align_corners)
 *   (projects/TemporalStereo/TemporalStereo.py:305-309), evaluated in registers; (h,w) == (Hg,Wg) is the plain loss.
 *   loss_count[0] = mean smooth-L1 over the valid pixels (0 if none), loss_count[1] = their number (the backward reads it).
 * grad_loss: one float on the device (d objective / d loss).  Deterministic (fixed-order reductions, gather adjoint).
 * ---------------------------------------------------------------------------------------- */
size_t ts_wasserstein_loss_workspace_bytes(int B, int H, int W);
int ts_wasserstein_loss_fwd(const float* cost, const float* offset, const float* sample, const float* gt, float* loss,
                            float* gt_scaled, void* workspace, int B, int D, int H, int W, int Hg, int Wg, float max_disp,
                            float start_disp, int sparse, void* stream);
int ts_wasserstein_loss_bwd(const float* cost, const float* offset, const float* sample, const float* gt_scaled,
                            const float* grad_loss, float* grad_cost, float* grad_offset, int B, int D, int H, int W, int Wg,
                            float max_disp, float start_disp, void* stream);
size_t ts_disp_smooth_l1_workspace_bytes(int B, int Hg, int Wg);
int ts_disp_smooth_l1_fwd(const float* est, const float* gt, float* loss_count, void* workspace, int B, int h, int w, int Hg,
                          int Wg, float max_disp, float start_disp, void* stream);
int ts_disp_smooth_l1_bwd(const float* est, const float* gt, const float* grad_loss, const float* loss_count, float* grad_est,
                          int B, int h, int w, int Hg, int Wg, float max_disp, float start_disp, void* stream);

/* ------------------------------------------------------------------------------------------
 * K4  disparity regression.  cost / sample / offset are [B,D,H,W].
 * ts_topk_softargmax_*: predict_disp()  .../aggregation/TemporalStereo/coarse.py:69-75
 *   (== fine.py:70-76, precise.py:61-67): top-k (1 <= k <= 8, ties: lowest index first) ->
 *   softmax -> gather(sample+offset) -> weighted sum.  disp [B,1,H,W]; topk_* [B,k,H,W];
 *   topk_index (int32, may be NULL in fwd) is what the backward needs.  Backward overwrites
 *   grad_cost and grad_sample ([B,D,H,W]; grad of offset == grad of sample); any grad_* input may be NULL.
 * ts_softargmin_*: SOFTARGMIN.forward  architecture/modeling/prediction/soft_argmin.py:38-59
 * ts_argmax_select_fwd: ARGMIN.forward  architecture/modeling/prediction/argmin.py:35-46
 * ---------------------------------------------------------------------------------------- */
int ts_topk_softargmax_fwd(const float* cost, const float* sample, const float* offset, float* disp,
                           float* topk_disp, float* topk_cost, int* topk_index,
                           int B, int D, int H, int W, int k, void* stream);

int ts_topk_softargmax_bwd(const float* topk_disp, const float* topk_cost, const int* topk_index,
                           const float* disp, const float* grad_disp, const float* grad_topk_disp,
                           const float* grad_topk_cost, float* grad_cost, float* grad_sample,
                           int B, int D, int H, int W, int k, void* stream);

int ts_softargmin_fwd(const float* cost, const float* sample, float* disp, float temperature,
                      int normalize, int B, int D, int H, int W, void* stream);

int ts_softargmin_bwd(const float* cost, const float* sample, const float* disp, const float* grad_disp,
                      float* grad_cost, float* grad_sample, float temperature, int normalize,
                      int B, int D, int H, int W, void* stream);

int ts_argmax_select_fwd(const float* cost, const float* sample, float* disp, int* index,
                         int B, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  temporal cost warp.
 * ts_softsplat_sum_*: the three kernels of architecture/modeling/layers/softsplat.py:8-177
 *   (updateOutput :14-52, updateGradInput :63-105, updateGradFlow :116-176); input/output [B,C,H,W],
 *   flow [B,2,H,W] (x then y).  fwd zero-fills `output` itself.  fp32 atomics: order not deterministic.
 * ts_softsplat_softmax_fwd: FunctionSoftsplat(..., 'softmax')  softsplat.py:334-360 fused for the
 *   detached use in update_map (projects/TemporalStereo/TemporalStereo.py:415-419): metric [B,1,H,W].
 * ts_project_to_3d_fwd: project_to_3d()  architecture/modeling/layers/inverse_warp.py:92-178;
 *   depth [B,C,H,W], K [B,k,k] (k = 3|4), inv_K [B,ik,ik], T [B,4,4] -> triangular_depth [B,C,H,W],
 *   optical_flow [B,2C,H,W], flow_mask [B,C,H,W] (uint8); any output may be NULL.
 * ---------------------------------------------------------------------------------------- */
int ts_softsplat_sum_fwd(const float* input, const float* flow, float* output,
                         int B, int C, int H, int W, void* stream);
/* order-independent summation splat (64-bit fixed-point accumulation, integer atomics): bit-identical from
 * run to run; workspace B*C*H*W*8 bytes; |values| < 2^22 */
int ts_softsplat_sum_fwd_deterministic(const float* input, const float* flow, float* output, void* workspace,
                                       int B, int C, int H, int W, void* stream);
int ts_softsplat_sum_bwd_input(const float* flow, const float* grad_output, float* grad_input,
                               int B, int C, int H, int W, void* stream);
int ts_softsplat_sum_bwd_flow(const float* input, const float* flow, const float* grad_output,
                              float* grad_flow, int B, int C, int H, int W, void* stream);
size_t ts_softsplat_softmax_workspace_bytes(int B, int C, int H, int W);
int ts_softsplat_softmax_fwd(const float* input, const float* flow, const float* metric, float* output,
                             void* workspace, int B, int C, int H, int W, void* stream);
int ts_project_to_3d_fwd(const float* depth, const float* K, const float* inv_K, const float* T,
                         float* triangular_depth, float* optical_flow, unsigned char* flow_mask,
                         int B, int C, int H, int W, int k_dim, int inv_k_dim, float eps, void* stream);

/* K2c  the temporal state update of one frame, fused: update_map's closures update_local_map and
 * update_past_cost, projects/TemporalStereo/TemporalStereo.py:340-384 / :386-426, three launches.
 *   prev_disp  full-resolution disparity of the previous frame, [B,1,full_h,full_w] (disp_bstride elements
 *              between batch items); resized to (h,w) with value scale w/full_w  (:357-359)
 *   mem_disp / mem_cost [B,k,h,w]  top-k candidates and costs kept from the previous frame (k may be 0)
 *   local_map [B,n_local_in,h,w]   previous local maps (may be NULL when n_local_out <= 1);
 *              planes re-projected = [resized prev_disp | local_map][: n_local_out]      (:365-367)
 *   K [B,k_dim,k_dim] full-resolution intrinsics (rows 0,1 divided by `factor` = full_w / w inside; when 4x4 the
 *              last row must be 0 0 0 1); T = T_a * T_b (T_b NULL: T = T_a), [B,4,4]     (:333-338)
 *   baseline   per-item array [B] or NULL -> the scalar `baseline`
 *   out_disp / out_cost [B,k,h,w], out_local [B,n_local_out,h,w]: soft-max splat with metric
 *              clamp(d - mean(d), +-50) of the resized disparity (global mean, :374)     (:373-379, :415-419)
 * fp32 atomics in the splat: summation order not deterministic (as the reference's atomicAdd). */
size_t ts_reproject_memory_workspace_bytes(int B, int h, int w, int k, int n_local_out);
int ts_reproject_memory_fwd(const float* prev_disp, long long disp_bstride, int full_h, int full_w,
                            const float* mem_disp, const float* mem_cost, int k,
                            const float* local_map, int n_local_in, int n_local_out,
                            const float* K, int k_dim, const float* T_a, const float* T_b,
                            const float* baseline_ptr, float baseline, float factor,
                            float* out_disp, float* out_cost, float* out_local, void* workspace,
                            int B, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3  3-D aggregation pyramid, inference form (BatchNorm folded into per-channel scale/shift,
 * activation fused: act 0 none, 1 SiLU, 2 ReLU, 3 tanh(x/100).clamp(-1,1)*act_param, 4 = channel 0 none and
 * channel 1 as 3 (both PredictionHeads outputs from one block-diagonal convolution, hw family only)).
 * Tensors may be channel slices of larger buffers: *_bstride / *_cstride are element strides of
 * the batch and channel axes.  Weights are pre-laid out by the host with the output channel
 * contiguous and zero-padded to ts_conv_cout_pad(Cout): [Cin][taps][CoutPad]; scale/shift [CoutPad].
 * ts_conv3d_hw_fwd: Conv3d (1,3,3) / ConvTranspose3d (1,3,3) of DepthwiseConv3D /
 *   DepthwiseConvTranspose3D (.../TemporalStereo/module.py:111-184) through the Conv3d wrappers
 *   (layers/basic_layers.py:194-235,340-388); also every 3x3 Conv2d (D = 1).
 * ts_conv3d_d_fwd: the (k,1,1) halves, Conv3d(5,1,1) of PyramidFusion (:408), Conv3d(3,1,1) of
 *   PredictionHeads (:369-378), 1x1 convolutions (k = 1).
 * ---------------------------------------------------------------------------------------- */
int ts_conv_cout_pad(int cout);
/* out[a][t][b] = b < nb ? w[a*stride_a + b*stride_b + (flip ? T-1-t : t)*stride_t] : 0, out [A][T][bpad]: the [Cin][taps][CoutPad]
 * layouts of ts_conv3d_*_fwd / *_bwd_data from a framework weight ([Cout][Cin][taps] or [Cin][Cout][taps]) in one launch. */
int ts_conv_weight_layout(const float* w, float* out, int A, int T, int nb, int bpad, long long stride_a, long long stride_b,
                          long long stride_t, int flip, void* stream);
/* The same for n weights in one launch (training: every convolution weight is re-laid once per optimizer update).
 * table: n entries in DEVICE memory; blocks_x: 256-thread workgroups per entry (grid-stride over larger entries). */
typedef struct {
  const float* w; float* out;
  int A, T, nb, bpad;
  long long stride_a, stride_b, stride_t;
  int flip, reserved;
} ts_weight_layout_desc;               /* 64 bytes */
int ts_conv_weight_layout_many(const void* table, int n, int blocks_x, void* stream);
/* General form (ABI 9): out[a*out_stride_a + t*out_stride_t + col0 + b] = w[a*stride_a + b*stride_b + (flip ? T-1-t : t)*stride_t] for
 * b < nb -- only those elements are written (padding is the caller's, zeroed once).  One entry per (destination region, source):
 * concatenations along Cout, block-diagonal pairs and re-pitched arrays are entries over the framework's own parameters, so an
 * inference engine re-folds ALL its kernel-layout weights after an optimizer step in one launch (aggregation/native.py Tape). */
typedef struct {
  const float* w; float* out;
  int A, T, nb, col0;
  long long stride_a, stride_b, stride_t;
  long long out_stride_a, out_stride_t;
  int flip, reserved;
} ts_weight_layout_desc2;              /* 80 bytes */
int ts_conv_weight_layout_many2(const void* table, int n, int blocks_x, void* stream);
/* Deferred finishes of ts_conv3d_{hw,d}_bwd_weight (ABI 9): between ts_conv_wgrad_defer(1) and (0) the calls ON THIS HOST THREAD leave
 * their partial sums in the caller's workspaces -- which must stay alive -- and dw unwritten; ts_conv_wgrad_take moves the kept
 * descriptors into a HOST table (48 bytes each), which the caller copies to the device and hands to ts_conv_wgrad_finish_many: one
 * launch sums every layer's partials (same sums, same order as the per-layer finish).  A training step saves ~95 launches with it. */
typedef struct { const float* part; float* dw; int gx, nitems, cob, Cin, Cout, KT, ciblocks, block0; } ts_wgrad_finish_desc;   /* 48 bytes */
int ts_conv_wgrad_defer(int on);           /* returns the previous setting */
int ts_conv_wgrad_pending(void);
int ts_conv_wgrad_take(void* host_table, size_t capacity_bytes, int* n, int* total_blocks);
int ts_conv_wgrad_finish_many(const void* table, int n, int total_blocks, void* stream);
/* Upper bound (8 | 16 | 32, default 32) on the input-channel chunk -- hence the LDS footprint -- of the
 * convolution launches that follow on this host thread: short chunks when kernels of several streams
 * should share the CUs, long chunks for a lone dependent chain.  Recordable in a plan. */
int ts_conv_set_chunk_cap(int cap);
int ts_conv3d_hw_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                     int B, int Cin, int Cout, int D, int H, int W, int stride, int dilation,
                     int transposed, int act, float act_param,
                     long long in_bstride, long long in_cstride, long long out_bstride,
                     long long out_cstride, const float* addend, long long addend_bstride,
                     void* workspace, size_t workspace_bytes, void* stream);
/* addend (may be NULL): [B, Cout, Ho, Wo] (batch stride addend_bstride elements), added to the raw sum of
 * EVERY depth plane before scale / shift / activation -- the D-invariant part of a convolution over a
 * volume whose leading channels are a broadcast (see ts_block_cost_sampled_warped_fwd). */
/* ABI 6 -- the first (1,3,3) layer of a sampled level without its warped input volume (reference
 * precise.py:88-91, fine.py:96-103 over block_cost.py:47-81).  The warp of block_cost's sampled path is a two-tap interpolation along
 * x whose weights do not depend on the channel, and the convolution contracts over channels, so the two commute:
 *     sum_c W[co][c][t] warp(right)[c][d][p]  ==  lerp(Q[t*Cout + co][row of p], x_p - disp[d][p]),
 *     Q[t*Cout + co] = sum_c W[co][C + c][t] right[c]      (a 1x1 convolution, [B, 9*Cout, H, W], made with ts_conv3d_d_fwd, k = 1)
 * i.e. layer(cat[left x D | warp | corr]) = act(scale * (conv_{W[:, 2C:]}(corr) + T) + shift),
 *     T[co][d][y][x] = base[co][y][x] + sum_{t=(ky,kx)} lerp(Q[t*Cout+co][y+(ky-1)dil], x+(kx-1)dil - disp[d][y+(ky-1)dil][x+(kx-1)dil])
 * with zero for tap pixels outside the image (the convolution's padding) and for columns outside the row (the warp's padding).
 *   corr [B,Cc,D,H,W] (ts_block_cost_sampled_corr_fwd), w_t [Cc][9][CoutPad], scale/shift [CoutPad], q [B,9*Cout,H,W] dense,
 *   disp [B,D,H,W] dense, base [B,Cout,H,W] (the left term conv_{W[:, :C]}(left), or NULL), y [B,Cout,D,H,W].
 *   workspace: ts_conv3d_hw_warp_workspace_bytes(B, Cout, D, H, W) bytes (T).  Strides in elements as for ts_conv3d_hw_fwd. */
size_t ts_conv3d_hw_warp_workspace_bytes(int B, int Cout, int D, int H, int W);
int ts_conv3d_hw_warp_fwd(const float* corr, const float* w_t, const float* scale, const float* shift, float* y,
                          const float* q, const float* disp, const float* base,
                          int B, int Cc, int Cout, int D, int H, int W, int dilation, int act, float act_param,
                          long long in_bstride, long long in_cstride, long long out_bstride, long long out_cstride,
                          long long base_bstride, void* workspace, size_t workspace_bytes, void* stream);

/* scratch ts_conv3d_hw_fwd can use to split a long reduction over more workgroups (0 = never splits at
 * this shape; passing NULL / too little simply disables the split) */
size_t ts_conv3d_hw_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W, int stride, int transposed);
/* The stride-1 (1,3,3) convolution with every fp32 product assembled from bf16 pieces on the bf16 matrix pipe:
 * a = a0 + a1 + a2 (bf16 parts), a*b ~= a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0, fp32 accumulation -- dropped terms are below
 * 2^-24 of the product, every 16-channel chunk summed apart and added to the running sum in fp32; measured max error against an
 * fp64 convolution 0.15e-6 ... 0.26e-6 of the output's magnitude for Cin 16 ... 512 (ts_conv3d_hw_fwd's f32 MFMA chain: 0.3e-6
 * ... 0.7e-6; tests/test_conv_x6_gpu.py), at 3/8 of its matrix time (f32-input MFMA = 1/16 of the bf16 rate on gfx950).
 *   ts_conv3d_hw_x6_supported     1 when the layer can take this path (Cin >= 16, 8 < Cout <= 512, W % 4 == 0, stride 1, dilation 1 | 2)
 *   ts_conv3d_hw_x6_weight_split  w_t (the [Cin][9][CoutPad] array of ts_conv3d_hw_fwd) -> w6, ts_conv3d_hw_x6_weight_bytes
 *                                 bytes: [Cin/16][part 3][tap slot 10][group 2][CoutPad][8] bf16
 *   ts_conv3d_hw_x6_fwd           arguments as ts_conv3d_hw_fwd with stride 1, not transposed; workspace (ABI 5): scratch of
 *                                 ts_conv3d_hw_x6_workspace_bytes bytes for the split-K form of small grids (a grid of a few dozen
 *                                 workgroups is cut into 2 | 4 | 8 slices of the input channels, summed in a fixed order by a second
 *                                 launch); NULL / too little: unsplit */
int ts_conv3d_hw_x6_supported(int Cin, int Cout, int W, int stride, int dilation, int transposed);
size_t ts_conv3d_hw_x6_weight_bytes(int Cin, int Cout);
size_t ts_conv3d_hw_x6_workspace_bytes(int B, int Cin, int Cout, int D, int H, int W);
int ts_conv3d_hw_x6_weight_split(const float* w_t, void* w6, int Cin, int Cout, void* stream);
/* ... the same from the FRAMEWORK's weight in one launch (ABI 9; training re-splits per call): element (ci, co, tap) of the
 * convolution being run = w[ci*stride_ci + co*stride_co + tap*stride_t], taps reversed when flip (input gradient of a stride-1 layer) */
int ts_conv3d_hw_x6_weight_split_from(const float* w, void* w6, int Cin, int Cout, long long stride_ci, long long stride_co,
                                      long long stride_t, int flip, void* stream);
int ts_conv3d_hw_x6_fwd(const float* x, const void* w6, const float* scale, const float* shift, float* y,
                        int B, int Cin, int Cout, int D, int H, int W, int dilation, int act, float act_param,
                        long long in_bstride, long long in_cstride, long long out_bstride, long long out_cstride,
                        const float* addend, long long addend_bstride, void* workspace, size_t workspace_bytes, void* stream);
/* Small-message exchange between the ranks of one node as kernels over peer-mapped (hipIpc, xGMI) memory (ABI 8; csrc/peer.hip):
 * the statistics exchanges of SyncBatchNorm (reference: Lightning's sync_batchnorm=True, projects/TemporalStereo/dist_train.py:94)
 * without a communicator launch per layer, and capturable in a hipGraph.  Set-up: every rank ts_peer_alloc()s its mailbox, the
 * 64-byte handles travel through any out-of-band channel (torch.distributed all_gather), every rank ts_peer_open()s the others and
 * fills a ts_peer_ctx.  All ranks must issue the same sequence of exchanges.  At most ts_peer_max_floats() floats per exchange. */
typedef struct ts_peer_ctx {
  void* region[8];        /* mailbox of rank r as mapped in this process (region[rank]: the local allocation) */
  int rank, world;
} ts_peer_ctx;
size_t ts_peer_region_bytes(void);
int ts_peer_max_floats(void);
int ts_peer_max_ranks(void);
int ts_peer_alloc(void** region, void* handle64);
int ts_peer_open(const void* handle64, void** region);
int ts_peer_close(void* region);
int ts_peer_free(void* region);
/* status: 0, or 1 + r when rank r did not answer an exchange within the bound (that exchange's results and every later one's are
 * undefined until ts_peer_reset).  ts_peer_status synchronises the stream; ts_peer_status_async (ABI 9) copies the word into PINNED
 * host memory behind the queued work, to be read once an event recorded after the call has completed. */
int ts_peer_status(const void* ctx, int* status, void* stream);
int ts_peer_status_async(const void* ctx, int* host_status, void* stream);
/* bound of one wait in milliseconds (default 120 000, or TS_PEER_TIMEOUT_MS); returns the previous one; ms <= 0 only queries */
long long ts_peer_set_timeout_ms(long long ms);
/* clears this rank's flags / sequence number / err word; every rank calls it, nothing in flight, between two barriers */
int ts_peer_reset(const void* ctx, void* stream);
/* dst [world][n] <- every rank's src [n] */
int ts_peer_all_gather(const void* ctx, const float* src, float* dst, int n, void* stream);
/* buf [n] <- (sum over the ranks in rank order: bit-identical everywhere) * (*scale, a device scalar, if not NULL) */
int ts_peer_all_reduce_sum(const void* ctx, float* buf, int n, const float* scale, void* stream);

/* The same bf16-split arithmetic for the STRIDED and TRANSPOSED forms (ABI 7; csrc/conv_x6s.hip), which ts_conv3d_hw_fwd /
 * ts_deconv2d_k4s2_fwd run on the f32-input MFMA.  mode 0: Conv3d (1,3,3) stride 2, padding 1 (Ho = (H-1)/2+1);  mode 1:
 * ConvTranspose3d (1,3,3) stride 2, padding 1, output_padding 1 (Ho = 2H);  mode 2: ConvTranspose2d 4x4 stride 2, padding 1
 * (D = 1, Ho = 2H) -- reference: aggregation/TemporalStereo/module.py:111-184,453-457.  K chunks of 8 input channels, four taps per MFMA.
 *   ts_conv3d_hw_x6s_supported     1 when the layer can take this path (Cin >= 16, Cout <= 512, W % 4 == 0, Wo % 4 == 0)
 *   ts_conv3d_hw_x6s_weight_split  w_t [Cin][9 | 16][w_pad] (the array of ts_conv3d_hw_fwd / ts_deconv2d_k4s2_fwd) -> w6,
 *                                  ts_conv3d_hw_x6s_weight_bytes bytes: [Cin/8][part 3][tap slot 12 | 16][Cout up to 16][8] bf16
 *   ts_conv3d_hw_x6s_fwd           scale / shift [>= Cout] or NULL; strides in elements (x / y may be channel slices) */
int ts_conv3d_hw_x6s_supported(int Cin, int Cout, int H, int W, int mode);
size_t ts_conv3d_hw_x6s_weight_bytes(int Cin, int Cout, int mode);
int ts_conv3d_hw_x6s_weight_split(const float* w_t, void* w6, int Cin, int Cout, int w_pad, int mode, void* stream);
int ts_conv3d_hw_x6s_fwd(const float* x, const void* w6, const float* scale, const float* shift, float* y,
                         int B, int Cin, int Cout, int D, int H, int W, int mode, int act, float act_param,
                         long long in_bstride, long long in_cstride, long long out_bstride, long long out_cstride, void* stream);
int ts_conv3d_d_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                    int B, int Cin, int Cout, int Din, int H, int W, int k, int stride, int dilation,
                    int padding, int transposed, int act, float act_param,
                    long long in_bstride, long long in_cstride, long long out_bstride,
                    long long out_cstride, void* stream);
/* Backward of the two families (training: raw convolutions; BatchNorm / activation stay framework ops in
 * train mode because they need batch statistics).  Geometry arguments always describe the FORWARD call.
 * scale / shift of the forward entries may be NULL (= 1 / 0): that is the raw convolution these invert.
 *   bwd_data  : dy (forward output shape) -> dx (forward input shape).  w_b is the weight re-laid by the host,
 *               [Cout][taps][ts_conv_cout_pad(Cin)]: taps flipped for a stride-1 forward, unflipped for a
 *               stride-2 forward (runs as the transposed form, cropped to the input size) and for a
 *               transposed forward (runs as a stride-2 convolution of dy).
 *   bwd_weight: x, dy -> dw [Cout][Cin][taps] in the framework's layout, OVERWRITTEN.  Every workgroup
 *               leaves its partial sums in `workspace` (ts_conv3d_bwd_weight_workspace_bytes(Cin, Cout,
 *               taps) bytes, taps = 9 | k; contents undefined afterwards) and a second launch adds them in a
 *               fixed order: the result is deterministic (no atomics).  Cout <= 64.
 *               Transposed forward: call with x := dy, dy := x, stride 2 -> [Cin][Cout][taps]
 *               (and size the workspace with the exchanged channel counts). */
size_t ts_conv3d_bwd_weight_workspace_bytes(int Cin, int Cout, int taps);
int ts_conv3d_hw_bwd_data(const float* dy, const float* w_b, float* dx, int B, int Cin, int Cout, int D, int H, int W,
                          int stride, int dilation, int transposed, long long dy_bstride, long long dy_cstride,
                          long long dx_bstride, long long dx_cstride, void* stream);
int ts_conv3d_hw_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int D, int H, int W,
                            int stride, int dilation, long long x_bstride, long long x_cstride, long long dy_bstride,
                            long long dy_cstride, void* workspace, size_t workspace_bytes, void* stream);
int ts_conv3d_d_bwd_data(const float* dy, const float* w_b, float* dx, int B, int Cin, int Cout, int Din, int H, int W,
                         int k, int stride, int dilation, int padding, int transposed, long long dy_bstride,
                         long long dy_cstride, long long dx_bstride, long long dx_cstride, void* stream);
int ts_conv3d_d_bwd_weight(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int Din, int H, int W,
                           int k, int stride, int dilation, int padding, long long x_bstride, long long x_cstride,
                           long long dy_bstride, long long dy_cstride, void* workspace, size_t workspace_bytes,
                           void* stream);
/* ResidualBlock3D up-steps: out = act(trilinear_align_corners(a -> (D,H,W)) + add)  module.py:285-295 */
int ts_resize3d_add_act_fwd(const float* a, const float* add, float* out, int B, int C, int Da, int Ha, int Wa,
                            int D, int H, int W, int act, long long a_bstride, long long a_cstride,
                            long long add_bstride, long long add_cstride, long long out_bstride,
                            long long out_cstride, void* stream);
/* PyramidFusion pooling branch: avg_pool3d + max_pool3d, kernel 5, stride 1, padding 2  module.py:415-417 */
int ts_pool3d5_avgmax_fwd(const float* x, float* out_avg, float* out_max, int B, int C, int D, int H, int W,
                          long long x_bstride, long long x_cstride, long long avg_bstride, long long avg_cstride,
                          long long max_bstride, long long max_cstride, void* stream);
/* temporal merge: past_conv(1->C)+BN+SiLU on the K memory costs, cat along D, stable sort of the
 * D0+K candidates, gather of the volume  coarse.py:84-105 / fine.py:105-122.  sample NULL = 0..D0-1;
 * mem_sample / mem_cost NULL = zeros. */
int ts_merge_candidates_fwd(const float* volume, const float* sample, const float* mem_sample, const float* mem_cost,
                            const float* past_w, const float* past_scale, const float* past_shift,
                            float* out_sample, float* out_volume, int B, int C, int D0, int K, int H, int W,
                            long long vol_bstride, long long vol_cstride, long long out_bstride,
                            long long out_cstride, void* stream);
/* K5 upsamplers: ConvexUpsample.forward module.py:336-353 (mask [B,9*f*f,H,W]); UNet.upsample :468-482
 * (mask [B,9,Ho,Wo]); ConvTranspose2d(4, stride 2, padding 1) of UNet :453-457 (w_t [Cin][4][4][CoutPad],
 * CoutPad = ts_conv_cout_pad(Cout) = 8 | 16 | 32 as for every other convolution entry since ABI 9; up to ABI 8 this entry
 * alone wanted 16 | 32, which its own training-side caller did not honour for Cout <= 8); bilinear align_corners resize
 * with a value scale. */
int ts_convex_upsample_fwd(const float* mask, const float* disp, float* out, int B, int H, int W, int factor,
                           float disp_scale, void* stream);
/* ts_convex_upsample_fwd + ts_range_candidates_fwd (on the upsampled map) as one launch */
int ts_convex_upsample_candidates_fwd(const float* mask, const float* disp, float* out, float* low, float* high,
                                      float* candidates, int B, int H, int W, int factor, float disp_scale, float range,
                                      int channel_offset, int channels_total, void* stream);
int ts_unet_upsample_fwd(const float* mask, const float* disp, float* out, int B, int h, int w, int Ho, int Wo,
                         void* stream);
/* Backward of the two upsamplers (training form).  grad_out has the output's shape; grad_mask / grad_disp are OVERWRITTEN
 * (ts_convex_upsample_bwd: either may be NULL; ts_unet_upsample_bwd: grad_disp may be NULL; workspace B*9*Ho*Wo floats).
 * Gather formulations: deterministic, no atomics. */
int ts_convex_upsample_bwd(const float* mask, const float* disp, const float* grad_out, float* grad_mask, float* grad_disp,
                           int B, int H, int W, int factor, float disp_scale, void* stream);
int ts_unet_upsample_bwd(const float* mask, const float* disp, const float* grad_out, float* grad_mask, float* grad_disp,
                         void* workspace, int B, int h, int w, int Ho, int Wo, void* stream);
int ts_deconv2d_k4s2_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                         int B, int Cin, int Cout, int H, int W, int act, long long out_bstride, void* stream);
int ts_resize_bilinear_fwd(const float* x, float* out, int B, int C, int h, int w, int Ho, int Wo, float value_scale,
                           long long out_bstride, void* stream);
/* two same-shaped dense maps in one launch (the top-k memory's candidates and costs, precise.py:100-103, coarse.py:91-96) */
int ts_resize_bilinear_pair_fwd(const float* x0, const float* x1, float* out0, float* out1, int B, int C, int h, int w,
                                int Ho, int Wo, float value_scale0, float value_scale1, void* stream);
/* search range of the next level and its five candidates from an upsampled disparity:
 * low = d - range, high = d + range (aggregation/TemporalStereo/TemporalStereo.py:110,119);
 * candidates[:, off+i] = |high-low| * {0,3,4,5,8}/8 + min(low,high)  (fine.py:82-87, precise.py:73-78) */
int ts_range_candidates_fwd(const float* disp, float* low, float* high, float* candidates, int B, int H, int W,
                            float range, int channel_offset, int channels_total, void* stream);

/* Backward of the element stages (training; dense NCDHW tensors, outputs OVERWRITTEN, fp32 atomics where a
 * gradient is a scatter -- the same non-deterministic order as the framework's own backward kernels):
 *   resize3d_add_act_bwd : d out / d(a, add) of ts_resize3d_add_act_fwd (grad_add may be NULL)
 *   pool3d5_avgmax_bwd   : grad_x = box5^3(grad_avg)/125 + grad_max routed to each window's arg-max
 *                          (first occurrence in (d,y,x) order, as max_pool3d_with_indices)
 *   merge_candidates_bwd : the sort + gather half of ts_merge_candidates_fwd (its K = 0 form):
 *                          grad_volume[:, j] = grad_out_volume[:, rank_j], likewise the samples */
int ts_resize3d_add_act_bwd(const float* a, const float* add, const float* grad_out, float* grad_a, float* grad_add,
                            int B, int C, int Da, int Ha, int Wa, int D, int H, int W, int act, void* stream);
int ts_pool3d5_avgmax_bwd(const float* x, const float* grad_avg, const float* grad_max, float* grad_x,
                          int B, int C, int D, int H, int W, void* stream);
int ts_merge_candidates_bwd(const float* sample, const float* grad_out_volume, const float* grad_out_sample,
                            float* grad_volume, float* grad_sample, int B, int C, int DT, int H, int W, void* stream);

/* dst[r * dst_pitch + c] = src[r * src_pitch + c] (elements): writes a tensor into a channel slice of
 * another -- the torch.cat / slice.copy_ of precise.py:60-63 and module.py:486-489 without torch. */
/* Channel splice of the backbone's per-frame feature memory (architecture/modeling/backbone/TemporalStereo.py:
 * 183-197: `torch.cat([memory, input[:, mc:]], 1)` in front of every residual block):
 *   out[b][c] = c < mc ? first[b][c] : second[b][c];  first [B, mc, N] (NULL = zeros: the adjoint towards `input`), second / out
 *   [B, C, N]; *_bstride in elements. */
int ts_channel_splice_fwd(const float* first, const float* second, float* out, int B, int C, int mc, long long N,
                          long long first_bstride, long long second_bstride, long long out_bstride, void* stream);
int ts_copy_rows_fwd(const float* src, float* dst, long long rows, long long row_elems, long long src_pitch,
                     long long dst_pitch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training-side glue (csrc/train_ops.hip): the element-wise steps around the levels that the framework otherwise runs
 * as ~280 tiny launches per training step, the pieces that run the UNet decoder's ConvTranspose2d(4, stride 2, padding 1)
 * (module.py:453-457) on the convolution kernels in training, and the optimizer step.
 *   candidates_in_range : candidates[:, off+i] = |high-low| * {0,3,4,5,8}/8 + min(low,high) (fine.py:82-87, precise.py:73-78)
 *                         from separate low / high maps [B,1,H,W]; bwd: d/d low, d/d high (|.|' = sign, min' split at ties)
 *   offset_head         : y = clamp(tanh(x/100), -1, 1) * delta  (PredictionHeads.regress_offset, module.py:384-390)
 *   space_to_depth2     : z[b][(py*2+px)*C + c][y][x] = x[b][c][2y+py][2x+px]   (x [B,C,2H,2W] -> z [B,4C,H,W])
 *   deconv2d_k4s2_weight_to_conv3 : w [Cin][Cout][4][4] -> out [(4*Cout)][9][cin_pad], the ts_conv3d_hw_fwd weight of the 3x3
 *                         stride-1 convolution of space_to_depth2(dy) that is d/dx of the transposed convolution
 *   deconv2d_k4s2_wgrad_from_conv3: dw3 [Cin][4*Cout][9] (ts_conv3d_hw_bwd_weight of that convolution) -> dw [Cin][Cout][4][4]
 *   clip_rmsprop_step   : clip_grad_norm_(max_norm) + RMSprop(lr, alpha, eps; no momentum / centring / weight decay) over
 *                         a DEVICE table of ts_opt_entry; two launches, deterministic; workspace's last float <- total norm
 * ---------------------------------------------------------------------------------------- */
typedef struct ts_opt_entry { float* param; const float* grad; float* square_avg; long long n; } ts_opt_entry;
int ts_candidates_in_range_fwd(const float* low, const float* high, float* candidates, int B, int H, int W,
                               int channel_offset, int channels_total, void* stream);
int ts_candidates_in_range_bwd(const float* low, const float* high, const float* grad_candidates, float* grad_low,
                               float* grad_high, int B, int H, int W, int channel_offset, int channels_total, void* stream);
int ts_offset_head_fwd(const float* x, float* y, long long n, float delta, void* stream);
int ts_offset_head_bwd(const float* x, const float* grad_y, float* grad_x, long long n, float delta, void* stream);
int ts_space_to_depth2_fwd(const float* x, float* z, int B, int C, int H, int W, void* stream);
int ts_deconv2d_k4s2_weight_to_conv3(const float* w, float* out, int Cin, int Cout, int cin_pad, void* stream);
int ts_deconv2d_k4s2_wgrad_from_conv3(const float* dw3, float* dw, int Cin, int Cout, void* stream);
/* eval-mode BatchNorm folded into the convolution epilogue for a whole step in one launch: table = n ts_bn_fold_entry in DEVICE memory;
 * scale[c] = gamma / sqrt(var + eps), shift[c] = beta + (bias - mean) * scale for c < C, 0 for C <= c < pad (gamma / beta / bias may be NULL;
 * mean / var NULL (ABI 9) = no BatchNorm behind the convolution: scale 1, shift = bias) */
typedef struct ts_bn_fold_entry { const float* gamma; const float* beta; const float* mean; const float* var; const float* bias;
                                  float* scale; float* shift; int C, pad; } ts_bn_fold_entry;
int ts_bn_fold_many(const void* table, int n, float eps, void* stream);
size_t ts_clip_rmsprop_workspace_bytes(int n_tensors);
int ts_clip_rmsprop_step(const void* table, int n_tensors, float max_norm, float lr, float alpha, float eps,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Native replay runtime (no reference counterpart; the reference runs eagerly under PyTorch).
 * A plan is the recorded sequence of launching calls of one pass -- device pointers, shapes and
 * streams baked in, the contract of a CUDA graph with static buffers -- re-issued by one host call.
 * ts_plan_add_call: `name` is one of the int-returning launch entry points above, `words` holds its
 * arguments in order, one 64-bit word each (ints / floats in the low bytes).
 * ts_stream_fork: to_stream waits for the work enqueued so far on from_stream.
 * ---------------------------------------------------------------------------------------- */
typedef struct ts_plan ts_plan;
ts_plan* ts_plan_create(void);
void ts_plan_destroy(ts_plan* plan);
int ts_plan_length(const ts_plan* plan);
int ts_plan_add_call(ts_plan* plan, const char* name, const unsigned long long* words, int n_words);
int ts_plan_run(ts_plan* plan);
int ts_stream_fork(void* from_stream, void* to_stream);
/* Named events (slot 0..255): record on one stream now, wait from another stream in a LATER call (an edge that spans
 * replays: "the pass that last used these buffers is done").  Waiting on a never-recorded slot is a no-op. */
int ts_event_record(int slot, void* stream);
int ts_event_wait(int slot, void* stream);

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
