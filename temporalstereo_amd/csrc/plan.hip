// Native replay runtime: a recorded pass (the exact sequence of C-ABI launches one aggregation pass
// makes, with its device pointers, shapes and streams) is kept as a flat array of calls and re-issued
// by one host call.
//
// Why: at batch 1 the pass is ~125 kernels of 5-150 us.  Issued from Python (tensor allocation, ctypes
// marshalling of ~20 arguments per launch) the host needs ~1.8 ms per pass -- as long as the device.
// hipGraph replays the captured pass no faster than its kernels run back to back and serialises the
// two-stream overlap (ROCm 7.2: 2.08 ms single-stream, 4.7 ms with a forked capture), so the pass is
// replayed natively instead: the host cost drops to the hipLaunchKernel calls themselves and the
// streams stay ordinary streams (priorities and cross-stream overlap keep working).
//
// The reference has no counterpart (it runs eagerly under PyTorch); the closest notion is a CUDA
// graph with static input/output buffers, and the contract is the same: pointers are baked in.
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ts_common.hpp"

namespace {

constexpr int kMaxWords = 32;

template <class T>
T unpack(unsigned long long w) {
  static_assert(sizeof(T) <= sizeof(unsigned long long), "argument wider than a plan word");
  T v;
  std::memcpy(&v, &w, sizeof(T));          // little-endian: the low bytes hold ints and floats
  return v;
}

template <class... A, size_t... I>
int call_with(int (*f)(A...), const unsigned long long* w, std::index_sequence<I...>) {
  return f(unpack<A>(w[I])...);
}
template <class... A>
int call_with(int (*f)(A...), const unsigned long long* w) {
  return call_with(f, w, std::index_sequence_for<A...>{});
}
template <class... A>
constexpr int arity(int (*)(A...)) { return static_cast<int>(sizeof...(A)); }

struct Entry {
  const char* name;
  int (*thunk)(const unsigned long long*);
  int nargs;
};

#define TS_PLAN_OP(fn) Entry{#fn, [](const unsigned long long* w) { return call_with(&fn, w); }, arity(&fn)}

// every launching entry point of include/ts_hip.h
const Entry kTable[] = {
    TS_PLAN_OP(ts_block_cost_int_fwd),      TS_PLAN_OP(ts_block_cost_sampled_fwd),
    TS_PLAN_OP(ts_block_cost_sampled_warped_fwd),
    TS_PLAN_OP(ts_block_cost_sampled_corr_fwd), TS_PLAN_OP(ts_conv3d_hw_warp_fwd),
    TS_PLAN_OP(ts_cat_fms_fwd),             TS_PLAN_OP(ts_dif_fms_fwd),             TS_PLAN_OP(ts_inverse_warp_3d_fwd),
    TS_PLAN_OP(ts_block_cost_int_bwd),      TS_PLAN_OP(ts_block_cost_sampled_bwd),
    TS_PLAN_OP(ts_topk_softargmax_fwd),     TS_PLAN_OP(ts_topk_softargmax_bwd),
    TS_PLAN_OP(ts_softargmin_fwd),          TS_PLAN_OP(ts_softargmin_bwd),
    TS_PLAN_OP(ts_argmax_select_fwd),       TS_PLAN_OP(ts_softsplat_sum_fwd),
    TS_PLAN_OP(ts_softsplat_sum_fwd_deterministic), TS_PLAN_OP(ts_softsplat_sum_bwd_input), TS_PLAN_OP(ts_softsplat_sum_bwd_flow),
    TS_PLAN_OP(ts_softsplat_softmax_fwd),   TS_PLAN_OP(ts_project_to_3d_fwd),
    TS_PLAN_OP(ts_conv3d_hw_fwd),           TS_PLAN_OP(ts_conv3d_d_fwd),
    TS_PLAN_OP(ts_conv3d_hw_bwd_data),      TS_PLAN_OP(ts_conv3d_hw_bwd_weight),
    TS_PLAN_OP(ts_conv3d_d_bwd_data),       TS_PLAN_OP(ts_conv3d_d_bwd_weight),
    TS_PLAN_OP(ts_resize3d_add_act_fwd),    TS_PLAN_OP(ts_pool3d5_avgmax_fwd),
    TS_PLAN_OP(ts_merge_candidates_fwd),    TS_PLAN_OP(ts_convex_upsample_fwd),
    TS_PLAN_OP(ts_convex_upsample_candidates_fwd),
    TS_PLAN_OP(ts_unet_upsample_fwd),       TS_PLAN_OP(ts_deconv2d_k4s2_fwd),
    TS_PLAN_OP(ts_resize_bilinear_fwd),     TS_PLAN_OP(ts_range_candidates_fwd),
    TS_PLAN_OP(ts_copy_rows_fwd),           TS_PLAN_OP(ts_stream_fork),
    TS_PLAN_OP(ts_event_record),            TS_PLAN_OP(ts_event_wait),
    TS_PLAN_OP(ts_resize3d_add_act_bwd),    TS_PLAN_OP(ts_pool3d5_avgmax_bwd),
    TS_PLAN_OP(ts_merge_candidates_bwd),
    TS_PLAN_OP(ts_reproject_memory_fwd),
    TS_PLAN_OP(ts_resize_bilinear_pair_fwd),
    TS_PLAN_OP(ts_calib_stream),             TS_PLAN_OP(ts_conv_set_chunk_cap),
    TS_PLAN_OP(ts_wasserstein_loss_fwd),     TS_PLAN_OP(ts_wasserstein_loss_bwd),
    TS_PLAN_OP(ts_disp_smooth_l1_fwd),       TS_PLAN_OP(ts_disp_smooth_l1_bwd),
    TS_PLAN_OP(ts_correlation_fwd),          TS_PLAN_OP(ts_correlation_bwd),
    TS_PLAN_OP(ts_bn_stats_fwd),             TS_PLAN_OP(ts_bn_apply_act_fwd),
    TS_PLAN_OP(ts_bn_act_bwd_reduce),        TS_PLAN_OP(ts_bn_act_bwd_apply),
    TS_PLAN_OP(ts_convex_upsample_bwd),      TS_PLAN_OP(ts_unet_upsample_bwd),
    TS_PLAN_OP(ts_conv_weight_layout),      TS_PLAN_OP(ts_conv_weight_layout_many),
    TS_PLAN_OP(ts_conv3d_hw_x6_fwd),        TS_PLAN_OP(ts_conv3d_hw_x6_weight_split), TS_PLAN_OP(ts_conv3d_hw_x6_weight_split_from),
    TS_PLAN_OP(ts_conv3d_hw_x6s_fwd),       TS_PLAN_OP(ts_conv3d_hw_x6s_weight_split),
    TS_PLAN_OP(ts_peer_all_gather),         TS_PLAN_OP(ts_peer_all_reduce_sum),
    TS_PLAN_OP(ts_bn_train_fwd),            TS_PLAN_OP(ts_bn_train_bwd),
    TS_PLAN_OP(ts_channel_splice_fwd),
    TS_PLAN_OP(ts_candidates_in_range_fwd), TS_PLAN_OP(ts_candidates_in_range_bwd),
    TS_PLAN_OP(ts_offset_head_fwd),         TS_PLAN_OP(ts_offset_head_bwd),
    TS_PLAN_OP(ts_space_to_depth2_fwd),
    TS_PLAN_OP(ts_deconv2d_k4s2_weight_to_conv3), TS_PLAN_OP(ts_deconv2d_k4s2_wgrad_from_conv3),
    TS_PLAN_OP(ts_clip_rmsprop_step),       TS_PLAN_OP(ts_bn_sync_merge),
    TS_PLAN_OP(ts_bn_fold_many),
    TS_PLAN_OP(ts_conv_weight_layout_many2), TS_PLAN_OP(ts_conv_wgrad_finish_many), TS_PLAN_OP(ts_channel_sum_fwd),
    TS_PLAN_OP(ts_block_cost_sampled_warped_bwd),
};

struct Call {
  const Entry* op;
  unsigned long long w[kMaxWords];
};

__global__ void __launch_bounds__(256)
copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, long long row_elems,
                 long long src_pitch, long long dst_pitch) {
  const long long n = rows * row_elems;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long r = i / row_elems, c = i - r * row_elems;
    dst[r * dst_pitch + c] = src[r * src_pitch + c];
  }
}

// rows of whole, 16-byte aligned quads (the engine's feature copies: one row of 2 M floats): a row per blockIdx.y, 16 bytes per
// lane, no division per element (the scalar form above spends a 64-bit division on every float: 6.4 us for 8 MB where this takes 4.4)
__global__ void __launch_bounds__(256)
copy_rows4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long row_quads, long long src_pitch_q,
                  long long dst_pitch_q) {
  const float4* s = src + blockIdx.y * src_pitch_q;
  float4* d = dst + blockIdx.y * dst_pitch_q;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < row_quads;
       i += static_cast<long long>(gridDim.x) * blockDim.x)
    d[i] = s[i];
}

}  // namespace

struct ts_plan {
  std::vector<Call> calls;
};

extern "C" ts_plan* ts_plan_create(void) { return new (std::nothrow) ts_plan(); }

extern "C" void ts_plan_destroy(ts_plan* plan) { delete plan; }

extern "C" int ts_plan_length(const ts_plan* plan) { return plan ? static_cast<int>(plan->calls.size()) : -1; }

extern "C" int ts_plan_add_call(ts_plan* plan, const char* name, const unsigned long long* words, int n_words) {
  TS_REQUIRE_PTR(plan); TS_REQUIRE_PTR(name);
  TS_REQUIRE(n_words >= 0 && n_words <= kMaxWords, TS_ERR_SHAPE, "plan: %d argument words", n_words);
  TS_REQUIRE(n_words == 0 || words != nullptr, TS_ERR_NULL, "plan: words is NULL");
  for (const Entry& e : kTable) {
    if (std::strcmp(e.name, name) != 0) continue;
    TS_REQUIRE(e.nargs == n_words, TS_ERR_SHAPE, "plan: %s takes %d arguments, got %d", name, e.nargs, n_words);
    Call c;
    c.op = &e;
    std::memset(c.w, 0, sizeof(c.w));
    if (n_words) std::memcpy(c.w, words, sizeof(unsigned long long) * n_words);
    plan->calls.push_back(c);
    return TS_OK;
  }
  return ts::fail(TS_ERR_UNSUPPORTED, "plan: %s is not a launching entry point", name);
}

// re-issues every recorded call in order; stops at (and returns) the first failure
extern "C" int ts_plan_run(ts_plan* plan) {
  TS_REQUIRE_PTR(plan);
  for (const Call& c : plan->calls) {
    const int rc = c.op->thunk(c.w);
    if (rc != TS_OK) return rc;
  }
  return TS_OK;
}

// `to_stream` waits for everything enqueued so far on `from_stream` (fork and join are the same edge)
extern "C" int ts_stream_fork(void* from_stream, void* to_stream) {
  constexpr int kRing = 64;                       // a wait is enqueued right after its record, so reuse is safe
  constexpr int kMaxDev = 16;                     // an event belongs to the device it was created on: one ring per device
  struct Ring { hipEvent_t ev[kRing]; int made = 0, next = 0; };
  static thread_local Ring rings[kMaxDev];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "stream_fork: %s", hipGetErrorString(e));
  TS_REQUIRE(dev >= 0 && dev < kMaxDev, TS_ERR_UNSUPPORTED, "stream_fork: device %d", dev);
  Ring& r = rings[dev];
  if (r.made < kRing && r.next == r.made) {
    e = hipEventCreateWithFlags(&r.ev[r.made], hipEventDisableTiming);
    if (e != hipSuccess) return ts::fail(static_cast<int>(e), "stream_fork: %s", hipGetErrorString(e));
    ++r.made;
  }
  hipEvent_t ev = r.ev[r.next];
  r.next = (r.next + 1) % kRing;
  e = hipEventRecord(ev, ts::as_stream(from_stream));
  if (e == hipSuccess) e = hipStreamWaitEvent(ts::as_stream(to_stream), ev, 0);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "stream_fork: %s", hipGetErrorString(e));
  return TS_OK;
}

// Named events (slots 0..255, created on first use): an edge whose record and wait are issued by DIFFERENT calls --
// e.g. "the pass that last used these buffers has finished" recorded at the end of one replay and waited on at the
// start of the replay after next (two passes in flight on double-buffered plans).  Waiting on a slot that was never
// recorded is a no-op.
namespace {
hipEvent_t g_slot[256];
bool g_slot_made[256];
bool g_slot_recorded[256];
std::mutex g_slot_lock;
}  // namespace

extern "C" int ts_event_record(int slot, void* stream) {
  TS_REQUIRE(slot >= 0 && slot < 256, TS_ERR_SHAPE, "event_record: slot %d", slot);
  std::lock_guard<std::mutex> hold(g_slot_lock);
  if (!g_slot_made[slot]) {
    hipError_t e = hipEventCreateWithFlags(&g_slot[slot], hipEventDisableTiming);
    if (e != hipSuccess) return ts::fail(static_cast<int>(e), "event_record: %s", hipGetErrorString(e));
    g_slot_made[slot] = true;
  }
  hipError_t e = hipEventRecord(g_slot[slot], ts::as_stream(stream));
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "event_record: %s", hipGetErrorString(e));
  g_slot_recorded[slot] = true;
  return TS_OK;
}

extern "C" int ts_event_wait(int slot, void* stream) {
  TS_REQUIRE(slot >= 0 && slot < 256, TS_ERR_SHAPE, "event_wait: slot %d", slot);
  std::lock_guard<std::mutex> hold(g_slot_lock);
  if (!g_slot_recorded[slot]) return TS_OK;
  hipError_t e = hipStreamWaitEvent(ts::as_stream(stream), g_slot[slot], 0);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "event_wait: %s", hipGetErrorString(e));
  return TS_OK;
}

// dst[r * dst_pitch + c] = src[r * src_pitch + c]: channel-slice concatenation without torch
// Channel splice of the backbone's feature memory (architecture/modeling/backbone/TemporalStereo.py:183-197: the first
// int(C * memory_percent) channels of a block's input are replaced by the previous frame's, `torch.cat([memory, input2], 1)`):
//   out[b][c] = c < mc ? first[b][c] : second[b][c]      first [B, mc, N] (NULL: zeros), second / out [B, C, N]
namespace {
__global__ void __launch_bounds__(256)
channel_splice_kernel(const float* __restrict__ first, const float* __restrict__ second, float* __restrict__ out, int C, int mc,
                      long long N, long long fb, long long sb, long long ob) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* src = c < mc ? (first ? first + b * fb + static_cast<long long>(c) * N : nullptr) : second + b * sb + static_cast<long long>(c) * N;
  float* dst = out + b * ob + static_cast<long long>(c) * N;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < N; i += static_cast<long long>(gridDim.x) * blockDim.x)
    dst[i] = src ? src[i] : 0.f;
}
}  // namespace

extern "C" int ts_channel_splice_fwd(const float* first, const float* second, float* out, int B, int C, int mc, long long N,
                                     long long first_bstride, long long second_bstride, long long out_bstride, void* stream) {
  TS_REQUIRE(B > 0 && C > 0 && N > 0 && mc >= 0 && mc <= C && B <= 65535 && C <= 65535, TS_ERR_SHAPE, "channel_splice: bad size");
  TS_REQUIRE_PTR(second); TS_REQUIRE_PTR(out);
  long long blocks = (N + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(channel_splice_kernel, dim3(static_cast<unsigned>(blocks), C, B), dim3(256), 0, ts::as_stream(stream), first, second,
                     out, C, mc, N, first_bstride, second_bstride, out_bstride);
  return ts::launched("channel_splice_kernel");
}

extern "C" int ts_copy_rows_fwd(const float* src, float* dst, long long rows, long long row_elems, long long src_pitch,
                                long long dst_pitch, void* stream) {
  TS_REQUIRE(rows > 0 && row_elems > 0 && src_pitch >= row_elems && dst_pitch >= row_elems, TS_ERR_SHAPE,
             "copy_rows: bad geometry");
  TS_REQUIRE_PTR(src); TS_REQUIRE_PTR(dst);
  if (rows <= 65535 && row_elems % 4 == 0 && src_pitch % 4 == 0 && dst_pitch % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) % 16 == 0) {
    const long long rq = row_elems / 4;
    long long bx = (rq + 255) / 256;
    const long long cap = (8192 + rows - 1) / rows;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL(copy_rows4_kernel, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(rows)), dim3(256), 0, ts::as_stream(stream),
                       reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), rq, src_pitch / 4, dst_pitch / 4);
    return ts::launched("copy_rows4_kernel");
  }
  long long blocks = (rows * row_elems + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, ts::as_stream(stream), src, dst,
                     rows, row_elems, src_pitch, dst_pitch);
  return ts::launched("copy_rows_kernel");
}
