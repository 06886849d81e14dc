// Small-message exchange between the ranks of ONE node as plain kernels over peer-mapped memory (xGMI): what SyncBatchNorm needs
// 176 times per training step -- an all_gather of 2C+1 floats forward, an all_reduce of 2C floats backward per layer
// (temporalstereo_amd/functional.py _ConvBNAct; the reference gets both from torch's SyncBatchNorm under Lightning's
// sync_batchnorm=True, projects/TemporalStereo/dist_train.py:94).
//
// Why not RCCL for these: (1) a collective of a few hundred bytes is all launch latency -- ~15-20 us each through the communicator's
// proxy against one ~3 us kernel here; (2) a kernel is CAPTURABLE: with the exchange inside the hipGraph the replayed training step
// (train.py, graph=True) is legal for world > 1, which rounds 2-3 had to refuse.  The gradient all-reduce (4.2 MB, once per step) stays
// on RCCL: that is what a ring over the xGMI links is for.
//
// Mechanism.  Every rank owns a mailbox in fine-grained device memory, mapped into every peer through hipIpc handles:
//     [flags  S x 8 u32][seq, err ...][mail  S x 8 x kPeerMaxN floats]         S = kPeerSlots exchange slots, 8 = ranks of a node
// Exchange number q (a device-side counter: replays of a captured graph advance it like eager launches do) uses slot q % S.  A rank
// stores its record into mail[slot][me] of EVERY rank, fences at system scope, raises flags[slot][me] = q + 1 there (release), then
// waits until its own flags[slot][r] == q + 1 for all r (acquire) and reads the records in rank order -- so every rank sums in the
// same order and gets bit-identical results (a ring all-reduce does not promise that).  A rank cannot finish exchange q + 1 before
// every peer has posted q + 1, i.e. has finished READING q: with S >= 2 a slot is never overwritten while someone still reads it.
// The wait is bounded (ts_peer_set_timeout_ms; default 120 s, TS_PEER_TIMEOUT_MS in the environment -- the order of a collective
// watchdog: ranks of a real job drift apart by seconds around checkpoints, validation and data-loader stalls, and a rank that
// merely arrives late must find its peers still waiting): a peer that never comes sets the err word instead of hanging the queue.
// After that the group is DEAD -- later exchanges do not wait and their results are undefined -- until ts_peer_reset: the host
// side reads the word every step (ts_peer_status_async, one step late, no synchronisation) and stops training (train.TrainStep).
#include <cstdlib>
#include <cstring>

#include "ts_common.hpp"

namespace {

constexpr int kPeerSlots = 4, kPeerRanks = 8, kPeerMaxN = 1024;
constexpr size_t kFlagBytes = 256, kCtlBytes = 256;                      // flags [S][8] u32 = 128 B; control words: seq, err
constexpr size_t kMailBytes = static_cast<size_t>(kPeerSlots) * kPeerRanks * kPeerMaxN * sizeof(float);
constexpr size_t kRegionBytes = kFlagBytes + kCtlBytes + kMailBytes;

struct PeerCtx {                      // == ts_peer_ctx of include/ts_hip.h
  void* region[kPeerRanks];           // every rank's mailbox as mapped HERE (region[rank] is the local allocation)
  int rank, world;
};

__device__ __forceinline__ unsigned* flags_of(void* region) { return reinterpret_cast<unsigned*>(region); }
__device__ __forceinline__ unsigned* ctl_of(void* region) { return reinterpret_cast<unsigned*>(static_cast<char*>(region) + kFlagBytes); }
__device__ __forceinline__ float* mail_of(void* region) { return reinterpret_cast<float*>(static_cast<char*>(region) + kFlagBytes + kCtlBytes); }

// one workgroup of 256 threads.  MODE 0: dst [world][n] = every rank's src (rank order);  MODE 1: buf[i] = (sum over ranks of buf[i], rank
// order) * (scale ? *scale : 1)
template <int MODE>
__global__ void __launch_bounds__(256)
peer_exchange_kernel(const PeerCtx ctx, const float* src, float* dst, int n, const float* __restrict__ scale,
                     unsigned long long timeout_ticks) {     // MODE 1: src == dst
  __shared__ unsigned s_q;
  unsigned* ctl = ctl_of(ctx.region[ctx.rank]);
  if (threadIdx.x == 0) s_q = ctl[0];
  __syncthreads();
  const unsigned q = s_q, slot = q % kPeerSlots, tag = q + 1u;
  // 1. my record into everyone's mailbox (peer stores over xGMI; my own copy too, so that the read phase is uniform)
  for (int r = 0; r < ctx.world; ++r) {
    float* m = mail_of(ctx.region[r]) + (static_cast<size_t>(slot) * kPeerRanks + ctx.rank) * kPeerMaxN;
    for (int i = threadIdx.x; i < n; i += 256) __builtin_nontemporal_store(src[i], m + i);
  }
  __threadfence_system();
  __syncthreads();
  // 2. raise my flag everywhere, 3. wait for everyone's flag here
  if (threadIdx.x < ctx.world) {
    const int r = threadIdx.x;
    __hip_atomic_store(flags_of(ctx.region[r]) + slot * kPeerRanks + ctx.rank, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = flags_of(ctx.region[ctx.rank]) + slot * kPeerRanks + r;
    const unsigned long long t0 = wall_clock64();                          // 100 MHz
    const bool dead = ctl[1] != 0u;                                        // an earlier exchange timed out: do not wait again (176 per step)
    while (!dead && __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
      if (wall_clock64() - t0 > timeout_ticks) {                            // a peer is not coming
        __hip_atomic_store(ctl + 1, 1u + static_cast<unsigned>(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __threadfence_system();
  // 4. read the records in rank order
  const float* mail = mail_of(ctx.region[ctx.rank]) + static_cast<size_t>(slot) * kPeerRanks * kPeerMaxN;
  if (MODE == 0) {
    for (int r = 0; r < ctx.world; ++r)
      for (int i = threadIdx.x; i < n; i += 256)
        dst[static_cast<size_t>(r) * n + i] = __builtin_nontemporal_load(mail + static_cast<size_t>(r) * kPeerMaxN + i);
  } else {
    const float sc = scale ? *scale : 1.f;
    for (int i = threadIdx.x; i < n; i += 256) {
      float s = 0.f;
      for (int r = 0; r < ctx.world; ++r) s += __builtin_nontemporal_load(mail + static_cast<size_t>(r) * kPeerMaxN + i);
      dst[i] = s * sc;
    }
  }
  if (threadIdx.x == 0) ctl[0] = q + 1u;
}

// Bound of one wait in ticks of the 100 MHz wall clock.  Process-wide like the chunk cap of the convolutions: a captured graph keeps
// the value it was captured with.
unsigned long long g_timeout_ticks = 0;
unsigned long long timeout_ticks() {
  if (g_timeout_ticks == 0) {
    long long ms = 120000;
    if (const char* e = getenv("TS_PEER_TIMEOUT_MS")) { const long long v = atoll(e); if (v > 0) ms = v; }
    g_timeout_ticks = static_cast<unsigned long long>(ms) * 100000ull;
  }
  return g_timeout_ticks;
}

int check_ctx(const PeerCtx* c, int n, const char* what) {
  TS_REQUIRE_PTR(c);
  TS_REQUIRE(c->world >= 1 && c->world <= kPeerRanks && c->rank >= 0 && c->rank < c->world, TS_ERR_SHAPE, "%s: rank %d of %d", what, c->rank, c->world);
  TS_REQUIRE(n > 0 && n <= kPeerMaxN, TS_ERR_SHAPE, "%s: %d floats (1..%d)", what, n, kPeerMaxN);
  for (int r = 0; r < c->world; ++r) TS_REQUIRE(c->region[r] != nullptr, TS_ERR_NULL, "%s: mailbox of rank %d is not mapped", what, r);
  return TS_OK;
}

}  // namespace

extern "C" size_t ts_peer_region_bytes(void) { return kRegionBytes; }
extern "C" int ts_peer_max_floats(void) { return kPeerMaxN; }
extern "C" int ts_peer_max_ranks(void) { return kPeerRanks; }

// Allocate this rank's mailbox (fine-grained device memory: peers' stores become visible to a running kernel), zeroed, and export it:
// handle64 receives the 64-byte hipIpcMemHandle_t every peer passes to ts_peer_open.
extern "C" int ts_peer_alloc(void** region, void* handle64) {
  TS_REQUIRE_PTR(region); TS_REQUIRE_PTR(handle64);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, kRegionBytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "peer_alloc: hipExtMallocWithFlags: %s", hipGetErrorString(e));
  e = hipMemset(p, 0, kRegionBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(handle64), p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return ts::fail(static_cast<int>(e), "peer_alloc: %s", hipGetErrorString(e));
  }
  *region = p;
  return TS_OK;
}

extern "C" int ts_peer_open(const void* handle64, void** region) {
  TS_REQUIRE_PTR(handle64); TS_REQUIRE_PTR(region);
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  hipError_t e = hipIpcOpenMemHandle(region, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "peer_open: hipIpcOpenMemHandle: %s", hipGetErrorString(e));
  return TS_OK;
}

extern "C" int ts_peer_close(void* region) {
  if (!region) return TS_OK;
  hipError_t e = hipIpcCloseMemHandle(region);
  return e == hipSuccess ? TS_OK : ts::fail(static_cast<int>(e), "peer_close: %s", hipGetErrorString(e));
}

extern "C" int ts_peer_free(void* region) {
  if (!region) return TS_OK;
  hipError_t e = hipFree(region);
  return e == hipSuccess ? TS_OK : ts::fail(static_cast<int>(e), "peer_free: %s", hipGetErrorString(e));
}

// 0 = every wait so far was answered; 1 + r = rank r did not post within the bound (the results of that exchange are garbage).
// Synchronises the stream first.
extern "C" int ts_peer_status(const void* ctx, int* status, void* stream) {
  const PeerCtx* c = static_cast<const PeerCtx*>(ctx);
  TS_REQUIRE_PTR(c); TS_REQUIRE_PTR(status);
  hipError_t e = hipStreamSynchronize(ts::as_stream(stream));
  unsigned words[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(words, static_cast<char*>(c->region[c->rank]) + kFlagBytes, sizeof(words), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return ts::fail(static_cast<int>(e), "peer_status: %s", hipGetErrorString(e));
  *status = static_cast<int>(words[1]);
  return TS_OK;
}

// The err word into pinned host memory behind the work queued on `stream` (no synchronisation): the caller reads *host_status once
// an event recorded after this call has completed -- train.TrainStep does that one step late, every step.
extern "C" int ts_peer_status_async(const void* ctx, int* host_status, void* stream) {
  const PeerCtx* c = static_cast<const PeerCtx*>(ctx);
  TS_REQUIRE_PTR(c); TS_REQUIRE_PTR(host_status);
  hipError_t e = hipMemcpyAsync(host_status, static_cast<char*>(c->region[c->rank]) + kFlagBytes + sizeof(unsigned), sizeof(int),
                                hipMemcpyDeviceToHost, ts::as_stream(stream));
  return e == hipSuccess ? TS_OK : ts::fail(static_cast<int>(e), "peer_status_async: %s", hipGetErrorString(e));
}

// Bound of one wait, milliseconds (> 0).  Returns the previous bound.
extern "C" long long ts_peer_set_timeout_ms(long long ms) {
  const long long old = static_cast<long long>(timeout_ticks() / 100000ull);
  if (ms > 0) g_timeout_ticks = static_cast<unsigned long long>(ms) * 100000ull;
  return old;
}

// Back to the state after ts_peer_alloc: flags, sequence number and err word of THIS rank's mailbox cleared.  Collective by
// contract: every rank calls it with nothing in flight, between two barriers of the caller's (PeerGroup.reset does that).
extern "C" int ts_peer_reset(const void* ctx, void* stream) {
  const PeerCtx* c = static_cast<const PeerCtx*>(ctx);
  TS_REQUIRE_PTR(c);
  hipError_t e = hipStreamSynchronize(ts::as_stream(stream));
  if (e == hipSuccess) e = hipMemset(c->region[c->rank], 0, kFlagBytes + kCtlBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e == hipSuccess ? TS_OK : ts::fail(static_cast<int>(e), "peer_reset: %s", hipGetErrorString(e));
}

// dst [world][n] <- every rank's src [n], rank order
extern "C" int ts_peer_all_gather(const void* ctx, const float* src, float* dst, int n, void* stream) {
  const PeerCtx* c = static_cast<const PeerCtx*>(ctx);
  if (int rc = check_ctx(c, n, "peer_all_gather")) return rc;
  TS_REQUIRE_PTR(src); TS_REQUIRE_PTR(dst);
  hipLaunchKernelGGL(peer_exchange_kernel<0>, dim3(1), dim3(256), 0, ts::as_stream(stream), *c, src, dst, n, nullptr, timeout_ticks());
  return ts::launched("peer_exchange_kernel");
}

// buf [n] <- (sum over ranks, rank order: bit-identical on every rank) * (*scale if scale is not NULL; a device scalar)
extern "C" int ts_peer_all_reduce_sum(const void* ctx, float* buf, int n, const float* scale, void* stream) {
  const PeerCtx* c = static_cast<const PeerCtx*>(ctx);
  if (int rc = check_ctx(c, n, "peer_all_reduce_sum")) return rc;
  TS_REQUIRE_PTR(buf);
  hipLaunchKernelGGL(peer_exchange_kernel<1>, dim3(1), dim3(256), 0, ts::as_stream(stream), *c, buf, buf, n, scale, timeout_ticks());
  return ts::launched("peer_exchange_kernel");
}
