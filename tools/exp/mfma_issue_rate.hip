// Experiment (round 6): cycles per v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD, as the ping-pong kernel's compute half issues them
// (csrc/conv_x6p.hip measured 40 per instruction where the tables say 32).  Variants: accumulator chains 1 / 2 / 4, the kernel's own
// operand pattern (two chains sharing the A fragment, six products per tap), with and without a ds_read_b128 behind every MFMA, and
// the 16x16x32 form.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_issue_rate tools/exp/mfma_issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

template <int CHAINS, int READS>     // READS: ds_read_b128 per MFMA (0 / 1)
__global__ void __launch_bounds__(512) k32(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps, int active) {
  __shared__ u32x4 lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048; i += blockDim.x) lds[i] = src[i & 255];
  __syncthreads();
  if (static_cast<int>(threadIdx.x >> 6) >= active) return;
  bf16x8 a[3], b[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(lane + i * 64) & 255]);
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(lane + 17 * i + 3) & 255]);
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  u32x4 sink = {0u, 0u, 0u, 0u};
  const unsigned long long t0 = now();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 108; ++m) {
      const int c = m % CHAINS;
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(m / CHAINS) % 3], b[(m / 2) % 6], acc[c], 0, 0, 0);
      if constexpr (READS > 0) {
        const u32x4 v = lds[(lane + m * 64) & 2047];
        sink.x ^= v.x; sink.y ^= v.y; sink.z ^= v.z; sink.w ^= v.w;
      }
    }
  }
  const unsigned long long t1 = now();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(sink.x ^ sink.y ^ sink.z ^ sink.w);
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS>
__global__ void __launch_bounds__(512) k16(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps, int active) {
  if (static_cast<int>(threadIdx.x >> 6) >= active) return;
  const int lane = threadIdx.x;
  bf16x8 a[3], b[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(lane + i * 64) & 255]);
#pragma unroll
  for (int i = 0; i < 6; ++i) b[i] = __builtin_bit_cast(bf16x8, src[(lane + 17 * i + 3) & 255]);
  f32x4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = now();
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 216; ++m) {
      const int c = m % CHAINS;
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(m / CHAINS) % 3], b[(m / 2) % 6], acc[c], 0, 0, 0);
    }
  }
  const unsigned long long t1 = now();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}


// The ping-pong kernel's own step: per tap 3 weight fragments + 3 x 2 pixel fragments read from LDS (one read pinned behind each of
// the first MFMAs of the previous tap), 12 MFMAs on two per-chunk accumulators, the chunk's sums added to the running sums at its end.
// MODE 0: as the kernel; 1: no chunk-end adds (MFMAs accumulate on); 2: reads issued but operands static (no waits on them);
// 3: as 0 with the reads two taps ahead (three fragment buffers).
template <int MODE>
__global__ void __launch_bounds__(512) k32f(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int reps, int active) {
  extern __shared__ u32x4 dl[];       // 27 * 64 weight fragments + 3 * 2 * 1024 pixel slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 27 * 64 + 6144; i += blockDim.x) dl[i] = src[i & 255];
  __syncthreads();
  if (wave >= active) return;
  const u32x4* wb = dl;
  const u32x4* in6 = dl + 27 * 64;
  const int aoff = lane;
  int rbase[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) rbase[i][j] = ((wave & 3) * 160 + i * 40 + j * 4 + (lane & 31) + (lane >> 5) * 512) & 1023;
  constexpr int NPB = 2;
  constexpr int NB = MODE == 3 ? 3 : 2;
  f32x16 acc[NPB];
#pragma unroll
  for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[pb][r] = 0.f;
  bf16x8 sa[3], sb[3][NPB];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sa[i] = __builtin_bit_cast(bf16x8, src[(lane + i * 64) & 255]);
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) sb[i][pb] = __builtin_bit_cast(bf16x8, src[(lane + 17 * i + 3 + pb) & 255]);
  }
  u32x4 sink = {0u, 0u, 0u, 0u};
  const unsigned long long t0 = now();
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
    f32x16 part[NPB];
#pragma unroll
    for (int a = 0; a < NPB; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[a][r] = MODE == 1 ? acc[a][r] : 0.f;
    bf16x8 af[NB][3], bf[NB][3][NPB];
    auto load_frag = [&](int tap, int buf) {
      constexpr int OA[3] = {0, 2, 1}, OB[3] = {2, 0, 1};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        af[buf][OA[k]] = __builtin_bit_cast(bf16x8, wb[(OA[k] * 9 + tap) * 64 + aoff]);
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) bf[buf][OB[k]][pb] = __builtin_bit_cast(bf16x8, in6[OB[k] * 2 * 1024 + rbase[(pb + tap / 3) & 1][tap % 3] + ((pb + tap / 3) >> 1) * 80]);
      }
    };
    constexpr int AHEAD = MODE == 3 ? 2 : 1;
    load_frag(0, 0);
    if (AHEAD == 2) load_frag(1, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, (3 + 3 * NPB) * AHEAD, 0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + AHEAD < 9) load_frag(tap + AHEAD, (tap + AHEAD) % NB);
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb) {
          if constexpr (MODE == 2) {
            part[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[PA[t]], sb[PB[t]][pb], part[pb], 0, 0, 0);
          } else {
            part[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tap % NB][PA[t]], bf[tap % NB][PB[t]][pb], part[pb], 0, 0, 0);
          }
        }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const u32x4 v = __builtin_bit_cast(u32x4, af[tap % NB][k]);
          sink.x ^= v.x;
#pragma unroll
          for (int pb = 0; pb < NPB; ++pb) { const u32x4 u = __builtin_bit_cast(u32x4, bf[tap % NB][k][pb]); sink.y ^= u.y; }
        }
      }
      if (tap + AHEAD < 9) {
#pragma unroll
        for (int k = 0; k < 3 + 3 * NPB; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NPB - (3 + 3 * NPB), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NPB, 0);
      }
    }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (MODE == 1) acc[pb][r] = part[pb][r];
        else acc[pb][r] += part[pb][r];
      }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const unsigned long long t1 = now();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NPB; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + __uint_as_float(sink.x ^ sink.y);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kernel, int threads, int active, int per_rep, const u32x4* src, float* out, unsigned long long* cyc, unsigned lds = 0) {
  const int blocks = 256, reps = 40;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, 0, src, out, cyc, reps, active);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double sum = 0; unsigned long long mx = 0, mn = ~0ull;
  for (auto v : h) { sum += v; if (v > mx) mx = v; if (v < mn) mn = v; }
  const double n = static_cast<double>(reps) * per_rep;
  printf("%-52s threads %3d active waves %d: %6.2f cycles / MFMA (min %6.2f max %6.2f)  launch %.1f us\n", name, threads, active,
         sum / blocks / n, mn / n, mx / n, ms * 1e3);
}

int main() {
  u32x4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 256 * sizeof(u32x4)); hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x3f803f80u ^ (static_cast<unsigned>(i) * 2654435761u & 0x007f007fu);     // bf16 pairs near 1.0
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  for (int active : {4, 8}) {
    const int threads = 512;
    run("32x32x16 1 chain", k32<1, 0>, threads, active, 108, src, out, cyc);
    run("32x32x16 2 chains", k32<2, 0>, threads, active, 108, src, out, cyc);
    run("32x32x16 3 chains", k32<3, 0>, threads, active, 108, src, out, cyc);
    run("32x32x16 4 chains", k32<4, 0>, threads, active, 108, src, out, cyc);
    run("32x32x16 2 chains + ds_read_b128 per MFMA", k32<2, 1>, threads, active, 108, src, out, cyc);
    run("32x32x16 4 chains + ds_read_b128 per MFMA", k32<4, 1>, threads, active, 108, src, out, cyc);
    run("16x16x32 1 chain", k16<1>, threads, active, 216, src, out, cyc);
    run("16x16x32 2 chains", k16<2>, threads, active, 216, src, out, cyc);
    run("16x16x32 4 chains", k16<4>, threads, active, 216, src, out, cyc);
  }
  const unsigned fl = (27 * 64 + 6144) * 16;
  for (int m = 0; m < 4; ++m) {
    auto k = m == 0 ? k32f<0> : m == 1 ? k32f<1> : m == 2 ? k32f<2> : k32f<3>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, fl);
  }
  run("kernel step (LDS-fed, pinned reads, chunk-end adds)", k32f<0>, 256, 4, 108, src, out, cyc, fl);
  run("... without the chunk-end adds", k32f<1>, 256, 4, 108, src, out, cyc, fl);
  run("... reads issued, operands static", k32f<2>, 256, 4, 108, src, out, cyc, fl);
  run("... reads two taps ahead", k32f<3>, 256, 4, 108, src, out, cyc, fl);
  run("32x32x16 2 chains, 256-thread workgroup", k32<2, 0>, 256, 4, 108, src, out, cyc);
  return 0;
}
