// Native correlation volumes: what the reference obtains from the third-party
// `spatial_correlation_sampler.SpatialCorrelationSampler` (unvendored, unpinned: architecture/modeling/aggregation/utils/
// correlation.py:4-7 imports it inside try/except) in
//   correlation      correlation.py:10-29   out[b, ph*pW+pw, y, x] = lrelu_0.1( sum_c L[b,c,y,x] * R[b,c,y+ph-pH/2, x+pw-pW/2] )
//   correlation1d    correlation.py:32-57   the same with patch (1, 2*max_disp-1), first max_disp planes kept:
//                                           plane k = shift x + k - (max_disp-1), i.e. disparity max_disp-1-k
// for the arguments the reference uses (kernel_size 1, stride 1, padding 0, dilation 1, dilation_patch 1: the sampler's
// published definition -- a plain sum over channels, zeros outside the image).  The literal "shift-and-correlate over D
// candidate disparities" of the north star; no shipped configuration calls it.
//
// One lane = 4 consecutive pixels of one output plane: the left values are one 16-byte load per channel (the same for every
// plane of the pixel: served by L1 / L2 after the first plane), the shifted right values four clamped scalar loads selected
// afterwards (no load sits behind a branch).  Backward: two gather kernels (deterministic, no atomics).
#include <cstdlib>

#include "ts_common.hpp"

namespace {

struct Corr {
  int B, C, H, W, pH, pW, keep;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.1f * v; }

__global__ void __launch_bounds__(256)
correlation_fwd_kernel(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ out, const Corr p) {
  const int Wq = (p.W + 3) / 4;
  const long long n = static_cast<long long>(p.B) * p.keep * p.H * Wq;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xq = static_cast<int>(i % Wq);
  long long t = i / Wq;
  const int y = static_cast<int>(t % p.H); t /= p.H;
  const int k = static_cast<int>(t % p.keep), b = static_cast<int>(t / p.keep);
  const int dy = k / p.pW - p.pH / 2, dx = k % p.pW - p.pW / 2;
  const int x0 = xq * 4, ys = y + dy;
  const bool rowok = ys >= 0 && ys < p.H;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* Lp = L + static_cast<size_t>(b) * p.C * HW + static_cast<size_t>(y) * p.W;
  const float* Rp = R + static_cast<size_t>(b) * p.C * HW + static_cast<size_t>(rowok ? ys : 0) * p.W;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int xs[4];
  bool ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int xr = x0 + q + dx;
    ok[q] = rowok && xr >= 0 && xr < p.W && x0 + q < p.W;
    xs[q] = min(max(xr, 0), p.W - 1);
  }
  for (int c = 0; c < p.C; ++c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float l = Lp[c * HW + min(x0 + q, p.W - 1)];
      const float r = Rp[c * HW + xs[q]];
      acc[q] += ok[q] ? l * r : 0.f;
    }
  }
  float* op = out + ((static_cast<size_t>(b) * p.keep + k) * p.H + y) * p.W;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (x0 + q < p.W) op[x0 + q] = lrelu(acc[q]);
}

// grad wrt the left map: gL[c,y,x] = sum_k g'[k,y,x] R[c, y+dy_k, x+dx_k],   g' = g * (out > 0 ? 1 : 0.1)
// grad wrt the right map: gR[c,y,x] = sum_k g'[k, y-dy_k, x-dx_k] L[c, y-dy_k, x-dx_k]
template <bool RIGHT>
__global__ void __launch_bounds__(256)
correlation_bwd_kernel(const float* __restrict__ other, const float* __restrict__ out, const float* __restrict__ g,
                       float* __restrict__ grad, const Corr p) {
  const long long n = static_cast<long long>(p.B) * p.C * p.H * p.W;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = static_cast<int>(i % p.W);
  long long t = i / p.W;
  const int y = static_cast<int>(t % p.H); t /= p.H;
  const int c = static_cast<int>(t % p.C), b = static_cast<int>(t / p.C);
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* op = other + (static_cast<size_t>(b) * p.C + c) * HW;
  float acc = 0.f;
  for (int k = 0; k < p.keep; ++k) {
    const int dy = k / p.pW - p.pH / 2, dx = k % p.pW - p.pW / 2;
    // LEFT: the output pixel is (y, x), the partner R(y+dy, x+dx); RIGHT: the output pixel is (y-dy, x-dx), the partner L there
    const int oy = RIGHT ? y - dy : y, ox = RIGHT ? x - dx : x;
    const int py = RIGHT ? oy : y + dy, px = RIGHT ? ox : x + dx;
    const bool ok = oy >= 0 && oy < p.H && ox >= 0 && ox < p.W && py >= 0 && py < p.H && px >= 0 && px < p.W;
    const size_t oi = ((static_cast<size_t>(b) * p.keep + k) * p.H + (ok ? oy : 0)) * p.W + (ok ? ox : 0);
    const float gv = g[oi] * (out[oi] > 0.f ? 1.f : 0.1f);
    const float ov = op[static_cast<size_t>(ok ? py : 0) * p.W + (ok ? px : 0)];
    acc += ok ? gv * ov : 0.f;
  }
  grad[i] = acc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row correlation on the matrix cores (round 3; the kernel above stays for 2-D patches and as the small-shape fallback).
//
// For a patch of height 1 -- `correlation1d`, the literal shift-and-correlate over D disparities -- the volume of one image row is a
// BAND of the Gram matrix  P[x][x'] = sum_c L[c][x] * R[c][x']:  out[k][x] = P[x][x + dxmin + k].  P is a dense K = C contraction,
// so it goes to v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation): a workgroup owns 64 pixels of one row, wave w
// the 16-pixel block w; per 32-channel chunk the left strip [32][64] and the right span [32][64 + keep - 1 (+pad)] are staged in
// LDS once (zeros outside the image) and each wave multiplies its A fragments against T = ceil((15 + keep) / 16) right tiles --
// 16 / (16 + keep) of the products fall outside the band (8 % at D = 192).  The accumulator tiles are then written through LDS
// into [k][x] order (a lane of an MFMA tile holds one x' and four x, i.e. four DIFFERENT planes: stored directly, every lane of
// a store would touch its own cache line) and leave as 256-byte plane rows.
// Against the lane-per-output form above (C x D scalar loads per output quad, 1.2-1.4 TFLOP/s): 17-27x, see
// profiles/r03_stress_bench.txt.  Phase ablation at [4,32,192,272,480] (249 us before the buffer-load staging, 221 us with it): staging ~95 us, matrix work ~53 us (= its share of
// the f32 MFMA peak), scatter ~23 us, plane-row stores ~80-100 us (401 MB: the HBM floor) -- the phases of a workgroup run one after
// the other and three co-resident workgroups overlap them only partly.  Tried without gain: 16-byte staging loads from a
// 4-aligned span start, dealing the strips of a row to one XCD (its L2 then serves the shared right row), 16-channel chunks with
// the output tile in 64-plane passes (22 KB of LDS, 5+ workgroups per CU: SLOWER, 320 us -- three passes over the accumulators).
constexpr int CX = 64;        // pixels per workgroup
constexpr int CKC = 32;       // channels per staged chunk
constexpr int CPL = 80;       // LDS pitch of the left strip  (== 16 mod 32: the four channel rows of an A fragment hit disjoint banks)
constexpr int CPO = 69;       // LDS pitch of the [k][x] output tile (== 5 mod 32: a tile store is 2-way conflicted, the minimum)

template <int TMAX>
__global__ void __launch_bounds__(256, TMAX > 8 ? 2 : 3)
corr_row_mfma_kernel(const float* __restrict__ L, const float* __restrict__ R, float* __restrict__ out, const Corr p, int T, int pitchR,
                     int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sL = lds;                         // [CKC][CPL]
  float* sR = lds + CKC * CPL;             // [CKC][pitchR]
  float* sO = lds;                         // [keep][CPO]: the output tile lives over the input tiles
  constexpr int NCOL = TMAX > 8 ? 2 : 1;   // right-span columns per thread (span <= 48 + 16 * TMAX)
  const int xs0 = blockIdx.x * CX, y_first = blockIdx.y * rows_per_wg, b = blockIdx.z;
  const int y_end = min(y_first + rows_per_wg, p.H);
  const int dxmin = -(p.pW / 2);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, kq = lane >> 4;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* Lb = L + static_cast<size_t>(b) * p.C * HW;
  const float* Rb = R + static_cast<size_t>(b) * p.C * HW;
  const int span = 48 + 16 * T;            // right pixels the four waves touch, starting at xs0 + dxmin
  const int nchunk = (p.C + CKC - 1) / CKC;
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};

  // this thread's share of a (row, 32-channel chunk) step: left column tid & 63 of channels (tid >> 6) + 4 m, right columns tid (+ 256)
  // of every channel.  Buffer loads: the batch item's map behind one descriptor, the lane's part of the address ONE 32-bit offset
  // (column, for the left strip also its channel phase), the step's part (row, chunk, channel) the scalar offset.  A column outside
  // the image gets an out-of-range offset and a channel past the end falls off the descriptor: both read 0 -- nothing is masked at
  // commit.  (Per-lane 64-bit pointers, hipcc's choice for plain loads here, cost 100 registers and spilled.)
  constexpr unsigned OOR = 0x80000000u;                        // the host checks C * H * W * 4 < 2 GiB
  const unsigned map_bytes = static_cast<unsigned>(p.C) * static_cast<unsigned>(HW) * 4u;
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Lb), 0, map_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rb), 0, map_bytes, 0x00020000);
  const int lxl = tid & 63, lcb = tid >> 6;
  const unsigned lvo = (xs0 + lxl < p.W) ? (static_cast<unsigned>(lcb) * static_cast<unsigned>(HW) + static_cast<unsigned>(xs0 + lxl)) * 4u : OOR;
  unsigned rvo[NCOL];
#pragma unroll
  for (int n = 0; n < NCOL; ++n) {
    const int x = xs0 + dxmin + tid + 256 * n;
    rvo[n] = (tid + 256 * n < span && x >= 0 && x < p.W) ? static_cast<unsigned>(x) * 4u : OOR;
  }
  float vl[CKC / 4], vr[NCOL][CKC];
  // The next step's loads are in flight while this step is multiplied, scattered and stored (a workgroup used to run its phases one
  // after the other: staging was the longest of them).
  const unsigned HWb = static_cast<unsigned>(HW) * 4u;
  auto prefetch = [&](int y, int c0) {
    const unsigned so = (static_cast<unsigned>(c0) * static_cast<unsigned>(HW) + static_cast<unsigned>(y) * static_cast<unsigned>(p.W)) * 4u;
#pragma unroll
    for (int m = 0; m < CKC / 4; ++m)
      vl[m] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(lrs, lvo, so + static_cast<unsigned>(4 * m) * HWb, 0));
#pragma unroll
    for (int c = 0; c < CKC; ++c)
#pragma unroll
      for (int n = 0; n < NCOL; ++n)
        vr[n][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, rvo[n], so + static_cast<unsigned>(c) * HWb, 0));
  };
  if (y_first < y_end) prefetch(y_first, 0);
#pragma unroll 1
  for (int y = y_first; y < y_end; ++y) {
#pragma unroll 1
    for (int ck = 0; ck < nchunk; ++ck) {
      const int c0 = ck * CKC;
      __syncthreads();                     // the previous step's fragments / output rows have been read
#pragma unroll
      for (int m = 0; m < CKC / 4; ++m) sL[(lcb + 4 * m) * CPL + lxl] = vl[m];
#pragma unroll
      for (int n = 0; n < NCOL; ++n) {
        if (tid + 256 * n < span) {
#pragma unroll
          for (int c = 0; c < CKC; ++c) sR[c * pitchR + tid + 256 * n] = vr[n][c];
        }
      }
      __syncthreads();
      if (ck + 1 < nchunk) prefetch(y, c0 + CKC);
      else if (y + 1 < y_end) prefetch(y + 1, 0);
      float a[CKC / 4];
#pragma unroll
      for (int q = 0; q < CKC / 4; ++q) a[q] = sL[(4 * q + kq) * CPL + 16 * wave + j];
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
          const float* rb = sR + kq * pitchR + 16 * wave + 16 * t + j;
#pragma unroll
          for (int q = 0; q < CKC / 4; ++q) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], rb[4 * q * pitchR], acc[t], 0, 0, 0);
        }
      }
    }
    // ---- band of the Gram tiles -> [k][x] in LDS (over the input tiles), then plane rows
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * kq + r;            // pixel of the wave's block; j = right pixel of tile t
          const int k = 16 * t + j - i;        // (x' - x) - dxmin
          if (k >= 0 && k < p.keep) sO[k * CPO + 16 * wave + i] = acc[t][r];
        }
        acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    float* op = out + ((static_cast<size_t>(b) * p.keep) * p.H + y) * p.W + xs0;
    const bool vec = (p.W & 3) == 0;
    for (int i = tid; i < p.keep * (CX / 4); i += 256) {
      const int k = i >> 4, xq = (i & 15) * 4;
      if (xs0 + xq >= p.W) continue;
      const float* sp = sO + k * CPO + xq;
      float4 v = make_float4(lrelu(sp[0]), lrelu(sp[1]), lrelu(sp[2]), lrelu(sp[3]));
      float* dst = op + static_cast<size_t>(k) * HW + xq;
      if (vec) {
        *reinterpret_cast<float4*>(dst) = v;
      } else {
        dst[0] = v.x;
        if (xs0 + xq + 1 < p.W) dst[1] = v.y;
        if (xs0 + xq + 2 < p.W) dst[2] = v.z;
        if (xs0 + xq + 3 < p.W) dst[3] = v.w;
      }
    }
  }
}

// 0: the shape goes to the lane-per-output kernel
int corr_row_tiles(const Corr& p) {
  if (p.pH != 1 || p.keep > 241 || p.C < 4 || p.W < 16) return 0;
  if (static_cast<unsigned long long>(p.C) * p.H * p.W * 4ull >= 0x7ff00000ull) return 0;      // one buffer descriptor per batch item
  return (15 + p.keep + 15) / 16;
}

int check(const Corr& p) {
  TS_REQUIRE(p.B > 0 && p.C > 0 && p.H > 0 && p.W > 0, TS_ERR_SHAPE, "correlation: non-positive size");
  TS_REQUIRE(p.pH >= 1 && p.pW >= 1 && (p.pH & 1) && (p.pW & 1), TS_ERR_SHAPE, "correlation: patch sizes must be odd and >= 1");
  TS_REQUIRE(p.keep >= 1 && p.keep <= p.pH * p.pW, TS_ERR_SHAPE, "correlation: keep outside 1..pH*pW");
  return TS_OK;
}

}  // namespace

extern "C" int ts_correlation_fwd(const float* left, const float* right, float* out, int B, int C, int H, int W, int patch_h,
                                  int patch_w, int keep, void* stream) {
  const Corr p{B, C, H, W, patch_h, patch_w, keep};
  if (int rc = check(p)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(out);
  if (const int T = corr_row_tiles(p)) {            // patch height 1: a band of the row's Gram matrix, on the matrix cores
    TS_REQUIRE(H <= 65535 && B <= 65535, TS_ERR_UNSUPPORTED, "correlation: grid too large");
    const int span = 48 + 16 * T;
    int pitchR = span;
    while ((pitchR & 31) != 16) ++pitchR;
    const size_t in_b = static_cast<size_t>(CKC) * (CPL + pitchR) * 4, out_b = static_cast<size_t>(keep) * CPO * 4;
    const size_t shm = in_b > out_b ? in_b : out_b;
    // keep >= 238 needs more than 64 KB of dynamic LDS: allowed up to the device's per-workgroup limit (160 KB on gfx950) once the
    // kernel's attribute is raised; a device that cannot hold the tile takes the lane-per-output kernel below instead of failing
    static const size_t lds_limit = [] {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return size_t(64 * 1024);
      return static_cast<size_t>(v);
    }();
    if (shm <= lds_limit) {
    // rows per workgroup (the next row's loads fly under this row's matrix work and stores): 2 where that still leaves four rounds of
    // workgroups, else 1 -- measured at [4,32,272,480] D=192: 1 row 222 us, 2 rows 221, 4 rows 233, 8 rows 264 (longer chains of
    // fewer, fatter workgroups lose more to the tail than the prefetch wins)
    const long long strips_all = static_cast<long long>((W + CX - 1) / CX) * H * B;
    int rpw = strips_all >= 8ll * 3 * ts::kNumCU ? 2 : 1;
    static const int rpw_env = getenv("TS_CORR_ROWS") ? atoi(getenv("TS_CORR_ROWS")) : 0;
    if (rpw_env > 0) rpw = rpw_env;
    const dim3 grid((W + CX - 1) / CX, (H + rpw - 1) / rpw, B);
    hipStream_t st = ts::as_stream(stream);
    if (T <= 4) hipLaunchKernelGGL(corr_row_mfma_kernel<4>, grid, dim3(256), shm, st, left, right, out, p, T, pitchR, rpw);
    else if (T <= 8) hipLaunchKernelGGL(corr_row_mfma_kernel<8>, grid, dim3(256), shm, st, left, right, out, p, T, pitchR, rpw);
    else {
      if (shm > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&corr_row_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(shm));
      hipLaunchKernelGGL(corr_row_mfma_kernel<16>, grid, dim3(256), shm, st, left, right, out, p, T, pitchR, rpw);
    }
    return ts::launched("corr_row_mfma_kernel");
    }
  }
  const long long n = static_cast<long long>(B) * keep * H * ((W + 3) / 4);
  hipLaunchKernelGGL(correlation_fwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, ts::as_stream(stream),
                     left, right, out, p);
  return ts::launched("correlation_fwd_kernel");
}

extern "C" int ts_correlation_bwd(const float* left, const float* right, const float* out, const float* grad_out,
                                  float* grad_left, float* grad_right, int B, int C, int H, int W, int patch_h, int patch_w,
                                  int keep, void* stream) {
  const Corr p{B, C, H, W, patch_h, patch_w, keep};
  if (int rc = check(p)) return rc;
  TS_REQUIRE_PTR(left); TS_REQUIRE_PTR(right); TS_REQUIRE_PTR(out); TS_REQUIRE_PTR(grad_out);
  const long long n = static_cast<long long>(B) * C * H * W;
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (grad_left) {
    hipLaunchKernelGGL(correlation_bwd_kernel<false>, grid, dim3(256), 0, ts::as_stream(stream), right, out, grad_out, grad_left, p);
    if (int rc = ts::launched("correlation_bwd_kernel<left>")) return rc;
  }
  if (grad_right) {
    hipLaunchKernelGGL(correlation_bwd_kernel<true>, grid, dim3(256), 0, ts::as_stream(stream), left, out, grad_out, grad_right, p);
    if (int rc = ts::launched("correlation_bwd_kernel<right>")) return rc;
  }
  return TS_OK;
}
