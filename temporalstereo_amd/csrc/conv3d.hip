// K3 -- separable 3-D convolution building blocks of the aggregation pyramid for gfx950 (MI355X),
// inference form: BatchNorm folded to a per-channel (scale, shift), activation fused.
//
// Replaces, layer by layer, what the reference reaches through torch/cuDNN for
//   Conv3d / ConvTranspose3d wrappers   architecture/modeling/layers/basic_layers.py:194-235,340-388
//   DepthwiseConv3D (separable pair)    architecture/modeling/aggregation/TemporalStereo/module.py:111-147
//   DepthwiseConvTranspose3D            module.py:149-184
// Every 3-D convolution of the model has a kernel that is 1 along D or 1 along H,W, so two kernel
// families cover them all:
//   conv_hw : (1,3,3) taps, stride/dilation in H,W; the (B,D) planes are independent images.
//   conv_d  : (k,1,1) taps (k = 1,3,5) along D; every pixel column is independent.
// plus their stride-2 transposed forms.
//
// Design (fp32 everywhere -- the parity bar |dEPE| < 1e-3 px is an fp32 bar; f32 MFMA runs at the
// VALU rate on gfx950, so the matrix cores buy no FLOPs here): direct convolution on the vector
// ALU with the weights of the current (channel, tap) held in SGPRs -- every lane of a wavefront
// needs the same Cout weights, so they come through the scalar cache with s_load_dwordxN and feed
// v_fmac as the scalar operand; a lane owns one output pixel and all Cout accumulators, so each
// input value fetched from LDS is used Cout times.  Input tiles (+halo) are staged through LDS a
// chunk of channels at a time with coalesced row loads; stores are row-contiguous.
// Weights are pre-laid out [Cin][taps][Cout] (Cout contiguous) by the host so one scalar load
// brings all output channels of a tap.
#include "ts_common.hpp"

namespace {

enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_TANH_OFFSET = 3 };

__device__ __forceinline__ float apply_act(float v, int act, float p) {
  switch (act) {
    case ACT_SILU: return v / (1.f + expf(-v));
    case ACT_RELU: return fmaxf(v, 0.f);
    // PredictionHeads.regress_offset (module.py:384-390): tanh(x/100).clamp(-1,1) * delta
    case ACT_TANH_OFFSET: return fminf(fmaxf(tanhf(v / 100.f), -1.f), 1.f) * p;
    default: return v;
  }
}

struct ConvHW {
  int B, Cin, Cout, D, H, W, Ho, Wo;
  int stride, dil, pad;        // in H and W
  int act;
  float act_param;
  int in_cstride_planes;       // input channel stride in (D*H*W) planes units == D (dense) ...
  long long in_bstride, out_bstride;   // elements between batch items (allows channel-sliced views)
  long long in_cstride, out_cstride;   // elements between channels
};

constexpr int TILE_Y = 8, TILE_X = 32;     // output pixels per workgroup (256 lanes, x fastest)
constexpr int CI_CHUNK = 8;

// y[b,co,d,oy,ox] = act( scale[co] * sum_{ci,ky,kx} w[ci][ky][kx][co] * x[b,ci,d,oy*s+ky*dl-p,ox*s+kx*dl-p] + shift[co] )
template <int COUT>
__global__ void __launch_bounds__(256)
conv_hw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
               const float* __restrict__ shift, float* __restrict__ y, const ConvHW p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tiles_x = (p.Wo + TILE_X - 1) / TILE_X;
  const int tile = blockIdx.x;
  const int ty0 = (tile / tiles_x) * TILE_Y, tx0 = (tile % tiles_x) * TILE_X;
  const int d = blockIdx.y, b = blockIdx.z;
  const int tx = threadIdx.x & (TILE_X - 1), ty = threadIdx.x / TILE_X;
  const int oy = ty0 + ty, ox = tx0 + tx;
  const int in_rows = (TILE_Y - 1) * p.stride + 2 * p.dil + 1;
  const int in_cols = (TILE_X - 1) * p.stride + 2 * p.dil + 1;
  const int in_cols_p = in_cols | 1;                        // odd row pitch: spreads banks for stride 2
  const int iy0 = ty0 * p.stride - p.pad, ix0 = tx0 * p.stride - p.pad;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* xb = x + static_cast<size_t>(b) * p.in_bstride + static_cast<size_t>(d) * HW;

  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;

  const int chan_elems = in_rows * in_cols_p;
  for (int c0 = 0; c0 < p.Cin; c0 += CI_CHUNK) {
    const int nc = min(CI_CHUNK, p.Cin - c0);
    __syncthreads();
    // stage nc channels x in_rows x in_cols (zero outside the image)
    for (int i = threadIdx.x; i < nc * in_rows * in_cols; i += blockDim.x) {
      const int cx = i % in_cols;
      const int r = i / in_cols;
      const int cy = r % in_rows, c = r / in_rows;
      const int gy = iy0 + cy, gx = ix0 + cx;
      float v = 0.f;
      if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        v = xb[static_cast<size_t>(c0 + c) * p.in_cstride + static_cast<size_t>(gy) * p.W + gx];
      lds[c * chan_elems + cy * in_cols_p + cx] = v;
    }
    __syncthreads();
    const float* lt = lds + (ty * p.stride) * in_cols_p + tx * p.stride;
    for (int c = 0; c < nc; ++c) {
      const float* wc = w + static_cast<size_t>(c0 + c) * 9 * COUT;       // wave-uniform -> scalar loads
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float xv = lt[c * chan_elems + ky * p.dil * in_cols_p + kx * p.dil];
          const float* wt = wc + (ky * 3 + kx) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wt[co], xv, acc[co]);
        }
    }
  }
  if (oy < p.Ho && ox < p.Wo) {
    float* yb = y + static_cast<size_t>(b) * p.out_bstride + (static_cast<size_t>(d) * p.Ho + oy) * p.Wo + ox;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      if (co < p.Cout) yb[static_cast<size_t>(co) * p.out_cstride] = apply_act(acc[co] * scale[co] + shift[co], p.act, p.act_param);
    }
  }
}

struct ConvD {
  int B, Cin, Cout, Din, Dout, HW;
  int k, stride, dil, pad;     // along D
  int act;
  float act_param;
  long long in_bstride, out_bstride, in_cstride, out_cstride;
};

// y[b,co,od,p] = act( scale[co] * sum_{ci,t} w[ci][t][co] * x[b,ci,od*s+t*dl-pad,p] + shift[co] )
template <int COUT>
__global__ void __launch_bounds__(256)
conv_d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
              const float* __restrict__ shift, float* __restrict__ y, const ConvD p) {
  const int od = blockIdx.y, b = blockIdx.z;
  const float* xb = x + static_cast<size_t>(b) * p.in_bstride;
  for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < p.HW; px += gridDim.x * blockDim.x) {
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int t = 0; t < p.k; ++t) {
      const int id = od * p.stride + t * p.dil - p.pad;       // uniform
      if (id < 0 || id >= p.Din) continue;
      const float* xp = xb + static_cast<size_t>(id) * p.HW + px;
      for (int ci = 0; ci < p.Cin; ++ci) {
        const float xv = xp[static_cast<size_t>(ci) * p.in_cstride];
        const float* wt = w + (static_cast<size_t>(ci) * p.k + t) * COUT;   // uniform -> scalar loads
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wt[co], xv, acc[co]);
      }
    }
    float* yb = y + static_cast<size_t>(b) * p.out_bstride + static_cast<size_t>(od) * p.HW + px;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      if (co < p.Cout) yb[static_cast<size_t>(co) * p.out_cstride] = apply_act(acc[co] * scale[co] + shift[co], p.act, p.act_param);
  }
}

// ---- transposed, stride 2, kernel 3, padding 1, output_padding 1 (module.py:248-258) ---------------
// H,W form: out is (2H, 2W).  out[oy] draws from ky with (oy + 1 - ky) even: even oy -> ky=1, iy=oy/2;
// odd oy -> ky=0 (iy=(oy+1)/2, valid if < H) and ky=2 (iy=(oy-1)/2).  Weight layout [Cin][ky][kx][Cout].
template <int COUT>
__global__ void __launch_bounds__(256)
deconv_hw_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                 const float* __restrict__ shift, float* __restrict__ y, const ConvHW p) {
  const int d = blockIdx.y, b = blockIdx.z;
  const size_t HW = static_cast<size_t>(p.H) * p.W;
  const float* xb = x + static_cast<size_t>(b) * p.in_bstride + static_cast<size_t>(d) * HW;
  const int n = p.Ho * p.Wo;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
    const int oy = o / p.Wo, ox = o - oy * p.Wo;
    // up to two source rows / columns with their tap index (-1: none)
    int iyA = -1, kyA = 0, iyB = -1, kyB = 0, ixA = -1, kxA = 0, ixB = -1, kxB = 0;
    if ((oy & 1) == 0) { iyA = oy >> 1; kyA = 1; }
    else { iyA = (oy - 1) >> 1; kyA = 2; iyB = (oy + 1) >> 1; kyB = 0; if (iyB >= p.H) iyB = -1; }
    if ((ox & 1) == 0) { ixA = ox >> 1; kxA = 1; }
    else { ixA = (ox - 1) >> 1; kxA = 2; ixB = (ox + 1) >> 1; kxB = 0; if (ixB >= p.W) ixB = -1; }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < p.Cin; ++ci) {
      const float* xc = xb + static_cast<size_t>(ci) * p.in_cstride;
      const float* wc = w + static_cast<size_t>(ci) * 9 * COUT;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int iy = a ? iyB : iyA, ky = a ? kyB : kyA;
        if (iy < 0) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int ix = c ? ixB : ixA, kx = c ? kxB : kxA;
          if (ix < 0) continue;
          const float xv = xc[static_cast<size_t>(iy) * p.W + ix];
          const float* wt = wc + (ky * 3 + kx) * COUT;      // per-lane tap: vector loads (tiny layers only)
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wt[co], xv, acc[co]);
        }
      }
    }
    float* yb = y + static_cast<size_t>(b) * p.out_bstride + static_cast<size_t>(d) * n + o;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      if (co < p.Cout) yb[static_cast<size_t>(co) * p.out_cstride] = apply_act(acc[co] * scale[co] + shift[co], p.act, p.act_param);
  }
}

// D form: Dout = 2 * Din.  even od -> t=1, id=od/2; odd od -> t=2 (id=(od-1)/2) and t=0 (id=(od+1)/2 if < Din).
template <int COUT>
__global__ void __launch_bounds__(256)
deconv_d_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                const float* __restrict__ shift, float* __restrict__ y, const ConvD p) {
  const int od = blockIdx.y, b = blockIdx.z;
  const float* xb = x + static_cast<size_t>(b) * p.in_bstride;
  int idA, tA, idB = -1, tB = 0;
  if ((od & 1) == 0) { idA = od >> 1; tA = 1; }
  else { idA = (od - 1) >> 1; tA = 2; idB = (od + 1) >> 1; tB = 0; if (idB >= p.Din) idB = -1; }
  for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < p.HW; px += gridDim.x * blockDim.x) {
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int id = a ? idB : idA, t = a ? tB : tA;       // uniform
      if (id < 0) continue;
      const float* xp = xb + static_cast<size_t>(id) * p.HW + px;
      for (int ci = 0; ci < p.Cin; ++ci) {
        const float xv = xp[static_cast<size_t>(ci) * p.in_cstride];
        const float* wt = w + (static_cast<size_t>(ci) * 3 + t) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wt[co], xv, acc[co]);
      }
    }
    float* yb = y + static_cast<size_t>(b) * p.out_bstride + static_cast<size_t>(od) * p.HW + px;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      if (co < p.Cout) yb[static_cast<size_t>(co) * p.out_cstride] = apply_act(acc[co] * scale[co] + shift[co], p.act, p.act_param);
  }
}

int cout_bucket(int cout) {
  for (int b : {1, 8, 16, 32, 64})
    if (cout <= b) return b;
  return -1;
}

}  // namespace

#define TS_DISPATCH_COUT(bucket, KERNEL, ...)                                      \
  switch (bucket) {                                                                \
    case 1: hipLaunchKernelGGL(KERNEL<1>, __VA_ARGS__); break;                     \
    case 8: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;                     \
    case 16: hipLaunchKernelGGL(KERNEL<16>, __VA_ARGS__); break;                   \
    case 32: hipLaunchKernelGGL(KERNEL<32>, __VA_ARGS__); break;                   \
    default: hipLaunchKernelGGL(KERNEL<64>, __VA_ARGS__); break;                   \
  }

// x [B,Cin,D,H,W] -> y [B,Cout,D,Ho,Wo]; w_t is [Cin][3][3][CoutPad] with CoutPad = ts_conv_cout_pad(Cout)
// (zero padded), scale/shift [CoutPad].  Channel/batch strides are in elements so that x / y may be
// channel slices of larger tensors (concatenation without a copy).  transposed != 0: stride-2
// ConvTranspose3d(1,3,3) with padding 1, output_padding 1 (Ho = 2H, Wo = 2W).
extern "C" int ts_conv3d_hw_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                                int B, int Cin, int Cout, int D, int H, int W, int stride, int dilation,
                                int transposed, int act, float act_param,
                                long long in_bstride, long long in_cstride, long long out_bstride,
                                long long out_cstride, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_hw: non-positive size");
  TS_REQUIRE(stride == 1 || stride == 2, TS_ERR_UNSUPPORTED, "conv3d_hw: stride must be 1 or 2");
  TS_REQUIRE(dilation == 1 || dilation == 2, TS_ERR_UNSUPPORTED, "conv3d_hw: dilation must be 1 or 2");
  TS_REQUIRE(!transposed || (stride == 2 && dilation == 1), TS_ERR_UNSUPPORTED, "conv3d_hw: transposed form is stride 2, dilation 1");
  TS_REQUIRE(act >= 0 && act <= 3, TS_ERR_SHAPE, "conv3d_hw: unknown activation");
  TS_REQUIRE(B <= 65535 && D <= 65535, TS_ERR_UNSUPPORTED, "conv3d_hw: grid too large");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(scale); TS_REQUIRE_PTR(shift); TS_REQUIRE_PTR(y);
  const int bucket = cout_bucket(Cout);
  TS_REQUIRE(bucket > 0, TS_ERR_UNSUPPORTED, "conv3d_hw: Cout=%d > 64", Cout);
  ConvHW p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.D = D; p.H = H; p.W = W;
  p.stride = stride; p.dil = dilation; p.pad = dilation; p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  hipStream_t st = ts::as_stream(stream);
  if (transposed) {
    p.Ho = 2 * H; p.Wo = 2 * W;
    int blocks = (p.Ho * p.Wo + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid(blocks, D, B);
    TS_DISPATCH_COUT(bucket, deconv_hw_kernel, grid, dim3(256), 0, st, x, w_t, scale, shift, y, p);
    return ts::launched("deconv_hw_kernel");
  }
  p.Ho = (H + 2 * p.pad - 2 * dilation - 1) / stride + 1;
  p.Wo = (W + 2 * p.pad - 2 * dilation - 1) / stride + 1;
  const int tiles = ((p.Ho + TILE_Y - 1) / TILE_Y) * ((p.Wo + TILE_X - 1) / TILE_X);
  const int in_rows = (TILE_Y - 1) * stride + 2 * dilation + 1;
  const int in_cols_p = ((TILE_X - 1) * stride + 2 * dilation + 1) | 1;
  const size_t lds_bytes = static_cast<size_t>(CI_CHUNK) * in_rows * in_cols_p * sizeof(float);
  const dim3 grid(tiles, D, B);
  TS_DISPATCH_COUT(bucket, conv_hw_kernel, grid, dim3(256), lds_bytes, st, x, w_t, scale, shift, y, p);
  return ts::launched("conv_hw_kernel");
}

// x [B,Cin,Din,H,W] -> y [B,Cout,Dout,H,W]; w_t is [Cin][k][CoutPad].  k in {1,3,5}.  transposed != 0:
// ConvTranspose3d(3,1,1) stride 2, padding 1, output_padding 1 (Dout = 2 Din).
extern "C" int ts_conv3d_d_fwd(const float* x, const float* w_t, const float* scale, const float* shift, float* y,
                               int B, int Cin, int Cout, int Din, int H, int W, int k, int stride, int dilation,
                               int padding, int transposed, int act, float act_param,
                               long long in_bstride, long long in_cstride, long long out_bstride,
                               long long out_cstride, void* stream) {
  TS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Din > 0 && H > 0 && W > 0, TS_ERR_SHAPE, "conv3d_d: non-positive size");
  TS_REQUIRE(k == 1 || k == 3 || k == 5, TS_ERR_UNSUPPORTED, "conv3d_d: k must be 1, 3 or 5");
  TS_REQUIRE(stride >= 1 && stride <= 2 && dilation >= 1 && padding >= 0, TS_ERR_UNSUPPORTED, "conv3d_d: bad stride/dilation/padding");
  TS_REQUIRE(!transposed || (k == 3 && stride == 2 && dilation == 1 && padding == 1), TS_ERR_UNSUPPORTED,
             "conv3d_d: transposed form is k=3, stride 2, padding 1");
  TS_REQUIRE(act >= 0 && act <= 3, TS_ERR_SHAPE, "conv3d_d: unknown activation");
  TS_REQUIRE_PTR(x); TS_REQUIRE_PTR(w_t); TS_REQUIRE_PTR(scale); TS_REQUIRE_PTR(shift); TS_REQUIRE_PTR(y);
  const int bucket = cout_bucket(Cout);
  TS_REQUIRE(bucket > 0, TS_ERR_UNSUPPORTED, "conv3d_d: Cout=%d > 64", Cout);
  ConvD p;
  p.B = B; p.Cin = Cin; p.Cout = Cout; p.Din = Din; p.HW = H * W;
  p.k = k; p.stride = stride; p.dil = dilation; p.pad = padding; p.act = act; p.act_param = act_param;
  p.in_bstride = in_bstride; p.in_cstride = in_cstride; p.out_bstride = out_bstride; p.out_cstride = out_cstride;
  p.Dout = transposed ? 2 * Din : (Din + 2 * padding - dilation * (k - 1) - 1) / stride + 1;
  TS_REQUIRE(p.Dout > 0 && p.Dout <= 65535 && B <= 65535, TS_ERR_SHAPE, "conv3d_d: bad output depth");
  int blocks = (p.HW + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid(blocks, p.Dout, B);
  hipStream_t st = ts::as_stream(stream);
  if (transposed) {
    TS_DISPATCH_COUT(bucket, deconv_d_kernel, grid, dim3(256), 0, st, x, w_t, scale, shift, y, p);
    return ts::launched("deconv_d_kernel");
  }
  TS_DISPATCH_COUT(bucket, conv_d_kernel, grid, dim3(256), 0, st, x, w_t, scale, shift, y, p);
  return ts::launched("conv_d_kernel");
}

extern "C" int ts_conv_cout_pad(int cout) { return cout_bucket(cout); }
