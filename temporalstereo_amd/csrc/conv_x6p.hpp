// Launch descriptor of ig_conv_x6p_kernel (conv_x6p.hip), filled by ts_conv3d_hw_x6_fwd (conv3d.hip).
#pragma once

namespace ts {

struct X6P {
  int Cin, Cout, coutp;         // coutp: padded channel count of the split weight / scale / shift arrays
  int B, D, H, W;
  int act;
  float act_param;
  long long in_bstride, in_cstride, out_bstride, out_cstride;
  unsigned in_bytes, w_bytes, out_bytes;      // extents of one batch element of x / of the weights / of one batch element of y
  const float* addend;          // [B][Cout][H*W] added to every depth plane's sum before scale/shift (or null)
  long long add_bstride, add_cstride;
  int xcd;                      // XCD-banded workgroup order
  int ksplit, kspan;            // split-K: this many slices of `kspan` 16-channel chunks each (1: unsplit)
  float* partial;               // [ksplit][B][Cout][D*H*W] raw sums when ksplit > 1
  unsigned part_bytes;          // extent of one (slice, batch) block of the partials
  int tiles_x, co_groups, tiles_pp, total_tiles;       // filled by x6p_launch (tiles per row / per plane / in all)
  int dbg;                      // experiment switches (TS_X6P_DBG; 0 in production)
  unsigned long long* trace;    // experiment: cycle stamps of workgroup trace_wg (TS_X6P_TRACE), else null
  int trace_wg;
};

// workgroups the launch would have (the caller's choice between this kernel and ig_conv_x6_kernel)
long long x6p_grid(const X6P& p);
int x6p_launch(const float* x, const void* w6, const float* scale, const float* shift, float* y, X6P p, void* stream);

}  // namespace ts
