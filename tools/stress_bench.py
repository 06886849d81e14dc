#!/usr/bin/env python
"""The dense siblings of K1 as large-tensor HBM stress kernels: `cat_fms`,
`dif_fms` and the native 1-D correlation at shapes whose traffic is beyond the 256 MiB Infinity Cache (>= 1 GB where the op
allows), timed back to back between one pair of HIP events on the launch stream (no_grad: forward kernels only), with the
algorithmic bytes (inputs once + output once, fp32) and the fraction of the 8.0 TB/s spec peak.  K1's own beyond-the-cache
figure (batch 4, 930 MB) is printed next to them by tools/k1_bench.py / bench.py (`roofline.beyond_infinity_cache`).

    python tools/stress_bench.py [--iters 30]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import temporalstereo_amd as ts  # noqa: E402

PEAK = 8.0e12


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rows = []
    with torch.no_grad():
        # the survey's stress shape: [4, 32, 48, 136, 240] = 48 integer candidates over a KITTI-like 1/4-resolution map
        for (B, C, D, H, W) in [(4, 32, 48, 136, 240), (2, 32, 192, 136, 240)]:
            L, R = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
            disp = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1).expand(B, D, H, W).contiguous()
            for name, fn, out_ch in (("cat_fms", lambda: ts.cat_fms(L, R, disp), 2 * C), ("dif_fms", lambda: ts.dif_fms(L, R, disp), C)):
                nbytes = 4 * B * H * W * (2 * C + D + out_ch * D)
                t = timed(fn, a.iters)
                rows.append(dict(op=name, shape=[B, C, D, H, W], algorithmic_bytes=nbytes, mean_us=t * 1e6, achieved_GBps=nbytes / t / 1e9,
                                 frac_of_8TBps=nbytes / t / PEAK, note="dif_fms is two passes over the inputs (tensor-wide max first, dif_fms.py:38)" if name == "dif_fms" else ""))
        # native 1-D correlation (correlation.py:32-57): output [B, max_disp, H, W]; arithmetic intensity C MACs per output element
        for (B, C, D, H, W) in [(4, 32, 48, 136, 240), (8, 64, 192, 136, 240), (4, 32, 192, 272, 480)]:
            L, R = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
            nbytes = 4 * B * H * W * (2 * C + D)
            t = timed(lambda: ts.correlation1d(L, R, D), a.iters)
            rows.append(dict(op="correlation1d", shape=[B, C, D, H, W], algorithmic_bytes=nbytes, mean_us=t * 1e6, achieved_GBps=nbytes / t / 1e9,
                             frac_of_8TBps=nbytes / t / PEAK, flops=2.0 * B * C * D * H * W, tflops=2.0 * B * C * D * H * W / t / 1e12,
                             note="a band of the row's Gram matrix on the matrix cores (v_mfma_f32_16x16x4_f32: corr_row_mfma_kernel, csrc/correlation.hip)"))
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
