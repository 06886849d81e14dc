#!/usr/bin/env python
"""Kernel-level timing of the hot-path ops at BASELINE config-2 sizes (GPU box only).

usage: python tools/microbench.py [--iters N] [--batch B]
Prints one line per op: mean time, algorithmic bytes, achieved GB/s, fraction of 8.0 / 6.29 TB/s.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts  # noqa: E402

LEVELS = {  # name: (C, H, W, D, sampled)   -- SURVEY.md section 8(a) K1a/K1b, config 2
    "coarse": (256, 34, 60, 12, False),
    "fine": (128, 68, 120, 5, True),
    "precise": (128, 136, 240, 5, True),
}


def alg_bytes(B, C, H, W, D, sampled):
    # inputs once + output once, fp32
    if sampled:
        return 4 * B * H * W * (2 * C + D + (2 * C + 3 * C // 8) * D)
    return 4 * B * H * W * (2 * C + (C + 3 * C // 8) * D)


def time_op(fn, iters, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e-3


def calibrate(dev, nbytes, iters):
    from temporalstereo_amd import _lib
    L = _lib.lib()
    a = torch.empty(nbytes // 4, device=dev); b = torch.ones(nbytes // 4, device=dev)
    st = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for kind, name, mult in ((0, "fill ", 1), (1, "copy ", 2), (2, "read ", 1)):
        tsec = time_op(lambda: _lib.check(L.ts_calib_stream(kind, _lib.ptr(a), _lib.ptr(b), nbytes, st), "calib"), iters)
        print("calib[%s] %7.1f MB  %8.1f us  %7.1f GB/s" % (name, nbytes / 1e6, tsec * 1e6, mult * nbytes / tsec / 1e9), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    tot_b, tot_t = 0, 0.0
    for mb in (64, 200, 1024, 4096):
        calibrate(dev, mb * 1000 * 1000 // 16 * 16, 50)
    for name, (C, H, W, D, sampled) in LEVELS.items():
        torch.manual_seed(0)
        L = torch.randn(B, C, H, W, device=dev)
        R = torch.randn(B, C, H, W, device=dev)
        if sampled:   # piecewise-smooth candidates like the pyramid produces (base +- range), not white noise
            yy = torch.linspace(0, 1, H, device=dev).view(1, 1, H, 1)
            xx = torch.linspace(0, 1, W, device=dev).view(1, 1, 1, W)
            base = 4.0 + 0.15 * W * (0.3 + 0.7 * yy) * (0.8 + 0.2 * torch.sin(6.28 * xx))
            steps = torch.tensor([0., 3., 4., 5., 8.], device=dev)[:D].view(1, D, 1, 1)
            disp = (base - 4.0 + steps + 0.05 * torch.rand(B, D, H, W, device=dev)).contiguous()
        else:
            disp = D
        tsec = time_op(lambda: ts.block_cost(L, R, disp, 3), a.iters)
        nb = alg_bytes(B, C, H, W, D, sampled)
        tot_b += nb
        tot_t += tsec
        print("block_cost[%-7s] B=%d  %8.1f us  %7.1f MB  %7.1f GB/s  %.3f of 8.0TB/s  %.3f of 6.29TB/s" % (
            name, B, tsec * 1e6, nb / 1e6, nb / tsec / 1e9, nb / tsec / 8.0e12, nb / tsec / 6.29e12), flush=True)
        if sampled:   # the inference form without the repeated left half (what the native pipeline launches)
            from temporalstereo_amd import functional as TF
            tw = time_op(lambda: TF.block_cost_warped(L, R, disp, 3), a.iters)
            nw = 4 * B * H * W * (2 * C + D + (C + 3 * C // 8) * D)
            print("  warped [%-7s] B=%d  %8.1f us  %7.1f MB  %7.1f GB/s  %.3f of 8.0TB/s" % (
                name, B, tw * 1e6, nw / 1e6, nw / tw / 1e9, nw / tw / 8.0e12), flush=True)
    print("block_cost[total  ] B=%d  %8.1f us  %7.1f MB  %7.1f GB/s  %.3f of 8.0TB/s" % (
        B, tot_t * 1e6, tot_b / 1e6, tot_b / tot_t / 1e9, tot_b / tot_t / 8.0e12))

    # The reference's own micro-benchmarks (KITTI 1/4 resolution 96x312, forward only, batch 1; GTX3090 figures
    # pasted in its sources: block_cost.py:88-92 1.7147 ms, cat_fms.py:41-42 5.3421 ms, dif_fms.py:49-50 8.3691 ms)
    from temporalstereo_amd import functional as TF
    H, W = 384 // 4, 1248 // 4
    L48, R48 = torch.rand(1, 48, H, W, device=dev), torch.rand(1, 48, H, W, device=dev)
    d4 = torch.linspace(0, 3, 4, device=dev).view(1, 4, 1, 1).expand(1, 4, H, W).contiguous()
    tsec = time_op(lambda: ts.block_cost(L48, R48, d4, 3), a.iters)
    nb = alg_bytes(1, 48, H, W, 4, True)
    print("ref-bench block_cost C=48 96x312 D=4   %8.1f us  %7.1f MB  %7.1f GB/s   (reference: 1714.7 us on a GTX3090)" % (tsec * 1e6, nb / 1e6, nb / tsec / 1e9))
    L32, R32 = torch.rand(1, 32, H, W, device=dev), torch.rand(1, 32, H, W, device=dev)
    d48 = torch.linspace(0, 47, 48, device=dev).view(1, 48, 1, 1).expand(1, 48, H, W).contiguous()
    for name, fn, ch, ref_us in (("cat_fms", TF.cat_fms, 64, 5342.1), ("dif_fms", TF.dif_fms, 32, 8369.1)):
        tsec = time_op(lambda: fn(L32, R32, d48), max(a.iters // 4, 20))
        nb = 4 * H * W * (2 * 32 + 48 + ch * 48)
        print("ref-bench %-7s C=32 96x312 D=48     %8.1f us  %7.1f MB  %7.1f GB/s   (reference: %.1f us on a GTX3090)" % (name, tsec * 1e6, nb / 1e6, nb / tsec / 1e9, ref_us))
    # the large-tensor stress shape of SURVEY.md section 8(f)-3
    Lb, Rb = torch.rand(4, 32, 136, 240, device=dev), torch.rand(4, 32, 136, 240, device=dev)
    db = torch.linspace(0, 47, 48, device=dev).view(1, 48, 1, 1).expand(4, 48, 136, 240).contiguous()
    for name, fn, ch in (("cat_fms", TF.cat_fms, 64), ("dif_fms", TF.dif_fms, 32)):
        tsec = time_op(lambda: fn(Lb, Rb, db), 20, warmup=5)
        nb = 4 * 4 * 136 * 240 * (2 * 32 + 48 + ch * 48)
        print("stress    %-7s [4,32,48,136,240]      %8.1f us  %7.1f MB  %7.1f GB/s  %.3f of 8.0TB/s" % (name, tsec * 1e6, nb / 1e6, nb / tsec / 1e9, nb / tsec / 8.0e12))


if __name__ == "__main__":
    main()
