#!/usr/bin/env python
"""the cost volume + first (1,3,3) layer of the sampled levels, timed two ways on BASELINE configs[1]
shapes (C ABI through the native wrappers, preallocated inputs, back-to-back launches between two HIP events):

  materialised  ts_block_cost_sampled_warped_fwd (volume without its reference half, 2 launches) + ts_conv3d_hw_fwd      [rounds 1-3]
  contracted    ts_block_cost_sampled_corr_fwd (correlation blocks, 2 launches) + ts_conv3d_hw_warp_fwd (gather + convolution)
                + the 1x1 pre-contraction right -> Q, which depends on the features only (issued at the start of a pass)

    python tools/fused_k1_bench.py [--iters 100] [--batches 1 4]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from temporalstereo_amd import functional as TF  # noqa: E402
from temporalstereo_amd.aggregation import native as N  # noqa: E402

LEVELS = {"fine": (128, 16, 68, 120, 5), "precise": (128, 8, 136, 240, 5)}


def timed(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for B in a.batches:
        for name, (C, cout, H, W, D) in LEVELS.items():
            torch.manual_seed(0)
            left, right = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
            yy = torch.linspace(0, 1, H, device=dev).view(1, 1, H, 1)
            xx = torch.linspace(0, 1, W, device=dev).view(1, 1, 1, W)
            base = 0.15 * W * (0.3 + 0.7 * yy) * (0.8 + 0.2 * torch.sin(6.28 * xx))
            steps = torch.tensor([0., 3., 4., 5., 8.], device=dev).view(1, 5, 1, 1)
            disp = (base + steps + 0.05 * torch.rand(B, 5, H, W, device=dev)).contiguous()
            w = torch.randn(cout, 2 * C + 3 * (C // 8), 1, 3, 3, device=dev) * 0.05
            f0 = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
            fl, rest, corr, q = N.split_sampled_first_layer(f0, 3)
            lterm = N.conv_hw(left.unsqueeze(2), fl, 1, 1)
            Q = N.conv_d(right.unsqueeze(2), q, 1).squeeze(2)
            t = {}
            t["k1_warped"] = timed(lambda: TF.block_cost_warped(left, right, disp, 3), a.iters)
            vol = TF.block_cost_warped(left, right, disp, 3)
            t["conv_rest"] = timed(lambda: N.conv_hw(vol, rest, 1, 1, addend=lterm), a.iters)
            t["materialised"] = timed(lambda: N.conv_hw(TF.block_cost_warped(left, right, disp, 3), rest, 1, 1, addend=lterm), a.iters)
            t["k1_corr"] = timed(lambda: N.block_cost_corr(left, right, disp, 3), a.iters)
            cv = N.block_cost_corr(left, right, disp, 3)
            t["conv_warp"] = timed(lambda: N.conv_hw_warp(cv, corr, Q, disp, lterm.squeeze(2), 1), a.iters)
            t["conv_corr_only"] = timed(lambda: N.conv_hw(cv, corr, 1, 1, addend=lterm), a.iters)
            t["q_1x1"] = timed(lambda: N.conv_d(right.unsqueeze(2), q, 1), a.iters)
            t["left_term"] = timed(lambda: N.conv_hw(left.unsqueeze(2), fl, 1, 1), a.iters)
            t["contracted"] = timed(lambda: N.conv_hw_warp(N.block_cost_corr(left, right, disp, 3), corr, Q, disp, lterm.squeeze(2), 1), a.iters)
            t["contracted_incl_q"] = t["contracted"] + t["q_1x1"]
            # the unfused op's algorithmic bytes over the time of the form that replaces it
            nb = 4 * B * H * W * (2 * C + D + (2 * C + 3 * C // 8) * D)
            row = dict(level=name, batch=B, us={k: round(v, 2) for k, v in t.items()}, unfused_algorithmic_bytes=nb,
                       fused_equivalent_TBps=nb / (t["contracted_incl_q"] * 1e-6) / 1e12)
            rows.append(row)
            print("%-8s B=%d  " % (name, B) + "  ".join("%s %.1f" % (k, v) for k, v in t.items()), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
