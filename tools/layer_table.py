#!/usr/bin/env python
"""Every C-ABI launch of one BASELINE configs[1] pass (or --batch N), in issue order, timed ALONE: the call is re-issued N times
back to back with the very arguments the pass used, between two HIP events.  For the convolutions the table adds the layer's
multiply-adds and the rate they run at.  This is the per-layer view the kernel trace cannot give (it groups by kernel and grid).

    python tools/layer_table.py [--batch 1] [--iters 100] [--json out.json] [--sort]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd import _lib  # noqa: E402
from temporalstereo_amd.aggregation.engine import InferenceEngine  # noqa: E402


class Spy:
    """Stands in for the library while one eager pass is issued: runs every call and remembers (name, args)."""

    def __init__(self):
        self.calls, self.keep = [], []
        self.real = _lib._real_lib()

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if name in _lib._QUERIES:
            return fn

        def call(*args):
            self.calls.append((name, args))
            return fn(*args)
        return call


def macs_of(name, a):
    """Multiply-adds of a convolution call from its integer arguments (include/ts_hip.h order)."""
    if name == "ts_conv3d_hw_fwd":
        B, Cin, Cout, D, H, W, stride, dil, transposed = a[5:14]
        if transposed:
            return B * Cin * Cout * D * H * W * 9, "(1,3,3)^T %d->%d  %dx%dx%d B%d" % (Cin, Cout, D, H, W, B)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        return B * Cin * Cout * D * Ho * Wo * 9, "(1,3,3) s%d d%d %d->%d  %dx%dx%d B%d" % (stride, dil, Cin, Cout, D, H, W, B)
    if name == "ts_conv3d_hw_x6_fwd":
        B, Cin, Cout, D, H, W, dil = a[5:12]
        return B * Cin * Cout * D * H * W * 9, "x6 (1,3,3) d%d %d->%d  %dx%dx%d B%d" % (dil, Cin, Cout, D, H, W, B)
    if name == "ts_conv3d_hw_x6s_fwd":
        B, Cin, Cout, D, H, W, mode = a[5:12]
        if mode == 0:
            return B * Cin * Cout * D * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) * 9, "x6s (1,3,3) s2 %d->%d  %dx%dx%d B%d" % (Cin, Cout, D, H, W, B)
        if mode == 1:
            return B * Cin * Cout * D * H * W * 9, "x6s (1,3,3)^T %d->%d  %dx%dx%d B%d" % (Cin, Cout, D, H, W, B)
        return B * Cin * Cout * H * W * 16, "x6s deconv 4x4 s2 %d->%d  %dx%d B%d" % (Cin, Cout, H, W, B)
    if name == "ts_conv3d_hw_warp_fwd":
        B, Cc, Cout, D, H, W = a[8:14]
        return B * Cc * Cout * D * H * W * 9, "warp (1,3,3) %d->%d  %dx%dx%d B%d (+ gather)" % (Cc, Cout, D, H, W, B)
    if name == "ts_conv3d_d_fwd":
        B, Cin, Cout, Din, H, W, k, stride, dil, pad, transposed = a[5:16]
        Do = 2 * Din if transposed else (Din + 2 * pad - dil * (k - 1) - 1) // stride + 1
        taps = k if not transposed else 1.5
        return int(B * Cin * Cout * Do * H * W * taps), "(%d,1,1)%s s%d %d->%d  %dx%dx%d B%d" % (k, "^T" if transposed else "", stride, Cin, Cout, Din, H, W, B)
    if name == "ts_deconv2d_k4s2_fwd":
        B, Cin, Cout, H, W = a[5:10]
        return B * Cin * Cout * H * W * 16, "deconv 4x4 s2 %d->%d  %dx%d B%d" % (Cin, Cout, H, W, B)
    return None, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--json", default=None)
    ap.add_argument("--sort", action="store_true", help="largest first instead of issue order")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 2
    net = bench.load_trained(bench.build_model(dev, seed)).eval()
    inputs, _ = bench.make_planted_inputs(dev, seed, a.batch)
    eng = InferenceEngine(net, backend="native", replay="eager", inputs="bind", pipeline=1)
    eng.net.overlap = False                               # one stream: the issue order is the dependency order
    with torch.no_grad():
        eng(*inputs, {})
        torch.cuda.synchronize()
        spy = Spy()
        orig = _lib.lib
        keep = []
        optr = _lib.ptr

        def ptr(t):
            if t is not None:
                keep.append(t)
            return optr(t)
        _lib.lib, _lib.ptr = (lambda: spy), ptr
        try:
            eng(*inputs, {})
        finally:
            _lib.lib, _lib.ptr = orig, optr
    torch.cuda.synchronize()
    rows = []
    for name, args in spy.calls:
        if name in ("ts_stream_fork", "ts_event_record", "ts_event_wait", "ts_conv_set_chunk_cap"):
            continue
        fn = getattr(spy.real, name)
        for _ in range(5):
            _lib.check(fn(*args), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        ints = [x for x in args if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 31)]
        macs, what = macs_of(name, list(args))
        rows.append(dict(call=name, what=what, us=us, gmac=(macs / 1e9 if macs else None), tflops=(2 * macs / us / 1e6 if macs else None), ints=ints[:14]))
    total = sum(r["us"] for r in rows)
    order = sorted(rows, key=lambda r: -r["us"]) if a.sort else rows
    print("%d launches, %.1f us when each runs alone (batch %d)" % (len(rows), total, a.batch))
    for r in order:
        print("%-34s %7.1f us %5.1f%%  %s%s" % (r["call"][3:], r["us"], 100 * r["us"] / total, r["what"] or str(r["ints"][:10]),
                                               ("   %.3f GMAC %.1f TF/s" % (r["gmac"], r["tflops"])) if r["gmac"] else ""))
    fam = {}
    for r in rows:
        fam.setdefault(r["call"], [0, 0.0])
        fam[r["call"]][0] += 1; fam[r["call"]][1] += r["us"]
    print("--- by entry point")
    for k, (n, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-36s %3d launches %8.1f us %5.1f%%" % (k, n, t, 100 * t / total))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(batch=a.batch, total_us=total, rows=rows), f, indent=1)


if __name__ == "__main__":
    main()
