#!/usr/bin/env python
"""Does a small convolution run slower when its weights come from a buffer nobody touched recently?  (GPU box only.)

A chain of N launches of one small (1,3,3) layer, back to back on one stream, timed between two events:
  same     every launch reads the SAME laid-out weight buffer
  distinct launch i reads buffer i of N distinct ones (all written long before)
  evicted  as `distinct`, with a 1 GiB fill between repetitions (nothing of the weights is left in any cache)
  relaid   launch i is preceded by a layout launch into ONE recycled buffer (what the replayed training step does per call)
usage: python tools/exp/cold_weights_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from temporalstereo_amd import _lib  # noqa: E402
from temporalstereo_amd import functional as TF  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = TF._stream()
    N = 64
    for (Cin, Cout, D, H, W) in ((32, 32, 12, 34, 60), (64, 64, 6, 17, 30), (16, 16, 5, 68, 120), (8, 8, 5, 136, 240)):
        x = torch.randn(1, Cin, D, H, W, device=dev)
        y = torch.empty(1, Cout, D, H, W, device=dev)
        ws = [torch.randn(Cout, Cin, 1, 3, 3, device=dev) for _ in range(N)]
        lay = [TF._layout_now(w, 1, 0, False)[0] for w in ws]
        big = torch.empty(256 << 20, device=dev)
        recycled = torch.empty_like(lay[0])
        cp = TF._cpad(Cout)

        def conv(wt):
            _lib.check(L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(wt), None, None, _lib.ptr(y), 1, Cin, Cout, D, H, W, 1, 1, 0, 0, 0.0,
                                          x.stride(0), x.stride(1), y.stride(0), y.stride(1), None, 0, None, 0, st), "conv")

        def relay(w):
            _lib.check(L.ts_conv_weight_layout(_lib.ptr(w), _lib.ptr(recycled), Cin, 9, Cout, cp, 9, Cin * 9, 1, 0, st), "layout")

        def run(kind, reps=20):
            tot = 0.0
            for r in range(reps + 2):
                if kind == "evicted":
                    big.fill_(1.0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(N):
                    if kind == "same":
                        conv(lay[0])
                    elif kind == "relaid":
                        relay(ws[i]); conv(recycled)
                    else:
                        conv(lay[i])
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    tot += e0.elapsed_time(e1)
            return tot / reps / N * 1e3
        print("%3d->%3d on %2dx%3dx%3d   us per launch: same %.2f  distinct %.2f  evicted %.2f  relaid (layout + conv) %.2f"
              % (Cin, Cout, D, H, W, run("same"), run("distinct"), run("evicted"), run("relaid")), flush=True)


if __name__ == "__main__":
    main()
