#!/usr/bin/env python
"""Train-mode BatchNorm + activation of one layer, forward and backward: the one-launch form against the two-launch form per size
(GPU box only).  us per call, 200 calls back to back between two events."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from temporalstereo_amd import _lib
from temporalstereo_amd import functional as TF


def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = TF._stream()
    for (C, D, H, W) in ((64, 3, 9, 15), (64, 6, 17, 30), (32, 2, 17, 30), (32, 3, 34, 60), (32, 12, 34, 60), (32, 14, 34, 60), (16, 3, 68, 120),
                         (16, 5, 68, 120), (16, 8, 68, 120), (64, 1, 68, 120), (64, 1, 34, 60), (48, 1, 136, 240), (8, 3, 136, 240)):
        N = D * H * W
        y = torch.randn(1, C, D, H, W, device=dev); g = torch.randn_like(y); out = torch.empty_like(y); dy = torch.empty_like(y)
        mean = torch.empty(C, device=dev); var = torch.empty(C, device=dev); s1 = torch.empty(C, device=dev); s2 = torch.empty(C, device=dev)
        gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
        ws = torch.empty(int(L.ts_bn_workspace_bytes(1, C, N)), device=dev, dtype=torch.uint8)
        res = []
        for cap in (1 << 30, 0):
            L.ts_bn_set_small_elems(cap)
            def fwd():
                _lib.check(L.ts_bn_train_fwd(_lib.ptr(y), _lib.ptr(mean), _lib.ptr(var), None, None, 0.1, None, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out), _lib.ptr(ws),
                                             1, C, N, C * N, N, C * N, N, 1e-5, 1, st), "f")
            def bwd():
                _lib.check(L.ts_bn_train_bwd(_lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(dy),
                                             _lib.ptr(ws), 1, C, N, C * N, N, C * N, N, 1e-5, 1, float(N), st), "b")
            for fn in (fwd, bwd):
                for _ in range(20): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200): fn()
                e1.record(); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 200 * 1e3)
        print("C %3d N %7d   fwd one launch %6.2f  two %6.2f    bwd one %6.2f  two %6.2f" % (C, N, res[0], res[2], res[1], res[3]), flush=True)
    L.ts_bn_set_small_elems(-1)


if __name__ == "__main__":
    main()
