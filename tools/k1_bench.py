#!/usr/bin/env python
"""K1 (cost-volume build) timed through the C ABI with preallocated buffers, back-to-back launches between two HIP events
on the launch stream: the three levels of BASELINE configs[1] at batch 1 (output inside the 256 MiB Infinity Cache) and at
a batch whose traffic exceeds 1 GB (SURVEY.md section 8(d) hygiene), the complete op and the pipeline's warped variant.

    python tools/k1_bench.py [--iters 200] [--batches 1 4]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from temporalstereo_amd import _lib  # noqa: E402

LEVELS = {"coarse": (256, 34, 60, 12, False), "fine": (128, 68, 120, 5, True), "precise": (128, 136, 240, 5, True)}


def alg_bytes(B, C, H, W, D, kind):
    if kind == "warped":
        return 4 * B * H * W * (2 * C + D + (C + 3 * C // 8) * D)
    if kind == "sampled":
        return 4 * B * H * W * (2 * C + D + (2 * C + 3 * C // 8) * D)
    return 4 * B * H * W * (2 * C + (C + 3 * C // 8) * D)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--levels", nargs="+", default=list(LEVELS))
    ap.add_argument("--backward", action="store_true", help="also time the backward (ts_block_cost_{int,sampled}_bwd) of each level")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for B in a.batches:
        for name in a.levels:
            C, H, W, D, sampled = LEVELS[name]
            torch.manual_seed(0)
            left, right = torch.randn(B, C, H, W, device=dev), torch.randn(B, C, H, W, device=dev)
            yy = torch.linspace(0, 1, H, device=dev).view(1, 1, H, 1)
            xx = torch.linspace(0, 1, W, device=dev).view(1, 1, 1, W)
            base = 0.15 * W * (0.3 + 0.7 * yy) * (0.8 + 0.2 * torch.sin(6.28 * xx))
            steps = torch.tensor([0., 3., 4., 5., 8.], device=dev).view(1, 5, 1, 1)
            disp = (base + steps + 0.05 * torch.rand(B, 5, H, W, device=dev)).contiguous()
            ws = torch.zeros(max(int(L.ts_block_cost_workspace_bytes(B, C, H, W, D, 3)), 256), device=dev, dtype=torch.uint8)
            kinds = ("sampled", "warped") if sampled else ("int",)
            for kind in kinds:
                ctot = {"sampled": 2 * C, "warped": C, "int": C}[kind] + 3 * (C // 8)
                out = torch.empty(B, ctot, D, H, W, device=dev)
                if kind == "int":
                    fn = lambda: L.ts_block_cost_int_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(out), _lib.ptr(ws), B, C, H, W, D, 3, st)
                elif kind == "sampled":
                    fn = lambda: L.ts_block_cost_sampled_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out), _lib.ptr(ws), B, C, H, W, D, 3, st)
                else:
                    fn = lambda: L.ts_block_cost_sampled_warped_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out), _lib.ptr(ws), B, C, H, W, D, 3, st)
                for _ in range(20):
                    _lib.check(fn(), "k1")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / a.iters * 1e-3
                nb = alg_bytes(B, C, H, W, D, kind)
                row = dict(level=name, kind=kind, batch=B, us=t * 1e6, algorithmic_bytes=nb, GBps=nb / t / 1e9, frac_of_8TBps=nb / t / 8e12,
                           frac_of_6p29TBps=nb / t / 6.29e12)
                rows.append(row)
                print("K1 %-8s %-8s B=%d  %8.2f us  %8.1f MB  %7.1f GB/s  %.3f of 8.0 TB/s  %.3f of 6.29 TB/s" % (
                    name, kind, B, t * 1e6, nb / 1e6, nb / t / 1e9, nb / t / 8e12, nb / t / 6.29e12), flush=True)
                del out
            if a.backward:
                # backward of the complete op: reads left, right, (disp,) grad_out once, writes grad_left, grad_right (, grad_disp) once
                ctot = (2 * C if sampled else C) + 3 * (C // 8)
                go = torch.randn(B, ctot, D, H, W, device=dev)
                gl, gr, gd = torch.empty_like(left), torch.empty_like(right), torch.empty(B, D, H, W, device=dev)
                wsb = torch.zeros(max(int(L.ts_block_cost_bwd_workspace_bytes(B, C, H, W, D, 3)), 256), device=dev, dtype=torch.uint8)
                if sampled:
                    fn = lambda: L.ts_block_cost_sampled_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(go), _lib.ptr(gl), _lib.ptr(gr),
                                                             _lib.ptr(gd), _lib.ptr(wsb), B, C, H, W, D, 3, st)
                else:
                    fn = lambda: L.ts_block_cost_int_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(go), _lib.ptr(gl), _lib.ptr(gr), _lib.ptr(wsb),
                                                         B, C, H, W, D, 3, st)
                for _ in range(5):
                    _lib.check(fn(), "k1 bwd")
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = max(a.iters // 4, 10)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) / n * 1e-3
                nb = 4 * B * H * W * (4 * C + (2 * D if sampled else 0) + ctot * D)
                rows.append(dict(level=name, kind="backward", batch=B, us=t * 1e6, algorithmic_bytes=nb, GBps=nb / t / 1e9, frac_of_8TBps=nb / t / 8e12))
                print("K1 %-8s %-8s B=%d  %8.2f us  %8.1f MB  %7.1f GB/s  %.3f of 8.0 TB/s" % (name, "backward", B, t * 1e6, nb / 1e6, nb / t / 1e9, nb / t / 8e12), flush=True)
                del go
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
