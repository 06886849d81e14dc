"""The bf16-split forms of the strided / transposed convolutions (csrc/conv_x6s.hip) against the f32-input MFMA kernels they replace,
layer by layer, in isolation (100 back-to-back launches between two events).  Shapes: the stride-2 encoder layers and the two UNet
deconvolutions of config 2 (module.py:453-466), plus the largest stride-2 / transposed layers of the hourglasses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from temporalstereo_amd.aggregation import native as N

dev = torch.device("cuda:0")
N._X6S_MIN_GRID = N._X6S_MIN_GRID_T3 = 1
SHAPES = [  # name, mode, Cin, Cout, D, H, W, views (2: both views as one batch)
    ("enc 32->64 s2 272x480 x2 views", N.X6S_S2, 32, 64, 1, 272, 480, 2),
    ("hourglass 32->64 s2 D12 34x60", N.X6S_S2, 32, 64, 12, 32, 64, 1),
    ("precise 16->32 s2 D5 136x240", N.X6S_S2, 16, 32, 5, 136, 240, 1),
    ("deconv4 32->32 136x240", N.X6S_T4, 32, 32, 1, 136, 240, 1),
    ("deconv2 32->9 272x480", N.X6S_T4, 32, 9, 1, 272, 480, 1),
    ("precise 16->8^T D3 68x120", N.X6S_T3, 16, 8, 3, 68, 120, 1),
    ("fine 32->16^T D3 34x60", N.X6S_T3, 32, 16, 3, 34, 60, 1),
]
level = object.__new__(N.NativePrecise)
for B in (1, 4):
    for name, mode, Cin, Cout, D, H, W, views in SHAPES:
        Bx = B * views
        g = torch.Generator().manual_seed(1)
        if mode == N.X6S_T4:
            x = torch.randn(Bx, Cin, H, W, generator=g).to(dev)
            w = (torch.randn(Cin, Cout, 4, 4, generator=g) / (4 * Cin) ** 0.5).to(dev)
            f = N.Folded(w, None, None, N.ACT_RELU, True, "deconv2d")
            out = torch.empty(Bx, Cout, 2 * H, 2 * W, device=dev)
            run = lambda: level._deconv(x, f, out, out.stride(0))
            macs = Bx * Cin * Cout * 16 * H * W
        else:
            x = torch.randn(Bx, Cin, D, H, W, generator=g).to(dev)
            tr = mode == N.X6S_T3
            w = (torch.randn(*((Cin, Cout) if tr else (Cout, Cin)), 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
            f = N.Folded(w, None, None, N.ACT_SILU, tr, "hw")
            run = lambda: N.conv_hw(x, f, 2, 1, transposed=tr)
            macs = Bx * Cin * Cout * 9 * D * (H * W if tr else ((H + 1) // 2) * ((W + 1) // 2))
        t, outs = {}, {}
        for on in (True, False):
            N.X6S = on
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                r = run()
            e1.record()
            torch.cuda.synchronize()
            t[on] = e0.elapsed_time(e1) * 10.0
            outs[on] = (out if mode == N.X6S_T4 else r).clone()
        N.X6S = True
        err = float((outs[True] - outs[False]).abs().max()) / max(float(outs[False].abs().max()), 1.0)
        print("B=%d %-32s f32 %7.1f us (%5.1f TF)   x6s %7.1f us (%5.1f TF eq.)   x%.2f   max rel diff %.2e" %
              (B, name, t[False], 2.0 * macs / t[False] / 1e6, t[True], 2.0 * macs / t[True] / 1e6, t[False] / t[True], err), flush=True)
