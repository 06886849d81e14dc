"""Experiment: the bf16-split (x6) convolution against the f32-MFMA kernel, layer by layer, in isolation
(100 back-to-back launches between two events).  Shapes: the stride-1 (1,3,3) layers of config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from temporalstereo_amd.aggregation import native as N

dev = torch.device("cuda:0")
SHAPES = [  # name, Cin, Cout, D, H, W
    ("unet 32->32 272x480", 32, 32, 1, 272, 480),
    ("unet 64->64 136x240", 64, 64, 1, 136, 240),
    ("unet 128->32 272x480", 128, 32, 1, 272, 480),
    ("unet 64->32 272x480", 64, 32, 1, 272, 480),
    ("precise 176->8 D5 136x240", 176, 8, 5, 136, 240),
    ("fine 16->16 D8 68x120", 16, 16, 8, 68, 120),
    ("fine 32->32 D4 34x60", 32, 32, 4, 34, 60),
    ("coarse 32->32 D14 34x60", 32, 32, 14, 34, 60),
    ("coarse 64->64 D7 17x30", 64, 64, 7, 17, 30),
]
for B in (1, 4):
    for name, Cin, Cout, D, H, W in SHAPES:
        x = torch.randn(B, Cin, D, H, W, device=dev)
        w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
        f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
        out = torch.empty(B, Cout, D, H, W, device=dev)
        t = {}
        for x6 in (True, False):
            N.X6 = x6
            for _ in range(5):
                N.conv_hw(x, f, 1, 1, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                N.conv_hw(x, f, 1, 1, out=out)
            e1.record()
            torch.cuda.synchronize()
            t[x6] = e0.elapsed_time(e1) * 10.0
        fl = 2.0 * B * Cin * Cout * 9 * D * H * W
        print("B=%d %-28s f32 %7.1f us (%5.1f TF)   x6 %7.1f us (%5.1f TF eq.)   x%.2f" % (B, name, t[False], fl / t[False] / 1e6, t[True], fl / t[True] / 1e6, t[False] / t[True]))
