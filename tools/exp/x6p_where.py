"""Experiment: where ts_conv3d_hw_x6_fwd (ping-pong form forced) differs from the f32 kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from temporalstereo_amd.aggregation import native as N
N._X6_MIN_GRID = 1
B, Cin, Cout, D, H, W = [int(v) for v in sys.argv[1:7]]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
f = N.Folded(w, None, None, N.ACT_NONE, False, "hw")
out = N.conv_hw(x, f, 1, 1)
N.X6 = False
ref = N.conv_hw(x, f, 1, 1)
torch.cuda.synchronize()
bad = (out - ref).abs() > 1e-4
print("bad elements %d of %d" % (int(bad.sum()), bad.numel()))
idx = bad.nonzero()
if len(idx):
    for d, name in enumerate("b co d y x".split()):
        v = idx[:, d].unique()
        print(name, "n=%d" % len(v), v[:40].tolist())
    print("first", idx[0].tolist(), float(out[tuple(idx[0])]), float(ref[tuple(idx[0])]))
