"""One x6 layer launched a few times (for rocprofv3 --pmc passes): python tools/exp/x6_pmc_case.py B Cin Cout D H W [x6|x6s2|f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from temporalstereo_amd.aggregation import native as N
B, Cin, Cout, D, H, W = [int(v) for v in sys.argv[1:7]]
kind = sys.argv[7] if len(sys.argv) > 7 else "x6"
dev = torch.device("cuda:0")
N._X6_MIN_GRID = N._X6S_MIN_GRID = 1
N.X6 = kind != "f32"
x = torch.randn(B, Cin, D, H, W, device=dev)
w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
stride = 2 if kind == "x6s2" else 1
for _ in range(6):
    N.conv_hw(x, f, stride, 1)
torch.cuda.synchronize()
