"""Experiment: cycle stamps (s_memtime) of one workgroup of ig_conv_x6p_kernel, first wave of each half (TS_X6P_TRACE=<workgroup id>).
    TS_X6P_MIN_WGS=1 TS_X6P_HR=8 TS_X6P_TRACE=300 python tools/exp/x6p_trace.py [B Cin Cout H W]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from temporalstereo_amd import _lib
from temporalstereo_amd.aggregation import native as N

N._X6_MIN_GRID = 1
a = [int(v) for v in sys.argv[1:6]] or [4, 128, 32, 272, 480]
B, Cin, Cout, H, W = a
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, 1, H, W, device=dev)
w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
out = torch.empty(B, Cout, 1, H, W, device=dev)
for _ in range(3):
    N.conv_hw(x, f, 1, 1, out=out)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.lib()._name)
buf = np.zeros((4096 + 8000 * 32) // 8, dtype=np.uint64)
rc = L.ts_x6p_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
nch = (Cin + 15) // 16
names = ["start", "H1 idle barrier"]
for c in range(nch):
    names += ["c%d fetch landed" % c, "c%d committed" % c, "c%d barrier" % c, "c%d matrix loop" % c, "c%d barrier" % c]
names += ["epilogue staged", "barrier", "flushed"]
t0 = min(int(buf[0]), int(buf[256]))
print("B%d %d->%d %dx%d dbg=%s   (cycles since the workgroup's start: half 0 | half 1, and the step's duration)" % (B, Cin, Cout, H, W, os.environ.get("TS_X6P_DBG", "0")))
for i, n in enumerate(names):
    h0, h1 = int(buf[i]) - t0, int(buf[256 + i]) - t0
    d0 = int(buf[i]) - int(buf[i - 1]) if i else 0
    d1 = int(buf[256 + i]) - int(buf[256 + i - 1]) if i else 0
    print("%-22s %8d (+%6d) | %8d (+%6d)" % (n, h0, d0, h1, d1))
