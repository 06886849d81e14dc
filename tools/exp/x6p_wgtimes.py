"""Experiment: start / end cycle stamps of every workgroup of one ig_conv_x6p_kernel launch (TS_X6P_TRACE=-2).
    TS_X6P_MIN_WGS=1 TS_X6P_HR=8 TS_X6P_TRACE=-2 python tools/exp/x6p_wgtimes.py [B Cin Cout H W]"""
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); N.conv_hw(x, f, 1, 1, out=out); e1.record(); torch.cuda.synchronize()
L = ctypes.CDLL(_lib.lib()._name)
buf = np.zeros((4096 + 8000 * 32) // 8, dtype=np.uint64)
assert L.ts_x6p_trace_read(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[512:].reshape(-1, 4).astype(np.int64)
n = int((t[:, 0] != 0).sum())
t = t[:n]
dur = t[:, 1] - t[:, 0]
real = (t[:, 2] - t[:, 3]) * 10.0          # s_memrealtime: 100 MHz
print("B%d %d->%d %dx%d: %d workgroups, launch %.1f us by events" % (B, Cin, Cout, H, W, n, e0.elapsed_time(e1) * 1e3))
print("workgroup duration, s_memtime ticks: min %d  median %d  max %d;  ns (s_memrealtime): min %d median %d max %d;  ticks/ns median %.3f" %
      (dur.min(), np.median(dur), dur.max(), real.min(), np.median(real), real.max(), np.median(dur / np.maximum(real, 1))))
r0 = t[:, 3].min()
st = np.sort((t[:, 3] - r0) * 10.0)
en = np.sort((t[:, 2] - r0) * 10.0)
print("start ns (sorted), every 64th:", st[::64].astype(int).tolist())
print("end ns (sorted), every 64th:", en[::64].astype(int).tolist(), "last", int(en[-1]))
