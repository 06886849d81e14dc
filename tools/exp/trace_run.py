"""Experimental: per-workgroup timeline of block_cost_main (uses tools/exp/block_cost_trace.hip)."""
import ctypes, os, subprocess, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
so = os.path.join(HERE, "libtrace.so")
lib = ctypes.CDLL(so)
B, C, H, W, D = int(os.environ.get("B", 1)), 128, 136, 240, 5
dev = torch.device("cuda:0")
L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
yy = torch.linspace(0, 1, H, device=dev).view(1, 1, H, 1); xx = torch.linspace(0, 1, W, device=dev).view(1, 1, 1, W)
base = 4.0 + 0.15 * W * (0.3 + 0.7 * yy) * (0.8 + 0.2 * torch.sin(6.28 * xx))
steps = torch.tensor([0., 3., 4., 5., 8.], device=dev).view(1, D, 1, 1)
disp = (base - 4.0 + steps + 0.05 * torch.rand(B, D, H, W, device=dev)).contiguous()
out = torch.empty(B, 2 * C + 3 * C // 8, D, H, W, device=dev)
lib.ts_block_cost_workspace_bytes.restype = ctypes.c_size_t
ws = torch.empty(lib.ts_block_cost_workspace_bytes(B, C, H, W, D, 3), dtype=torch.uint8, device=dev)
nwg = 34 * 16 * B
trace = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run():
    rc = lib.ts_block_cost_sampled_fwd(P(L), P(R), P(disp), P(out), P(ws), B, C, H, W, D, 3, None)
    assert rc == 0, rc
for _ in range(5): run()
torch.cuda.synchronize()
ctypes.c_void_p.in_dll(lib, "g_trace").value = trace.data_ptr()
run(); torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(nwg, 8)

w0 = t[:, 5].min()
ws_, we_ = (t[:, 5] - w0) / 100.0, (t[:, 6] - w0) / 100.0     # us (100 MHz wall clock)
print("wall: kernel span %.1f us; WG wall duration mean %.1f us (min %.1f max %.1f); cycles/WG mean %.0f -> %.2f GHz" % (
    we_.max(), (we_ - ws_).mean(), (we_ - ws_).min(), (we_ - ws_).max(), (t[:, 2] - t[:, 0]).mean(), (t[:, 2] - t[:, 0]).mean() / ((we_ - ws_).mean() * 1e3)))
print("start pct (us):", np.round(np.percentile(ws_, [0, 10, 25, 50, 75, 90, 100]), 1))
print("end   pct (us):", np.round(np.percentile(we_, [0, 10, 25, 50, 75, 90, 100]), 1))
xcc = t[:, 3] & 0xf
hw = t[:, 4]
key = xcc * (1 << 32) + (hw & 0xfffff00)     # drop wave/simd bits
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("distinct CUs:", len(u), "WGs per CU: min %d max %d" % (cnt.min(), cnt.max()))
maxc = []
for ci in range(len(u)):
    idx = np.where(inv == ci)[0]
    ev = sorted([(ws_[i], 1) for i in idx] + [(we_[i], -1) for i in idx])
    c = m = 0
    for _, dlt in ev:
        c += dlt; m = max(m, c)
    maxc.append(m)
print("max concurrent WGs on a CU histogram:", np.bincount(maxc))
for ci in range(3):
    idx = np.where(inv == ci)[0]
    print("CU", hex(int(u[ci])), [(round(float(ws_[i]), 1), round(float(we_[i]), 1)) for i in idx])
