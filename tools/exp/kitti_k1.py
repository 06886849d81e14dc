import os, sys; sys.path.insert(0, ".")
import torch
import temporalstereo_amd.functional as TF
dev = torch.device("cuda:0")
B, C, H, W, D = 2, 128, 96, 312, 5
torch.manual_seed(0)
L = torch.randn(B, C, H, W, device=dev, requires_grad=True); R = torch.randn(B, C, H, W, device=dev, requires_grad=True)
disp = (torch.rand(B, 1, H, W, device=dev) * 40 + torch.arange(D, device=dev).view(1, D, 1, 1) * 1.3 - 2).contiguous().requires_grad_(True)
def t(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
with torch.no_grad():
    tc = t(lambda: TF.block_cost_corr(L, R, disp, 3))
out = TF.block_cost(L, R, disp, 3)
g = torch.randn_like(out)
def bwd():
    L.grad = R.grad = disp.grad = None
    out.backward(g, retain_graph=True)
tb = t(bwd, 30)
print("KITTI 1/4 level [2,128,96,312] x 5: corr_fwd %.1f us, backward (autograd call) %.1f us  (CORR_ROWS=%s BWD_ROWS=%s)" % (tc, tb, os.environ.get("TS_K1_CORR_ROWS", "1"), os.environ.get("TS_K1_BWD_ROWS", "1")))
