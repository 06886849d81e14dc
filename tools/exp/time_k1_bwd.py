import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts
dev = torch.device("cuda:0")
for name, (B, C, H, W, D, sampled) in {"coarse": (1, 256, 34, 60, 12, False), "fine": (1, 128, 68, 120, 5, True), "precise": (1, 128, 136, 240, 5, True), "precise B=4": (4, 128, 136, 240, 5, True)}.items():
    L = torch.randn(B, C, H, W, device=dev, requires_grad=True); R = torch.randn(B, C, H, W, device=dev, requires_grad=True)
    disp = (torch.rand(B, D, H, W, device=dev) * 40).requires_grad_() if sampled else D
    out = ts.block_cost(L, R, disp, 3)
    g = torch.randn_like(out)
    def bwd():
        out.backward(g, retain_graph=True)
    for _ in range(3): bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bwd()
    e1.record(); torch.cuda.synchronize()
    nbytes = 4 * (out.numel() + 4 * L.numel() + (2 * disp.numel() if sampled else 0))
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("block_cost backward [%s]: %.1f us  (%.0f MB algorithmic -> %.0f GB/s)" % (name, us, nbytes / 1e6, nbytes / us / 1e3), flush=True)
