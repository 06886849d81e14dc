"""ts_block_cost_sampled_corr_fwd at the 1/4 and 1/8 levels: block_cost_fast<corr only> (TS_K1_CORR_ROWS=0) against
block_cost_corr_rows (default), both followed by the expansion launch; and that the two agree bit for bit."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SHAPES = ((1, 128, 136, 240, 5), (1, 128, 68, 120, 5), (4, 128, 136, 240, 5), (1, 128, 135, 240, 5), (2, 64, 34, 60, 7))
if len(sys.argv) > 1:
    import torch
    import temporalstereo_amd.functional as TF
    dev = torch.device("cuda:0")
    for (B, C, H, W, D) in SHAPES:
        torch.manual_seed(0)
        L = torch.randn(B, C, H, W, device=dev); R = torch.randn(B, C, H, W, device=dev)
        base = torch.rand(B, 1, H, W, device=dev) * 40
        disp = (base + torch.arange(D, device=dev).view(1, D, 1, 1) * 1.3 - 2).contiguous()
        for _ in range(20): out = TF.block_cost_corr(L, R, disp, 3)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            e0.record()
            for _ in range(200): TF.block_cost_corr(L, R, disp, 3)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 5)
        torch.save(out.cpu(), "/tmp/k1corr_%s_%d_%d_%d.pt" % (os.environ.get("TS_K1_CORR_ROWS", "1"), B, H, W))
        print("rows=%s  [%d,%d,%d,%d] x %d  %7.2f us" % (os.environ.get("TS_K1_CORR_ROWS", "1"), B, C, H, W, D, best), flush=True)
else:
    import torch
    for a in ("0", "1"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, TS_K1_CORR_ROWS=a))
    for (B, C, H, W, D) in SHAPES:
        a = torch.load("/tmp/k1corr_0_%d_%d_%d.pt" % (B, H, W)); b = torch.load("/tmp/k1corr_1_%d_%d_%d.pt" % (B, H, W))
        print("B=%d %dx%d: max |fast - rows| = %.3g  (bit-identical: %s)" % (B, H, W, (a - b).abs().max().item(), torch.equal(a, b)))
