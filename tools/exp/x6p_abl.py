"""Experiment: phase ablation of ig_conv_x6p_kernel through TS_X6P_DBG (1 no input fetch, 2 no matrix loop, 4 no commit, 8 no output stores, 16 no weight DMA)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SHAPES = [("128->32 272x480", 128, 32, 1, 272, 480), ("32->32 272x480", 32, 32, 1, 272, 480), ("64->64 136x240", 64, 64, 1, 136, 240)]
if "--child" in sys.argv:
    import torch
    from temporalstereo_amd.aggregation import native as N
    N._X6_MIN_GRID = 1
    dev = torch.device("cuda:0")
    row = []
    for B in (1, 4):
        for name, Cin, Cout, D, H, W in SHAPES:
            x = torch.randn(B, Cin, D, H, W, device=dev)
            w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
            f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
            out = torch.empty(B, Cout, D, H, W, device=dev)
            for _ in range(5):
                N.conv_hw(x, f, 1, 1, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                N.conv_hw(x, f, 1, 1, out=out)
            e1.record()
            torch.cuda.synchronize()
            row.append("%7.1f" % (e0.elapsed_time(e1) * 10.0))
    print("dbg %2s HR %s : %s" % (os.environ.get("TS_X6P_DBG", "0"), os.environ.get("TS_X6P_HR", "-"), " ".join(row)), flush=True)
    sys.exit(0)
print("columns: B=1 then B=4 of " + ", ".join(s[0] for s in SHAPES))
for hr in ("8",):
    for dbg in sys.argv[1:] or ("0", "1", "2", "4", "8", "16", "3", "6", "7", "15", "31", "17", "21", "29"):
        e = dict(os.environ); e.update({"TS_X6P_MIN_WGS": "1", "TS_X6P_HR": hr, "TS_X6P_DBG": dbg})
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, timeout=300)
