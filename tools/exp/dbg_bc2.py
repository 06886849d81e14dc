import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts, oracle, synth
from helpers import t
dev = torch.device("cuda:0")
for (B, C, H, W, D) in [(1, 8, 8, 12, 4), (1, 8, 8, 12, 5), (1, 8, 8, 12, 6), (1, 8, 8, 16, 5), (1, 8, 8, 12, 12), (1, 8, 4, 12, 5), (1, 16, 8, 12, 5)]:
    L = synth.normal(7, "L", (B, C, H, W)); R = synth.normal(7, "R", (B, C, H, W))
    exp = oracle.block_cost(t(L), t(R), D, 3).numpy()
    got = ts.block_cost(t(L, dev), t(R, dev), D, 3).cpu().numpy()
    diff = np.abs(got - exp); bad = np.argwhere(diff > 1e-4)
    print((B, C, H, W, D), "max", diff.max(), "nbad", len(bad))
    if len(bad):
        for ax, nm in enumerate("bcdyx"):
            print("  bad along", nm, np.bincount(bad[:, ax], minlength=diff.shape[ax]))
