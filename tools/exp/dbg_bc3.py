import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts, oracle, synth
from helpers import t
dev = torch.device("cuda:0")
B, C, H, W, D = 1, int(os.environ.get("C", 256)), 34, 60, 12
L = synth.normal(9, "L", (B, C, H, W)); R = synth.normal(9, "R", (B, C, H, W))
ds = np.broadcast_to(np.arange(D, dtype=np.float32).reshape(1, D, 1, 1), (B, D, H, W)).copy()
for mode in ("int", "sampled"):
    arg = D if mode == "int" else t(ds)
    exp = oracle.block_cost(t(L), t(R), arg, 3).numpy()
    got = ts.block_cost(t(L, dev), t(R, dev), D if mode == "int" else t(ds, dev), 3).cpu().numpy()
    diff = np.abs(got - exp); bad = np.argwhere(diff > 1e-3)
    print(mode, "max", diff.max(), "nbad", len(bad))
    if len(bad):
        for ax, nm in enumerate("bcdyx"):
            print("  bad along", nm, np.bincount(bad[:, ax], minlength=diff.shape[ax]).tolist())
