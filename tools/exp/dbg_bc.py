import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts
from helpers import load, t
for name in sys.argv[1:]:
    g = load(name); dev = torch.device("cuda:0")
    if "num_disp" in g:
        out = ts.block_cost(t(g["left"], dev), t(g["right"], dev), int(g["num_disp"]), int(g["scales"]))
    else:
        out = ts.block_cost(t(g["left"], dev), t(g["right"], dev), t(g["disp"], dev), int(g["scales"]))
    diff = np.abs(out.cpu().numpy() - g["out"])
    bad = np.argwhere(diff > 1e-4)
    print(name, "shape", diff.shape, "max", diff.max(), "nbad", len(bad))
    if len(bad):
        for ax, nm in enumerate("bcdyx"):
            print("  bad along", nm, np.bincount(bad[:, ax], minlength=diff.shape[ax]))
        print("  first:", bad[:5].tolist())
    if len(bad) and "num_disp" in g:
        o = out.cpu().numpy(); L = g["left"]; R = g["right"]
        for (b, c, d, y, x) in bad[:6].tolist():
            got = o[b, c, d, y, x]; l = L[b, c, y, x]
            # which source value reproduces got = -(l - r)^2 ?
            cands = {(cc, xx): -(l - R[b, cc, y, xx]) ** 2 for cc in range(L.shape[1]) for xx in range(L.shape[3])}
            cands[("zero", 0)] = -(l) ** 2
            best = min(cands, key=lambda k: abs(cands[k] - got))
            print("  at c%d d%d y%d x%d got %.5f exp %.5f -> matches source %s (err %.2g)" % (c, d, y, x, got, g["out"][b, c, d, y, x], best, abs(cands[best] - got)))
