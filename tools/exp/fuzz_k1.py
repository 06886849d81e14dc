"""Random-shape fuzz of the cost-volume kernels against the oracle (GPU box)."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, oracle
import oracle.cost_volume as ocv
import temporalstereo_amd as ts
from temporalstereo_amd import functional as TF
dev = torch.device("cuda:0")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for it in range(N):
    B = int(rng.choice([1, 2, 3])); C = int(rng.choice([8, 16, 32, 64, 128])); H = int(rng.randint(4, 70)); W = int(rng.choice([rng.randint(4, 130), 4 * rng.randint(1, 80)]))
    D = int(rng.randint(2, 13)); sc = int(rng.choice([1, 2, 3]))
    l = torch.from_numpy(synth.normal(it, "l", (B, C, H, W))); r = torch.from_numpy(synth.normal(it, "r", (B, C, H, W)))
    d = torch.from_numpy(synth.uniform(it, "d", (B, D, H, W), -3.0, W * 0.6))
    errs = {}
    exp = oracle.block_cost(l, r, d, sc)
    for rep in range(2):
        errs["sampled"] = max(errs.get("sampled", 0), float((ts.block_cost(l.to(dev), r.to(dev), d.to(dev), sc).cpu() - exp).abs().max()))
        errs["warped"] = max(errs.get("warped", 0), float((TF.block_cost_warped(l.to(dev), r.to(dev), d.to(dev), sc).cpu() - exp[:, C:]).abs().max()))
    errs["int"] = float((ts.block_cost(l.to(dev), r.to(dev), D, sc).cpu() - oracle.block_cost(l, r, D, sc)).abs().max())
    errs["cat"] = float((ts.cat_fms(l.to(dev), r.to(dev), d.to(dev)).cpu() - ocv.cat_fms(l, r, d)).abs().max())
    errs["dif"] = float((ts.dif_fms(l.to(dev), r.to(dev), d.to(dev)).cpu() - ocv.dif_fms(l, r, d)).abs().max())
    tol = 2e-3 if max(errs.values()) < 1e9 else 0
    flag = any(v > 2e-3 for v in errs.values())
    bad += flag
    if flag or it % 10 == 0:
        print(("BAD " if flag else "ok  ") + str((B, C, H, W, D, sc)), {k: "%.1e" % v for k, v in errs.items()}, flush=True)
print("cases", N, "bad", bad)
