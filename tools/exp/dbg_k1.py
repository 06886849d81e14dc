import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, oracle
import temporalstereo_amd as ts
from temporalstereo_amd import functional as TF
dev = torch.device("cuda:0")
for (B, C, H, W, D) in [(2, 128, 60, 80, 8), (2, 128, 60, 80, 7), (2, 128, 60, 80, 6), (1, 128, 68, 120, 8), (1, 16, 12, 80, 8)]:
    l = synth.normal(1, "l", (B, C, H, W)); r = synth.normal(2, "r", (B, C, H, W)); d = synth.uniform(3, "d", (B, D, H, W), 0.0, 30.0)
    t = lambda a: torch.from_numpy(a)
    exp = oracle.block_cost(t(l), t(r), t(d), 3)
    got = ts.block_cost(t(l).to(dev), t(r).to(dev), t(d).to(dev), 3).cpu()
    w = TF.block_cost_warped(t(l).to(dev), t(r).to(dev), t(d).to(dev), 3).cpu()
    e1 = (got - exp).abs(); e2 = (w - exp[:, C:]).abs()
    print((B, C, H, W, D), "full max err %.3e  warped max err %.3e" % (float(e1.max()), float(e2.max())),
          " per-channel-block max:", ["%.1e" % float(e1[:, a:b].max()) for a, b in ((0, C), (C, 2 * C), (2 * C, 2 * C + C // 8), (2 * C + C // 8, 2 * C + C // 4), (2 * C + C // 4, 2 * C + 3 * C // 8))])
B, C, H, W, D = 2, 128, 60, 80, 8
l = synth.normal(1, "l", (B, C, H, W)); r = synth.normal(2, "r", (B, C, H, W)); d = synth.uniform(3, "d", (B, D, H, W), 0.0, 30.0)
exp = oracle.block_cost(t(l), t(r), t(d), 3)[:, C:]
w = TF.block_cost_warped(t(l).to(dev), t(r).to(dev), t(d).to(dev), 3).cpu()
e = (w - exp).abs()
bad = (e > 1e-3)
print("bad fraction", float(bad.double().mean()))
idx = bad.nonzero()
print("bad channels", sorted(set(idx[:, 1].tolist()))[:40], "...", len(set(idx[:, 1].tolist())))
print("bad d", sorted(set(idx[:, 2].tolist())), "bad b", sorted(set(idx[:, 0].tolist())))
print("bad rows", sorted(set(idx[:, 3].tolist()))[:70])
print("bad cols", sorted(set(idx[:, 4].tolist()))[:90])
print("first few", idx[:10].tolist())
