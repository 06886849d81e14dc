import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth, oracle
from temporalstereo_amd import functional as TF
import temporalstereo_amd as ts
dev = torch.device("cuda:0")
B, C, H, W, D = 2, 128, 60, 80, 8
l = torch.from_numpy(synth.normal(1, "l", (B, C, H, W))); r = torch.from_numpy(synth.normal(2, "r", (B, C, H, W))); d = torch.from_numpy(synth.uniform(3, "d", (B, D, H, W), 0.0, 30.0))
exp = oracle.block_cost(l, r, d, 3)[:, C:]
def run(l, r, d, exp, tag):
    for i in range(3):
        w = TF.block_cost_warped(l.to(dev), r.to(dev), d.to(dev), 3).cpu()
        bad = ((w - exp).abs() > 1e-3)
        idx = bad.nonzero()
        print(tag, "run", i, "bad", int(bad.sum()), "b", sorted(set(idx[:, 0].tolist())), "ch%8", sorted(set((idx[:, 1] % 8).tolist())) if len(idx) else [], "nch", len(set(idx[:, 1].tolist())) if len(idx) else 0)
    return w
w = run(l, r, d, exp, "B=2      ")
run(l.flip(0).contiguous(), r.flip(0).contiguous(), d.flip(0).contiguous(), exp.flip(0), "B=2 flip ")
run(l[1:], r[1:], d[1:], exp[1:], "B=1 (b1) ")
l3 = torch.cat([l, l[:1]]); r3 = torch.cat([r, r[:1]]); d3 = torch.cat([d, d[:1]]); e3 = torch.cat([exp, exp[:1]])
run(l3, r3, d3, e3, "B=3      ")
# values
bad = ((w - exp).abs() > 1e-3).nonzero()
for (b, c, dd, y, x) in bad[:6].tolist():
    print((b, c, dd, y, x), "got %.5f exp %.5f | neighbours got" % (float(w[b, c, dd, y, x]), float(exp[b, c, dd, y, x])), w[b, c, dd, y, x:x+4].tolist(), "exp", exp[b, c, dd, y, x:x+4].tolist())
    # is the wrong value found elsewhere in exp?
    m = (exp - w[b, c, dd, y, x]).abs() < 1e-6
    print("   value appears in expected at", m.nonzero()[:4].tolist())
