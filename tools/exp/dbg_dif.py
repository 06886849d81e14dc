import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import oracle.cost_volume as ocv
import temporalstereo_amd as ts
dev = torch.device("cuda:0")
rng = np.random.RandomState(1)
for it in range(80):
    B = int(rng.choice([1, 2, 3])); C = int(rng.choice([8, 16, 32, 64, 128])); H = int(rng.randint(4, 70)); W = int(rng.choice([rng.randint(4, 130), 4 * rng.randint(1, 80)]))
    D = int(rng.randint(2, 13)); sc = int(rng.choice([1, 2, 3]))
    if (B, C, H, W, D) != (2, 128, 31, 232, 8): continue
    l = torch.from_numpy(synth.normal(it, "l", (B, C, H, W))); r = torch.from_numpy(synth.normal(it, "r", (B, C, H, W)))
    d = torch.from_numpy(synth.uniform(it, "d", (B, D, H, W), -3.0, W * 0.6))
    exp = ocv.dif_fms(l, r, d); got = ts.dif_fms(l.to(dev), r.to(dev), d.to(dev)).cpu()
    tgt = ocv.warp_candidates(r, d)
    bad = ((got - exp).abs() > 1e-3).nonzero()
    print("bad elements", len(bad), "of", exp.numel())
    for (b, c, dd, y, x) in bad[:8].tolist():
        print((b, c, dd, y, x), "got %.4f exp %.4f  oracle warped value %.3e  disp %.6f  x-disp %.6f" % (float(got[b, c, dd, y, x]), float(exp[b, c, dd, y, x]), float(tgt[b, c, dd, y, x]), float(d[b, dd, y, x]), x - float(d[b, dd, y, x])))
