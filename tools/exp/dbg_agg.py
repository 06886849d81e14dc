import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import temporalstereo_amd as ts
from helpers import load, t, dims_from_golden, synth_state, aggregator_inputs, epe
from test_aggregator_gpu import _build
g = load("agg_config1_256x512"); dev = torch.device("cuda:0")
dims = dims_from_golden(g)
net = _build(dims, int(g["seed"]), dev, golden=g)
lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
with torch.no_grad():
    disps, costs, samples, offs, ranges, info = net(lf, rf, il, ir, prev)
for nm, a, b, sc in (("full_sub4", disps[0][:, :, ::4, ::4], g["disp_full_sub4"], 1), ("precise", disps[1], g["disp_precise"], 4),
                     ("fine_up", disps[2], g["disp_fine_up"], 4), ("coarse_up", disps[3], g["disp_coarse_up"], 8)):
    d = (a.cpu() - t(b)).abs() * sc
    print("%-10s epe %.3e  max %.3e  frac>1e-2 %.2e  frac>1 %.2e" % (nm, d.mean(), d.max(), (d > 1e-2).float().mean(), (d > 1).float().mean()))
d = (costs[2].cpu() - t(g["cost_coarse"])).abs(); print("cost_coarse epe %.3e max %.3e" % (d.mean(), d.max()))
d = (samples[1].cpu() - t(g["samp_fine"])).abs(); print("samp_fine epe %.3e max %.3e" % (d.mean(), d.max()))
print("mean full", float(disps[0].double().mean()), float(g["disp_full_mean"]))
