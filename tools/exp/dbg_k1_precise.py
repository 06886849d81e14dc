import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
import parity_tools as PT
from temporalstereo_amd import functional as TF
import oracle
dev = torch.device("cuda:0")
name = list(PT.CONFIGS)[0]
case = PT.Case(PT.CONFIGS[name], synth.SEED0 + 11, dev)
out, tr, _ = case.oracle_frame(0, {})
L, R, ds = tr["precise_left"], tr["precise_right"], tr["precise_ds"]
ours = TF.block_cost(L.to(dev), R.to(dev), ds.to(dev).contiguous(), 3).cpu().double()
ref = tr["precise_raw"].double()
ref64 = oracle.block_cost(L.double(), R.double(), ds.double(), 3)
C = L.shape[1]
for nm, lo, hi in (("left", 0, C), ("warped", C, 2 * C), ("corr s0", 2 * C, 2 * C + C // 8), ("corr s1", 2 * C + C // 8, 2 * C + 2 * (C // 8)), ("corr s2", 2 * C + 2 * (C // 8), 2 * C + 3 * (C // 8))):
    d = (ours[:, lo:hi] - ref[:, lo:hi]).abs(); d64 = (ours[:, lo:hi] - ref64[:, lo:hi]).abs(); r64 = (ref[:, lo:hi] - ref64[:, lo:hi]).abs()
    i = int(d.argmax()); idx = [int(v) for v in torch.unravel_index(torch.tensor(i), d.shape)]
    print("%-8s ours-o32 max %.3g mean %.3g | ours-o64 max %.3g mean %.3g | o32-o64 max %.3g mean %.3g | refmax %.3g | at %s ours %.6f ref %.6f ref64 %.6f" % (
        nm, float(d.max()), float(d.mean()), float(d64.max()), float(d64.mean()), float(r64.max()), float(r64.mean()), float(ref[:, lo:hi].abs().max()), idx,
        float(ours[:, lo:hi].flatten()[i]), float(ref[:, lo:hi].flatten()[i]), float(ref64[:, lo:hi].flatten()[i])))
    b, c, dd, y, x = idx
    print("     disp at max:", float(ds[b, dd, y, x]), "x", x, "src", x - float(ds[b, dd, y, x]))
