import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from temporalstereo_amd import functional as TF
from temporalstereo_amd.aggregation.blocks import PyramidFusion
from oracle import aggregation as oagg
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return float((a.detach().double().cpu() - b.detach().double().cpu()).norm() / b.detach().double().cpu().norm())
C, shape = 16, (2, 16, 7, 16, 24)
m = PyramidFusion(C).to(dev).train()
x = torch.randn(*shape, device=dev)
sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
def branches(xh, which, hip):
    out = []
    if "i" in which: out.append(xh)
    if "c" in which: out.append(m.conv_5x5(xh) if hip else oagg.conv3d(oagg.StateView(sd, "conv_5x5.", True), xh, 1, (2, 0, 0)))
    if "a" in which or "m" in which:
        a, mx = TF.pool5_avgmax(xh) if hip else (F.avg_pool3d(xh, 5, 1, 2), F.max_pool3d(xh, 5, 1, 2))
        if "a" in which: out.append(a)
        if "m" in which: out.append(mx)
    return out
for which in ("ic", "ia", "im", "ca", "cm", "am", "iam", "icam", "cam"):
    for mode in ("cat", "sum"):
        xh = x.clone().requires_grad_(True)
        bs = branches(xh, which, True)
        gz = [torch.randn_like(b) for b in bs]
        if mode == "cat":
            torch.cat(bs, 1).backward(torch.cat(gz, 1))
        else:
            torch.autograd.backward(bs, gz)
        xr = x.double().cpu().requires_grad_(True)
        torch.autograd.backward(branches(xr, which, False), [g.double().cpu() for g in gz])
        print(which, mode, "%.3g" % rel(xh.grad, xr.grad))
