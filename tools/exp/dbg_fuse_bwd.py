"""Which contribution to d PyramidFusion / d input is off?  (round 3: 5-7 % in the stage-wise backward test)"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from temporalstereo_amd import functional as TF
from temporalstereo_amd.layers import Conv3d
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for shape in [(2, 16, 7, 16, 24), (2, 32, 6, 8, 12), (1, 8, 9, 20, 33)]:
    x = torch.randn(*shape, device=dev)
    ga, gm = torch.randn_like(x), torch.randn_like(x)
    xh = x.clone().requires_grad_(True)
    a, m = TF.pool5_avgmax(xh)
    torch.autograd.backward([a], [ga]); g_avg_h = xh.grad.clone(); xh.grad = None
    a, m = TF.pool5_avgmax(xh)
    torch.autograd.backward([m], [gm]); g_max_h = xh.grad.clone(); xh.grad = None
    a, m = TF.pool5_avgmax(xh)
    torch.autograd.backward([a, m], [ga, gm]); g_both_h = xh.grad.clone()
    xr = x.double().cpu().requires_grad_(True)
    ar = F.avg_pool3d(xr, 5, 1, 2); ar.backward(ga.double().cpu()); g_avg_r = xr.grad.clone(); xr.grad = None
    mr = F.max_pool3d(xr, 5, 1, 2); mr.backward(gm.double().cpu()); g_max_r = xr.grad.clone()
    print(shape, "avg only %.3g  max only %.3g  both %.3g" % (rel(g_avg_h, g_avg_r), rel(g_max_h, g_max_r), rel(g_both_h, g_avg_r + g_max_r)))
    C = shape[1]
    conv = Conv3d(C, C, (5, 1, 1), 1, (2, 0, 0), bias=False, norm=('BN3d', C), activation='SiLU').to(dev).train()
    xh = x.clone().requires_grad_(True)
    y = conv(xh); y.backward(ga)
    w = conv.weight.detach().double().cpu(); bn = conv.norm
    xr = x.double().cpu().requires_grad_(True)
    yr = F.silu(F.batch_norm(F.conv3d(xr, w, None, 1, (2, 0, 0)), None, None, bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), True, 0.0, 1e-5))
    yr.backward(ga.double().cpu())
    print("   conv(5,1,1)+BN+SiLU fwd %.3g  d/dx %.3g  d/dw %.3g" % (rel(y, yr), rel(xh.grad, xr.grad), rel(conv.weight.grad, torch.autograd.grad(F.silu(F.batch_norm(F.conv3d(xr, w.requires_grad_(True), None, 1, (2, 0, 0)), None, None, bn.weight.detach().double().cpu(), bn.bias.detach().double().cpu(), True, 0.0, 1e-5)), w, ga.double().cpu())[0])))
