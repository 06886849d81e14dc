import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from temporalstereo_amd import functional as TF
from temporalstereo_amd.aggregation.blocks import PyramidFusion, DepthwiseConv3D
from oracle import aggregation as oagg
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return float((a.detach().double().cpu() - b.detach().double().cpu()).norm() / b.detach().double().cpu().norm())
for C, shape, const in [(16, (2, 16, 7, 16, 24), False), (16, (2, 16, 7, 16, 24), True), (32, (2, 32, 6, 8, 12), True)]:
    m = PyramidFusion(C).to(dev).train()
    for p in m.parameters():
        p.data.normal_(0, 0.3)
    x = torch.randn(*shape, device=dev)
    if const:
        x[:, :, :2] = torch.randn(1, C, 1, 1, 1, device=dev)
    g = torch.randn_like(x)
    xh = x.clone().requires_grad_(True)
    y = m(xh); y.backward(g)
    sd = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
    xr = x.double().cpu().requires_grad_(True)
    yr = oagg.pyramid_fusion(oagg.StateView(sd, "", True), xr); yr.backward(g.double().cpu())
    print(C, shape, const, "fwd %.3g  d/dx %.3g" % (rel(y, yr), rel(xh.grad, xr.grad)))
    # pieces: conv_fuse alone on a random 4C input
    z = torch.randn(shape[0], 4 * C, *shape[2:], device=dev)
    zh = z.clone().requires_grad_(True)
    yy = m.conv_fuse(zh); yy.backward(g)
    zr = z.double().cpu().requires_grad_(True)
    yyr = oagg.sep_conv3d(oagg.StateView(sd, "conv_fuse.", True), zr, act=None); yyr.backward(g.double().cpu())
    print("    conv_fuse alone: fwd %.3g  d/dz %.3g ; per quarter of the channels:" % (rel(yy, yyr), rel(zh.grad, zr.grad)),
          ["%.2g" % rel(zh.grad[:, i * C:(i + 1) * C], zr.grad[:, i * C:(i + 1) * C]) for i in range(4)])
    # the cat in front of it
    xh2 = x.clone().requires_grad_(True)
    a, mx = TF.pool5_avgmax(xh2)
    cat = torch.cat([xh2, m.conv_5x5(xh2), a, mx], 1)
    gz = torch.randn_like(cat)
    cat.backward(gz)
    xr2 = x.double().cpu().requires_grad_(True)
    catr = torch.cat([xr2, oagg.conv3d(oagg.StateView(sd, "conv_5x5.", True), xr2, 1, (2, 0, 0)), F.avg_pool3d(xr2, 5, 1, 2), F.max_pool3d(xr2, 5, 1, 2)], 1)
    catr.backward(gz.double().cpu())
    print("    cat of the four branches: fwd %.3g  d/dx %.3g" % (rel(cat, catr), rel(xh2.grad, xr2.grad)))
