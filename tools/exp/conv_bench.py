import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from temporalstereo_amd.layers import Conv3d
from temporalstereo_amd.aggregation import native
dev = torch.device("cuda:0")
def t_op(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(64, 32, 1, 272, 480, 1, 1), (304, 8, 5, 136, 240, 1, 1), (352, 32, 12, 34, 60, 1, 1), (304, 16, 5, 68, 120, 1, 1),
          (128, 32, 1, 136, 240, 1, 1), (256, 64, 1, 34, 60, 1, 1), (32, 64, 12, 34, 60, 2, 1), (8, 8, 5, 136, 240, 1, 1), (3, 32, 1, 544, 960, 2, 1)]
for (cin, cout, D, H, W, s, dl) in shapes:
    m = Conv3d(cin, cout, (1, 3, 3), (1, s, s), (0, dl, dl), (1, dl, dl), bias=False, norm=('BN3d', cout), activation='SiLU').to(dev).eval()
    f = native.fold_wrapper(m, "hw")
    x = torch.randn(1, cin, D, H, W, device=dev)
    us = t_op(lambda: native.conv_hw(x, f, s, dl))
    with torch.no_grad():
        us_t = t_op(lambda: m(x))
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    gmac = cin * 9 * cout * D * Ho * Wo / 1e9
    print("conv_hw %4d->%-3d D=%-2d %3dx%-3d s%d d%d: %8.1f us  %6.2f TFLOP/s   (torch/MIOpen %8.1f us)" % (cin, cout, D, H, W, s, dl, us, 2 * gmac / us * 1e-3 * 1e3 / 1e3 * 1e0 if False else 2 * gmac * 1e9 / (us * 1e-6) / 1e12, us_t), flush=True)
