"""Weight-gradient kernel (ts_conv3d_hw_bwd_weight / ts_conv3d_d_bwd_weight) timed alone on the training step's largest layers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from temporalstereo_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [  # B, Cin, Cout, D, H, W, stride
    (1, 32, 32, 1, 272, 480, 1), (1, 64, 32, 1, 272, 480, 1), (1, 128, 32, 1, 272, 480, 1), (1, 64, 64, 1, 136, 240, 1),
    (1, 176, 8, 5, 136, 240, 1), (1, 304, 8, 5, 136, 240, 1), (1, 304, 16, 5, 68, 120, 1), (1, 32, 64, 1, 272, 480, 2), (1, 352, 32, 12, 34, 60, 1), (1, 32, 32, 12, 34, 60, 1), (1, 16, 16, 5, 68, 120, 1),
]
for B, Cin, Cout, D, H, W, s in SHAPES:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, Cin, D, H, W, device=dev)
    dy = torch.randn(B, Cout, D, Ho, Wo, device=dev)
    dw = torch.empty(Cout, Cin, 1, 3, 3, device=dev)
    nws = int(L.ts_conv3d_bwd_weight_workspace_bytes(Cin, Cout, 9))
    ws = torch.empty(nws, device=dev, dtype=torch.uint8)
    fn = lambda: L.ts_conv3d_hw_bwd_weight(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), B, Cin, Cout, D, H, W, s, 1, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1),
                                          _lib.ptr(ws), nws, st)
    for _ in range(5):
        _lib.check(fn(), "wgrad")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * B * Cin * Cout * 9 * D * Ho * Wo
    print("wgrad B=%d %3d->%-3d D=%-2d %3dx%-3d s%d  %7.1f us  %5.1f TFLOP/s  (dw checksum %.4e)" % (B, Cin, Cout, D, H, W, s, t, fl / t / 1e6, float(dw.double().abs().sum())), flush=True)
