"""Every (k,1,1) convolution of a config-2 pass timed alone through the C ABI (200 back-to-back launches) at K-chunk caps 8 / 16 / 32
(ts_conv_set_chunk_cap), and the f32 (1,3,3) layers that the x6 kernel does not take, the same way."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd import _lib  # noqa: E402
from temporalstereo_amd.aggregation import native as N  # noqa: E402
from temporalstereo_amd.aggregation.engine import InferenceEngine  # noqa: E402

dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.load_trained(bench.build_model(dev, seed)).eval()
inputs, _ = bench.make_planted_inputs(dev, seed, 1)
dshapes, hshapes = collections.OrderedDict(), collections.OrderedDict()
od, oh = N.conv_d, N.conv_hw
L = _lib.lib()


def spy_d(x, f, k, stride=1, dilation=1, padding=0, transposed=False, **kw):
    key = tuple(x.shape) + (f.cout, k, stride, dilation, padding, bool(transposed))
    dshapes[key] = dshapes.get(key, 0) + 1
    return od(x, f, k, stride, dilation, padding, transposed, **kw)


def spy_h(x, f, stride=1, dilation=1, transposed=False, **kw):
    B, Cin, D, H, W = x.shape
    if kw.get("second") is not None:
        B = 2
    if not L.ts_conv3d_hw_x6_supported(Cin, f.cout, W, stride, dilation, int(transposed)):
        key = (B, Cin, D, H, W, f.cout, stride, dilation, bool(transposed))
        hshapes[key] = hshapes.get(key, 0) + 1
    return oh(x, f, stride, dilation, transposed, **kw)


N.conv_d, N.conv_hw = spy_d, spy_h
for cls in (N.SepConv, N.Heads, N._MergingLevel):
    pass
eng = InferenceEngine(net, backend="native", replay="eager", inputs="bind", pipeline=1)
with torch.no_grad():
    eng(*inputs, {})
torch.cuda.synchronize()
N.conv_d, N.conv_hw = od, oh
st = N._stream()


def timed(fn, n=200):
    for _ in range(10):
        _lib.check(fn(), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {8: 0.0, 16: 0.0, 32: 0.0}
for (B, Cin, D, H, W, Cout, k, stride, dil, pad, tr), calls in dshapes.items():
    x = torch.randn(B, Cin, D, H, W, device=dev)
    wshape = (Cin, Cout, k, 1, 1) if tr else (Cout, Cin, k, 1, 1)
    w = torch.randn(*wshape, device=dev) / (k * Cin) ** 0.5
    f = N.Folded(w, None, None, N.ACT_SILU, tr, "d")
    Do = 2 * D if tr else (D + 2 * pad - dil * (k - 1) - 1) // stride + 1
    out = torch.empty(B, Cout, Do, H, W, device=dev)
    a = (x.stride(0), x.stride(1), out.stride(0), out.stride(1))
    t = {}
    for cap in (8, 16, 32):
        L.ts_conv_set_chunk_cap(cap)
        t[cap] = timed(lambda: L.ts_conv3d_d_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, k, stride, dil,
                                               pad, int(tr), N.ACT_SILU, 0.0, *a, st))
        tot[cap] += t[cap] * calls
    wgs = ((H * W + 255) // 256) * Do * B * ((Cout + 15) // 16)
    print("d  x%d B=%d %3d->%-3d D=%-2d %3dx%-3d k%d s%d%s  ~%4d wg   cap8 %5.1f  cap16 %5.1f  cap32 %5.1f us" % (
        calls, B, Cin, Cout, D, H, W, k, stride, " T" if tr else "  ", wgs, t[8], t[16], t[32]), flush=True)
print("conv_d per pass: cap8 %.0f us, cap16 %.0f us, cap32 %.0f us" % (tot[8], tot[16], tot[32]))
tot = {8: 0.0, 16: 0.0, 32: 0.0}
for (B, Cin, D, H, W, Cout, stride, dil, tr), calls in hshapes.items():
    x = torch.randn(B, Cin, D, H, W, device=dev)
    wshape = (Cin, Cout, 1, 3, 3) if tr else (Cout, Cin, 1, 3, 3)
    w = torch.randn(*wshape, device=dev) / (9 * Cin) ** 0.5
    f = N.Folded(w, None, None, N.ACT_SILU, tr, "hw")
    Ho, Wo = (2 * H, 2 * W) if tr else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    out = torch.empty(B, Cout, D, Ho, Wo, device=dev)
    wsb = int(L.ts_conv3d_hw_workspace_bytes(B, Cin, Cout, D, H, W, stride, int(tr)))
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    a = (x.stride(0), x.stride(1), out.stride(0), out.stride(1))
    t = {}
    for cap in (8, 16, 32):
        L.ts_conv_set_chunk_cap(cap)
        t[cap] = timed(lambda: L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, stride, dil,
                                                int(tr), N.ACT_SILU, 0.0, *a, None, 0, _lib.ptr(ws) if wsb else None, wsb, st))
        tot[cap] += t[cap] * calls
    print("hw x%d B=%d %3d->%-3d D=%-2d %3dx%-3d s%d d%d%s %s  cap8 %5.1f  cap16 %5.1f  cap32 %5.1f us" % (
        calls, B, Cin, Cout, D, H, W, stride, dil, " T" if tr else "  ", "split" if wsb else "     ", t[8], t[16], t[32]), flush=True)
print("f32 conv_hw per pass: cap8 %.0f us, cap16 %.0f us, cap32 %.0f us" % (tot[8], tot[16], tot[32]))
L.ts_conv_set_chunk_cap(32)
