"""Every (1,3,3) convolution of a config-2 pass for which the f32 kernel's split-K applies (ts_conv3d_hw_workspace_bytes > 0), timed alone
through the C ABI with and without the workspace (200 back-to-back launches): what the finishing launch costs against what the slices save."""
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
shapes = collections.OrderedDict()
orig = N.conv_hw
L = _lib.lib()


def spy(x, f, stride=1, dilation=1, transposed=False, **kw):
    B, Cin, D, H, W = x.shape
    if kw.get("second") is not None:
        B = 2
    if not transposed and int(L.ts_conv3d_hw_workspace_bytes(B, Cin, f.cout, D, H, W, stride, 0)):
        x6 = bool(L.ts_conv3d_hw_x6_supported(Cin, f.cout, W, stride, dilation, 0))
        key = (B, Cin, f.cout, D, H, W, stride, dilation, x6)
        shapes[key] = shapes.get(key, 0) + 1
    return orig(x, f, stride, dilation, transposed, **kw)


N.conv_hw = spy
eng = InferenceEngine(net, backend="native", replay="eager", inputs="bind", pipeline=1)
with torch.no_grad():
    eng(*inputs, {})
torch.cuda.synchronize()
N.conv_hw = orig
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


for (B, Cin, Cout, D, H, W, stride, dil, x6), calls in shapes.items():
    x = torch.randn(B, Cin, D, H, W, device=dev)
    w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
    f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty(B, Cout, D, Ho, Wo, device=dev)
    wsb = int(L.ts_conv3d_hw_workspace_bytes(B, Cin, Cout, D, H, W, stride, 0))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    a = (x.stride(0), x.stride(1), out.stride(0), out.stride(1))
    call = lambda buf, nb: L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, stride, dil, 0,
                                             N.ACT_SILU, 0.0, *a, None, 0, buf, nb, st)
    ts = timed(lambda: call(_lib.ptr(ws), wsb))
    tu = timed(lambda: call(None, 0))
    ks = wsb // (B * Cout * D * Ho * Wo * 4)
    print("x%d B=%d %3d->%-3d D=%-2d %3dx%-3d stride %d dil %d %s  split-%d %6.1f us   unsplit %6.1f us" % (
        calls, B, Cin, Cout, D, H, W, stride, dil, "(x6 takes it)" if x6 else "             ", ks, ts, tu), flush=True)
