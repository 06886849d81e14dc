"""Every stride-1 (1,3,3) convolution of a config-2 pass that the x6 kernel can take, timed alone three ways through the C ABI
(200 back-to-back launches): f32 MFMA kernel (with its own split-K where it splits), x6 unsplit, x6 split-K."""
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


def spy(x, f, stride=1, dilation=1, transposed=False, **kw):
    B, Cin, D, H, W = x.shape
    if kw.get("second") is not None:
        B = 2
    if _lib.lib().ts_conv3d_hw_x6_supported(Cin, f.cout, W, stride, dilation, int(transposed)):
        key = (B, Cin, f.cout, D, H, W, dilation)
        shapes[key] = shapes.get(key, 0) + 1
    return orig(x, f, stride, dilation, transposed, **kw)


N.conv_hw = spy
eng = InferenceEngine(net, backend="native", replay="eager", inputs="bind", pipeline=1)
with torch.no_grad():
    eng(*inputs, {})
torch.cuda.synchronize()
N.conv_hw = orig
L = _lib.lib()
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


tot = [0.0, 0.0, 0.0]
for (B, Cin, Cout, D, H, W, dil), calls in shapes.items():
    x = torch.randn(B, Cin, D, H, W, device=dev)
    w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
    f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
    out = torch.empty(B, Cout, D, H, W, device=dev)
    w6 = N.x6_weights(f)
    wsb = int(L.ts_conv3d_hw_workspace_bytes(B, Cin, Cout, D, H, W, 1, 0))
    ws = torch.empty(max(wsb, 16), device=dev, dtype=torch.uint8)
    wsb6 = int(L.ts_conv3d_hw_x6_workspace_bytes(B, Cin, Cout, D, H, W))
    ws6 = torch.empty(max(wsb6, 16), device=dev, dtype=torch.uint8)
    a = (x.stride(0), x.stride(1), out.stride(0), out.stride(1))
    t32 = timed(lambda: L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, 1, dil, 0,
                                           N.ACT_SILU, 0.0, *a, None, 0, _lib.ptr(ws) if wsb else None, wsb, st))
    t6 = timed(lambda: L.ts_conv3d_hw_x6_fwd(_lib.ptr(x), _lib.ptr(w6), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, dil,
                                             N.ACT_SILU, 0.0, *a, None, 0, None, 0, st))
    t6s = timed(lambda: L.ts_conv3d_hw_x6_fwd(_lib.ptr(x), _lib.ptr(w6), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W, dil,
                                              N.ACT_SILU, 0.0, *a, None, 0, _lib.ptr(ws6), wsb6, st)) if wsb6 else t6
    grid = ((H + 7) // 8) * ((W + 31) // 32) * D * B * ((Cout + 31) // 32)
    ks = wsb6 // (B * Cout * D * H * W * 4) if wsb6 else 1
    for i, t in enumerate((t32, t6, t6s)):
        tot[i] += t * calls
    print("x%d B=%d %3d->%-3d D=%-2d %3dx%-3d dil %d  grid %4d  f32%s %6.1f us   x6 %6.1f us   x6 split-%d %6.1f us" % (
        calls, B, Cin, Cout, D, H, W, dil, grid, "(split)" if wsb else "       ", t32, t6, ks, t6s), flush=True)
print("per pass: f32 %.0f us, x6 unsplit %.0f us, x6 with split-K %.0f us" % tuple(tot))
