"""Random-shape sweep of the INFERENCE forms of the convolutions (aggregation/native.py over the C ABI): every kernel family forced
where its `supported` query allows it -- f32-input MFMA (1,3,3) in all stride / dilation / transposed forms, bf16-split x6 and x6s
(stride 2, transposed (1,3,3), the 4x4 deconvolution), the (k,1,1) family, the plain 4x4 deconvolution -- with a folded scale / shift,
an activation and (where the entry takes one) a per-plane addend, against float64 torch.  Companion of tests/fuzz_ops.py."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from temporalstereo_amd import _lib
from temporalstereo_amd.aggregation import native as N

dev = torch.device("cuda:0")
FAILS = []


def rnd(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def finish(y, scale, shift, act):
    y = y * scale.view(1, -1, *([1] * (y.dim() - 2))) + shift.view(1, -1, *([1] * (y.dim() - 2)))
    return y if act == N.ACT_NONE else (F.silu(y) if act == N.ACT_SILU else F.relu(y))


def folded(w64, cout, scale, shift, act, transposed, kind):
    f = N.Folded(w64.float().to(dev), None, None, act, transposed, kind)
    f.scale[:cout] = scale.float().to(dev)
    f.shift[:cout] = shift.float().to(dev)
    return f


def compare(what, desc, got, want, k_terms):
    got = got.detach().cpu().double()
    if got.shape != want.shape:
        FAILS.append((what, desc, "shape %s vs %s" % (tuple(got.shape), tuple(want.shape))))
        return
    scale = float(want.abs().max()) + 1e-6
    err = float((got - want).abs().max()) if got.numel() else 0.0
    if not (err <= 3e-6 * scale * max(1.0, k_terms ** 0.5) + 1e-6) or not torch.isfinite(got).all():
        FAILS.append((what, desc, "max err %.3g at scale %.3g" % (err, scale)))


def fuzz_hw(r, g):
    L = _lib.lib()
    form = r.choice(["s1", "s1", "d2", "s2", "T"])
    B, Cin, Cout = r.randint(1, 3), r.randint(1, 72), r.randint(1, 64)
    D, H, W = r.randint(1, 7), r.randint(1, 40), 4 * r.randint(1, 20) if r.random() < 0.7 else r.randint(1, 70)
    stride, dil, tr = (2, 1, False) if form == "s2" else ((1, 2, False) if form == "d2" else ((2, 1, True) if form == "T" else (1, 1, False)))
    x = rnd(g, B, Cin, D, H, W)
    w = rnd(g, *((Cin, Cout) if tr else (Cout, Cin)), 1, 3, 3, scale=1.0 / (9 * Cin) ** 0.5)
    scale, shift = rnd(g, Cout).abs() + 0.5, rnd(g, Cout)
    act = r.choice([N.ACT_NONE, N.ACT_SILU, N.ACT_RELU])
    if tr:
        raw = F.conv_transpose3d(x, w, None, (1, 2, 2), (0, 1, 1), (0, 1, 1))
    else:
        raw = F.conv3d(x, w, None, (1, stride, stride), (0, dil, dil), (1, dil, dil))
    f = folded(w, Cout, scale, shift, act, tr, "hw")
    xg = x.float().to(dev)
    desc = "%s B%d %d->%d %dx%dx%d act%d" % (form, B, Cin, Cout, D, H, W, act)
    Ho, Wo = raw.shape[-2:]
    use_add = (not tr) and stride == 1 and r.random() < 0.4
    add = rnd(g, B, Cout, 1, Ho, Wo) if use_add else None
    want = finish(raw + (add if use_add else 0.0), scale, shift, act)
    addg = add.float().to(dev).contiguous() if use_add else None
    ib, ic = N._strides5(xg)
    # 1. the f32-input MFMA kernel
    out = torch.empty((B, Cout, D, Ho, Wo), device=dev)
    ob, oc = N._strides5(out)
    wsb = int(L.ts_conv3d_hw_workspace_bytes(B, Cin, Cout, D, H, W, stride, int(tr)))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8) if wsb else None
    try:
        _lib.check(L.ts_conv3d_hw_fwd(_lib.ptr(xg), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, D, H, W,
                                      stride, dil, int(tr), act, 0.0, ib, ic, ob, oc, _lib.ptr(addg), addg.stride(0) if use_add else 0,
                                      _lib.ptr(ws), wsb, N._stream()), "ts_conv3d_hw_fwd")
        compare("hw_f32", desc + (" +addend" if use_add else ""), out, want, 9 * Cin)
    except Exception as e:
        FAILS.append(("hw_f32", desc, "raised %s" % str(e)[:160]))
    # 2. x6 (stride 1)
    if not tr and stride == 1 and L.ts_conv3d_hw_x6_supported(Cin, Cout, W, 1, dil, 0):
        out = torch.full((B, Cout, D, Ho, Wo), float("nan"), device=dev)
        wsb6 = int(L.ts_conv3d_hw_x6_workspace_bytes(B, Cin, Cout, D, H, W))
        ws6 = torch.empty(wsb6, device=dev, dtype=torch.uint8) if wsb6 else None
        try:
            _lib.check(L.ts_conv3d_hw_x6_fwd(_lib.ptr(xg), _lib.ptr(N.x6_weights(f)), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin,
                                             Cout, D, H, W, dil, act, 0.0, ib, ic, ob, oc, _lib.ptr(addg), addg.stride(0) if use_add else 0,
                                             _lib.ptr(ws6), wsb6, N._stream()), "ts_conv3d_hw_x6_fwd")
            compare("hw_x6", desc + (" +addend" if use_add else "") + (" splitK" if wsb6 else ""), out, want, 9 * Cin)
        except Exception as e:
            FAILS.append(("hw_x6", desc, "raised %s" % str(e)[:160]))
    # 3. x6s (stride 2 / transposed)
    if stride == 2 and not use_add:
        mode = N.X6S_T3 if tr else N.X6S_S2
        if L.ts_conv3d_hw_x6s_supported(Cin, Cout, H, W, mode):
            out = torch.full((B, Cout, D, Ho, Wo), float("nan"), device=dev)
            try:
                _lib.check(L.ts_conv3d_hw_x6s_fwd(_lib.ptr(xg), _lib.ptr(N.x6s_weights(f, mode)), _lib.ptr(f.scale), _lib.ptr(f.shift),
                                                  _lib.ptr(out), B, Cin, Cout, D, H, W, mode, act, 0.0, ib, ic, ob, oc, N._stream()),
                           "ts_conv3d_hw_x6s_fwd")
                compare("hw_x6s", desc, out, want, 9 * Cin)
            except Exception as e:
                FAILS.append(("hw_x6s", desc, "raised %s" % str(e)[:160]))


def fuzz_d(r, g):
    form = r.choice(["k1", "k3", "k3d2", "k5", "k3s2", "T"])
    B, Cin, Cout = r.randint(1, 3), r.randint(1, 72), r.randint(1, 64)
    D, H, W = r.randint(1, 14), r.randint(1, 30), r.randint(1, 60)
    k, stride, dil, pad, tr = {"k1": (1, 1, 1, 0, False), "k3": (3, 1, 1, 1, False), "k3d2": (3, 1, 2, 2, False), "k5": (5, 1, 1, 2, False),
                               "k3s2": (3, 2, 1, 1, False), "T": (3, 2, 1, 1, True)}[form]
    x = rnd(g, B, Cin, D, H, W)
    w = rnd(g, *((Cin, Cout) if tr else (Cout, Cin)), k, 1, 1, scale=1.0 / (k * Cin) ** 0.5)
    scale, shift = rnd(g, Cout).abs() + 0.5, rnd(g, Cout)
    act = r.choice([N.ACT_NONE, N.ACT_SILU, N.ACT_RELU])
    raw = F.conv_transpose3d(x, w, None, (2, 1, 1), (1, 0, 0), (1, 0, 0)) if tr else F.conv3d(x, w, None, (stride, 1, 1), (pad, 0, 0), (dil, 1, 1))
    f = folded(w, Cout, scale, shift, act, tr, "d")
    desc = "%s B%d %d->%d %dx%dx%d act%d" % (form, B, Cin, Cout, D, H, W, act)
    try:
        out = N.conv_d(x.float().to(dev), f, k, stride, dil, pad, tr)
        compare("conv_d", desc, out, finish(raw, scale, shift, act), k * Cin)
    except Exception as e:
        FAILS.append(("conv_d", desc, "raised %s" % str(e)[:160]))


def fuzz_deconv(r, g):
    L = _lib.lib()
    B, Cin, Cout, H, W = r.randint(1, 2), r.randint(1, 64), r.randint(1, 32), r.randint(1, 40), r.randint(1, 60)
    x, w = rnd(g, B, Cin, H, W), rnd(g, Cin, Cout, 4, 4, scale=1.0 / (4 * Cin) ** 0.5)
    scale, shift = rnd(g, Cout).abs() + 0.5, rnd(g, Cout)
    act = r.choice([N.ACT_NONE, N.ACT_SILU, N.ACT_RELU])
    want = finish(F.conv_transpose2d(x, w, None, 2, 1), scale, shift, act)
    f = folded(w, Cout, scale, shift, act, True, "deconv2d")
    xg = x.float().to(dev)
    desc = "B%d %d->%d %dx%d act%d" % (B, Cin, Cout, H, W, act)
    out = torch.full((B, Cout, 2 * H, 2 * W), float("nan"), device=dev)
    try:
        _lib.check(L.ts_deconv2d_k4s2_fwd(_lib.ptr(xg), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout, H, W,
                                          act, out.stride(0), N._stream()), "ts_deconv2d_k4s2_fwd")
        compare("deconv_f32", desc, out, want, 4 * Cin)
    except Exception as e:
        FAILS.append(("deconv_f32", desc, "raised %s" % str(e)[:160]))
    if L.ts_conv3d_hw_x6s_supported(Cin, Cout, H, W, N.X6S_T4):
        out = torch.full((B, Cout, 2 * H, 2 * W), float("nan"), device=dev)
        try:
            _lib.check(L.ts_conv3d_hw_x6s_fwd(_lib.ptr(xg), _lib.ptr(N.x6s_weights(f, N.X6S_T4)), _lib.ptr(f.scale), _lib.ptr(f.shift),
                                              _lib.ptr(out), B, Cin, Cout, 1, H, W, N.X6S_T4, act, 0.0, Cin * H * W, H * W, out.stride(0),
                                              4 * H * W, N._stream()), "ts_conv3d_hw_x6s_fwd")
            compare("deconv_x6s", desc, out, want, 4 * Cin)
        except Exception as e:
            FAILS.append(("deconv_x6s", desc, "raised %s" % str(e)[:160]))


OPS = dict(hw=fuzz_hw, d=fuzz_d, deconv=fuzz_deconv)


def sweep(name, n, seed):
    r = random.Random(seed * 1000 + sum(map(ord, name)))
    g = torch.Generator().manual_seed(seed)
    del FAILS[:]
    for _ in range(n):
        OPS[name](r, g)
    torch.cuda.synchronize()
    return list(FAILS)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ops", default=",".join(OPS))
    a = ap.parse_args()
    bad = []
    for name in a.ops.split(","):
        try:
            found = sweep(name, a.n, a.seed)
        except Exception as e:
            found = list(FAILS) + [(name, "-", "the sweep itself raised %s: %s" % (type(e).__name__, e))]
        print("%-8s %d cases, %d findings" % (name, a.n, len(found)), flush=True)
        bad += found
    for f in bad:
        print("FINDING", *f)
    sys.exit(1 if bad else 0)
