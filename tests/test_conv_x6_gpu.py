"""ts_conv3d_hw_x6_fwd: the stride-1 (1,3,3) convolution with fp32 products assembled from bf16 pieces (six bf16 MFMAs per
product block, a chunk's products summed apart and added to the running sum in fp32) against an fp64 convolution, next to the
f32-MFMA kernel it replaces (reference: the Conv3d wrappers of layers/basic_layers.py:194-235 in eval mode, BatchNorm folded to
scale / shift).  Measured (tools/exp/x6_accuracy_sweep.py): max error / max |output| 0.15e-6 ... 0.26e-6 for Cin 16 ... 512, the
f32 kernel 0.3e-6 ... 0.7e-6 -- the split form is the more accurate of the two at every reduction length.  (Accumulating all
chunks inside the matrix core's accumulator instead was up to 3.6x WORSE than the f32 chain at 1,600+ terms: the core aligns an
instruction's 32 products to the largest addend, the accumulator included.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _x6_on_every_grid(monkeypatch):
    """The engine keeps small grids on the f32 kernel (native._X6_MIN_GRID); these tests are about the x6 kernel itself."""
    from temporalstereo_amd.aggregation import native as N
    monkeypatch.setattr(N, "_X6_MIN_GRID", 1)


def _run(B, Cin, Cout, D, H, W, dilation, act, with_addend, seed):
    from temporalstereo_amd import _lib
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
    bn = torch.nn.BatchNorm3d(Cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    f = N.Folded(w, None, bn, act, False, "hw")
    addend = torch.randn(B, Cout, 1, H, W, generator=g).to(dev) if with_addend else None
    outs = {}
    for x6 in (True, False):
        N.X6 = x6
        try:
            outs[x6] = N.conv_hw(x, f, 1, dilation, addend=addend)
        finally:
            N.X6 = True
    torch.cuda.synchronize()
    xd, wd = x.double(), w.double()
    ref = F.conv3d(xd, wd, padding=(0, dilation, dilation), dilation=(1, dilation, dilation))
    if addend is not None:
        ref = ref + addend.double()
    ref = ref * f.scale[:Cout].double().view(1, -1, 1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1, 1)
    if act == N.ACT_SILU:
        ref = F.silu(ref)
    e6 = float((outs[True].double() - ref).abs().max())
    e32 = float((outs[False].double() - ref).abs().max())
    return e6, e32, float(ref.abs().max())


CASES = [
    # B, Cin, Cout, D, H, W, dilation, with_addend
    (1, 16, 16, 2, 24, 64, 1, False),
    (2, 32, 32, 3, 37, 44, 1, False),          # ragged tile edges
    (1, 176, 12, 5, 34, 60, 1, True),          # 11 chunks, a ragged channel block, addend
    (1, 20, 24, 2, 16, 36, 1, False),          # ragged channel counts on both sides
    (1, 64, 64, 1, 40, 72, 1, False),          # two output-channel groups
    (1, 48, 32, 2, 24, 40, 2, False),          # dilation 2
    (1, 128, 32, 1, 68, 120, 1, False),
    (1, 32, 144, 1, 24, 40, 1, False),         # more than 64 output channels: 32-channel groups over the grid's z
    (1, 512, 16, 1, 32, 64, 1, False),         # 4,608-term reductions
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_%dto%d_D%d_%dx%d_dil%d%s" % (c[:7] + ("_addend" if c[7] else "",)))
def test_x6_is_as_accurate_as_the_f32_mfma_kernel(case):
    from temporalstereo_amd.aggregation import native as N
    B, Cin, Cout, D, H, W, dil, add = case
    for act in (N.ACT_NONE, N.ACT_SILU):
        e6, e32, scale = _run(B, Cin, Cout, D, H, W, dil, act, add, seed=Cin * 7 + Cout)
        # both within a few fp32 ulps of the exact result; the split form within 1e-6 of the output's magnitude and no worse than
        # the f32 chain (+1 ulp of slack)
        assert e32 <= 4e-6 * max(scale, 1.0), (e32, scale)
        assert e6 <= 1e-6 * max(scale, 1.0), (e6, scale)
        assert e6 <= 1.0 * e32 + 2.5e-7 * max(scale, 1.0), (e6, e32, scale)


def test_x6_layer_selection():
    from temporalstereo_amd import _lib
    L = _lib.lib()
    assert L.ts_conv3d_hw_x6_supported(16, 16, 240, 1, 1, 0) == 1
    assert L.ts_conv3d_hw_x6_supported(176, 8, 240, 1, 1, 0) == 0         # Cout <= 8: the row-paired f32 kernel is the faster one
    assert L.ts_conv3d_hw_x6_supported(176, 64, 60, 1, 2, 0) == 1
    assert L.ts_conv3d_hw_x6_supported(8, 8, 240, 1, 1, 0) == 0          # fewer channels than one K chunk
    assert L.ts_conv3d_hw_x6_supported(32, 32, 30, 1, 1, 0) == 0         # rows are staged as aligned quads: W % 4 == 0
    assert L.ts_conv3d_hw_x6_supported(32, 32, 240, 2, 1, 0) == 0        # stride 2 stays on the f32 kernel
    assert L.ts_conv3d_hw_x6_supported(32, 32, 240, 2, 1, 1) == 0
    assert L.ts_conv3d_hw_x6_weight_bytes(176, 8) == 11 * 3 * 10 * 2 * 8 * 16          # [chunk][part][slot][group][CoutPad 8][8 bf16]


def test_x6_fuzz_against_the_f32_kernel():
    """Random geometries (tiny images, one-row tiles, W = 4, ragged channel counts on both sides, many planes, dilation 2):
    the two kernels must agree to fp32 rounding everywhere, including the zero padding and the tile edges."""
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(20260929)
    for case in range(30):
        B = int(rng.randint(1, 4)); Cin = int(rng.choice([16, 17, 24, 31, 32, 40, 64, 100])); Cout = int(rng.choice([9, 12, 16, 20, 32, 33, 48, 64, 80]))
        D = int(rng.randint(1, 6)); H = int(rng.randint(1, 40)); W = 4 * int(rng.randint(1, 20)); dil = int(rng.choice([1, 1, 2]))
        g = torch.Generator().manual_seed(1000 + case)
        x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
        f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, N.ACT_NONE, False, "hw")
        outs = {}
        for x6 in (True, False):
            N.X6 = x6
            try:
                outs[x6] = N.conv_hw(x, f, 1, dil)
            finally:
                N.X6 = True
        torch.cuda.synchronize()
        scale = float(outs[False].abs().max())
        err = float((outs[True] - outs[False]).abs().max())
        assert err <= 4e-6 * max(scale, 1.0), (case, (B, Cin, Cout, D, H, W, dil), err, scale)


@pytest.mark.parametrize("shape", [(1, 352, 32, 12, 34, 60, 1), (1, 256, 64, 1, 34, 60, 1), (2, 272, 16, 2, 9, 16, 1), (1, 512, 24, 3, 17, 32, 2)],
                         ids=lambda s: "B%d_%dto%d_D%d_%dx%d_dil%d" % s)
def test_x6_split_k_equals_the_unsplit_form(shape):
    """Small grids with long reductions (Cin >= 256) cut the input channels into slices (raw sums to a workspace, summed in a fixed order by conv_splitk_finish);
    without a workspace the same entry point runs unsplit: the two differ only in the order of a few fp32 additions -- and both
    stay within the x6 bound of the fp64 result, with scale / shift / SiLU / addend applied by the finishing launch."""
    from temporalstereo_amd import _lib
    from temporalstereo_amd.aggregation import native as N
    B, Cin, Cout, D, H, W, dil = shape
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
    f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, N.ACT_SILU, False, "hw")
    addend = torch.randn(B, Cout, 1, H, W, generator=g).to(dev)
    L = _lib.lib()
    wsb = int(L.ts_conv3d_hw_x6_workspace_bytes(B, Cin, Cout, D, H, W))
    assert wsb > 0 and wsb % (B * Cout * D * H * W * 4) == 0, "these grids are small enough to be split"
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    outs = []
    for buf, nbytes in ((None, 0), (ws, wsb)):
        out = torch.full((B, Cout, D, H, W), float("nan"), device=dev)
        rc = L.ts_conv3d_hw_x6_fwd(_lib.ptr(x), _lib.ptr(N.x6_weights(f)), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out), B, Cin, Cout,
                                   D, H, W, dil, N.ACT_SILU, 0.0, x.stride(0), x.stride(1), out.stride(0), out.stride(1), _lib.ptr(addend),
                                   addend.stride(0), _lib.ptr(buf), nbytes, N._stream())
        _lib.check(rc, "ts_conv3d_hw_x6_fwd")
        outs.append(out)
    torch.cuda.synchronize()
    ref = F.conv3d(x.double(), w.double(), padding=(0, dil, dil), dilation=(1, dil, dil)) + addend.double()
    ref = F.silu(ref * f.scale[:Cout].double().view(1, -1, 1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1, 1))
    scale = max(float(ref.abs().max()), 1.0)
    for out in outs:
        assert torch.isfinite(out).all()
        assert float((out.double() - ref).abs().max()) <= 1e-6 * scale
    assert float((outs[0] - outs[1]).abs().max()) <= 5e-7 * scale


@pytest.mark.parametrize("rows", ["8", "4"])
def test_ping_pong_form_forced_on_every_grid(rows):
    """ig_conv_x6p_kernel (csrc/conv_x6p.hip) is chosen by grid size, so the small shapes above mostly run the older kernel.  Its
    switches are read once per process: a child process with the form forced on every grid (TS_X6P_MIN_WGS=1) and the half-tile height
    pinned runs the accuracy sweep of tools/exp/x6p_check.py -- ragged tiles, ragged channel counts on both sides, 5 channel groups,
    an addend, one-row / four-column images, 272 x 480 (several tiles per persistent workgroup) -- against fp64 convolutions with the
    bound of this file (1e-6 of the output's magnitude)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TS_X6P_MIN_WGS="1", TS_X6P_HR=rows)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "x6p_check.py"), "--child", "acc"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("acc ")]
    assert len(lines) >= 30 and not any("FAIL" in l for l in lines), "\n".join(l for l in lines if "FAIL" in l)
    worst = float([l for l in r.stdout.splitlines() if l.startswith("worst")][0].split()[1])
    assert worst <= 1e-6
    # ... and 60 random geometries against the f32-MFMA kernel (to fp32 rounding: 4e-6 of the output's magnitude), SiLU / ReLU / none,
    # with and without an addend, split-K where the workspace rule gives one
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "exp", "x6p_check.py"), "--child", "fuzz"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("fuzz ")]
    assert len(lines) == 60 and not any("FAIL" in l for l in lines), "\n".join(l for l in lines if "FAIL" in l)


def test_silu_epilogue_at_the_ends_of_the_range():
    """The epilogues' SiLU is v * rcp(1 + exp2(-v log2 e)) with one Newton step (csrc/conv_common.hpp silu_fast).  Below -88.7 e^-v is inf,
    the reciprocal 0 and a Newton step there inf * 0 = NaN (found by tests/test_native_gpu.py loading weights far from the trained scale):
    every conv kernel and the resize-add-SiLU kernel over sums from -1e30 to 1e30 -- finite everywhere, within 2 ulp of torch's fp64
    SiLU plus the exponent's rounding (|v| 2^-24 relative, on results that are ~|v| e^-|v| themselves), -0 where fp32's own v e^v is."""
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    ends = [-1e30, -1e8, -1e4, -200.0, -104.0, -100.0, -90.0, -88.8, -88.7, -87.0, -83.2, -83.0, -50.0, -20.0, -5.0, -1.0, -1e-3, -1e-30,
            0.0, 1e-30, 1e-3, 1.0, 5.0, 20.0, 50.0, 83.2, 88.8, 100.0, 200.0, 1e4, 1e8, 1e30]
    vals = torch.tensor(ends, dtype=torch.float32).double()
    ref = F.silu(vals)
    tol = ref.abs() * (2.5e-7 + 1.2e-7 * (-vals).clamp_min(0.0)) + 1e-35

    def check(got, what, lo=0):
        got = got.double().cpu()
        n = min(len(got), len(ends) - lo)
        got, r, t = got[:n], ref[lo:lo + n], tol[lo:lo + n]
        assert bool(torch.isfinite(got).all()), (what, got)
        bad = (got - r).abs() > t
        assert not bool(bad.any()), (what, [(ends[lo + i], float(got[i]), float(r[i])) for i in torch.nonzero(bad).flatten().tolist()])

    # (1, 32, 4, 64, 256) with 32 output channels: 256 work items, the ping-pong kernel; the small grids stay on ig_conv_x6_kernel / the f32 kernel
    for shape, cout, x6 in (((1, 32, 4, 64, 256), 32, True), ((1, 32, 2, 16, 32), 32, True), ((1, 32, 2, 16, 32), 16, True), ((1, 32, 2, 16, 32), 32, False)):
        x = torch.zeros(shape, device=dev)
        w = torch.zeros(cout, 32, 1, 3, 3, device=dev)
        for lo in range(0, len(ends), cout):
            bias = (ends[lo:lo + cout] + [0.0] * cout)[:cout]
            f = N.Folded(w, torch.tensor(bias, device=dev), None, N.ACT_SILU, False, "hw")
            N.X6 = x6
            try:
                y = N.conv_hw(x, f, 1, 1)
            finally:
                N.X6 = True
            assert bool((y == y[:, :, :1, :1, :1]).all()), (shape, cout, x6)
            check(y[0, :, 0, 0, 0], (shape, cout, x6), lo)
    a = torch.tensor(ends, device=dev).view(1, len(ends), 1, 1, 1).expand(1, len(ends), 2, 4, 8).contiguous()
    check(N.resize_add_act(a, None, (2, 4, 8), N.ACT_SILU)[0, :, 0, 0, 0], "resize_add_act")
    check(N.resize_add_act(a * 0.5, a * 0.5, (2, 4, 8), N.ACT_SILU)[0, :, 1, 3, 7], "resize_add_act with an addend")
