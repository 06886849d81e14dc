"""ts_conv3d_hw_x6s_fwd (csrc/conv_x6s.hip): the bf16-split ("x6") arithmetic for the stride-2 (1,3,3) convolution, the stride-2
transposed (1,3,3) convolution and the UNet's 4x4 stride-2 deconvolution, against fp64 convolutions and next to the f32-input MFMA
kernels they replace (reference layers: aggregation/TemporalStereo/module.py:111-184 -- DepthwiseConv3D / DepthwiseConvTranspose3D --
and module.py:453-457 -- UNet.deconv4 / deconv2; eval mode, BatchNorm folded to scale / shift).  Same bounds as tests/test_conv_x6_gpu.py:
within 1e-6 of the output's magnitude of the fp64 result and no worse than the f32 chain."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _x6s_on_every_grid(monkeypatch):
    """The engine keeps small grids on the f32 kernels (native._X6S_MIN_GRID); these tests are about the x6s kernel itself."""
    from temporalstereo_amd.aggregation import native as N
    monkeypatch.setattr(N, "_X6S_MIN_GRID", 1)
    monkeypatch.setattr(N, "_X6S_MIN_GRID_T3", 1)


def _bn(Cout, g, dev, nd):
    bn = (torch.nn.BatchNorm3d if nd == 3 else torch.nn.BatchNorm2d)(Cout).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    return bn


def _both(fn):
    from temporalstereo_amd.aggregation import native as N
    outs = {}
    for on in (True, False):
        N.X6S = on
        try:
            outs[on] = fn()
        finally:
            N.X6S = True
    torch.cuda.synchronize()
    return outs


def _errs(outs, ref):
    return (float((outs[True].double() - ref).abs().max()), float((outs[False].double() - ref).abs().max()), max(float(ref.abs().max()), 1.0))


def _check(e6, e32, scale):
    assert e32 <= 4e-6 * scale, (e32, scale)
    assert e6 <= 1e-6 * scale, (e6, scale)
    assert e6 <= e32 + 2.5e-7 * scale, (e6, e32, scale)


S2_CASES = [(1, 32, 64, 1, 40, 64), (2, 16, 32, 2, 17, 40), (1, 64, 64, 3, 33, 72), (1, 20, 12, 2, 24, 48), (1, 128, 80, 1, 16, 56),
            (2, 32, 64, 1, 272, 480)]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "B%d_%dto%d_D%d_%dx%d" % c)
def test_x6s_stride2(case):
    from temporalstereo_amd.aggregation import native as N
    B, Cin, Cout, D, H, W = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
    for act in (N.ACT_NONE, N.ACT_SILU):
        f = N.Folded(w, None, _bn(Cout, g, dev, 3), act, False, "hw")
        assert N._lib.lib().ts_conv3d_hw_x6s_supported(Cin, Cout, H, W, N.X6S_S2) == 1
        outs = _both(lambda: N.conv_hw(x, f, 2, 1))
        ref = F.conv3d(x.double(), w.double(), stride=(1, 2, 2), padding=(0, 1, 1))
        ref = ref * f.scale[:Cout].double().view(1, -1, 1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1, 1)
        if act == N.ACT_SILU:
            ref = F.silu(ref)
        assert outs[True].shape == ref.shape
        _check(*_errs(outs, ref))


T3_CASES = [(1, 64, 32, 2, 17, 32), (2, 32, 32, 3, 9, 20), (1, 16, 16, 1, 34, 60), (1, 40, 24, 2, 8, 36), (1, 64, 64, 6, 17, 32)]


@pytest.mark.parametrize("case", T3_CASES, ids=lambda c: "B%d_%dto%d_D%d_%dx%d" % c)
def test_x6s_transposed_k3(case):
    from temporalstereo_amd.aggregation import native as N
    B, Cin, Cout, D, H, W = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 11 + Cout)
    x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
    w = (torch.randn(Cin, Cout, 1, 3, 3, generator=g) / (9 * Cin / 4) ** 0.5).to(dev)      # ConvTranspose3d layout
    for act in (N.ACT_NONE, N.ACT_SILU):
        f = N.Folded(w, None, _bn(Cout, g, dev, 3), act, True, "hw")
        outs = _both(lambda: N.conv_hw(x, f, 2, 1, transposed=True))
        ref = F.conv_transpose3d(x.double(), w.double(), stride=(1, 2, 2), padding=(0, 1, 1), output_padding=(0, 1, 1))
        ref = ref * f.scale[:Cout].double().view(1, -1, 1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1, 1)
        if act == N.ACT_SILU:
            ref = F.silu(ref)
        assert outs[True].shape == ref.shape
        _check(*_errs(outs, ref))


T4_CASES = [(1, 32, 32, 24, 40, 16), (2, 32, 9, 17, 36, 0), (1, 64, 32, 136, 240, 16), (1, 48, 20, 9, 64, 3), (1, 32, 9, 272, 480, 0)]


@pytest.mark.parametrize("case", T4_CASES, ids=lambda c: "B%d_%dto%d_%dx%d_slice%d" % c)
def test_x6s_deconv_k4(case):
    """The 4x4 stride-2 deconvolution, written into a channel slice of a wider tensor (the UNet writes deconv4 next to the skip
    connection it is concatenated with, module.py:486-488): the channels beside the slice must stay untouched."""
    from temporalstereo_amd.aggregation import native as N
    B, Cin, Cout, H, W, extra = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 13 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    w = (torch.randn(Cin, Cout, 4, 4, generator=g) / (4 * Cin) ** 0.5).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    for act, bn in ((N.ACT_RELU, _bn(Cout, g, dev, 2)), (N.ACT_NONE, None)):
        f = N.Folded(w, bias, bn, act, True, "deconv2d")
        level = object.__new__(N.NativePrecise)

        def run():
            out = torch.full((B, Cout + extra, 2 * H, 2 * W), 7.0, device=dev)
            level._deconv(x, f, out, out.stride(0))
            return out
        outs = _both(run)
        ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
        ref = ref * f.scale[:Cout].double().view(1, -1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1)
        if act == N.ACT_RELU:
            ref = F.relu(ref)
        for o in outs.values():
            assert float((o[:, Cout:] - 7.0).abs().max()) == 0.0 if extra else True
        _check(*_errs({k: v[:, :Cout] for k, v in outs.items()}, ref))


def test_x6s_two_views_as_one_batch():
    """conv_hw(second=...): the two views of the image encoder as ONE batch of two through the batch stride (module.py:459-466)."""
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    a = torch.randn(1, 32, 1, 64, 96, generator=g).to(dev); b = torch.randn(1, 32, 1, 64, 96, generator=g).to(dev)
    w = (torch.randn(64, 32, 1, 3, 3, generator=g) / 17.0).to(dev)
    f = N.Folded(w, torch.randn(64, generator=g).to(dev), None, N.ACT_SILU, False, "hw")
    got = N.conv_hw(a, f, 2, 1, second=b)
    ref = F.silu(F.conv3d(torch.cat([a, b]).double(), w.double(), f.shift[:64].double(), stride=(1, 2, 2), padding=(0, 1, 1)))
    assert float((got.double() - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1.0)


def test_x6s_layer_selection_and_fallback():
    from temporalstereo_amd import _lib
    from temporalstereo_amd.aggregation import native as N
    L = _lib.lib()
    assert L.ts_conv3d_hw_x6s_supported(32, 64, 272, 480, N.X6S_S2) == 1
    assert L.ts_conv3d_hw_x6s_supported(32, 64, 34, 60, N.X6S_S2) == 0          # Wo = 30: output rows leave as aligned quads
    assert L.ts_conv3d_hw_x6s_supported(3, 32, 544, 960, N.X6S_S2) == 0         # the 3-channel image layer stays on the f32 kernel
    assert L.ts_conv3d_hw_x6s_supported(32, 9, 272, 480, N.X6S_T4) == 1
    assert L.ts_conv3d_hw_x6s_supported(32, 32, 17, 30, N.X6S_T3) == 0          # W % 4
    assert L.ts_conv3d_hw_x6s_weight_bytes(32, 9, N.X6S_T4) == 4 * 3 * 16 * 16 * 16
    assert L.ts_conv3d_hw_x6s_weight_bytes(32, 64, N.X6S_S2) == 4 * 3 * 12 * 64 * 16
    # unsupported geometry through the engine's entry: falls back to the f32 kernel, same answer
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 32, 2, 34, 60, generator=g).to(dev)
    w = (torch.randn(32, 32, 1, 3, 3, generator=g) / 17.0).to(dev)
    f = N.Folded(w, None, None, N.ACT_NONE, False, "hw")
    got = N.conv_hw(x, f, 2, 1)
    ref = F.conv3d(x.double(), w.double(), stride=(1, 2, 2), padding=(0, 1, 1))
    assert float((got.double() - ref).abs().max()) <= 4e-6 * max(float(ref.abs().max()), 1.0)


def test_x6s_fuzz_against_the_f32_kernels():
    """Random geometries (one-row images, W = 4 | 8, ragged channel counts, many planes): the two kernels must agree to fp32 rounding
    everywhere, including the zero padding, the tile edges and the parity classes of the transposed forms."""
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(20260930)
    for case in range(36):
        mode = case % 3
        B = int(rng.randint(1, 4)); Cin = int(rng.choice([16, 17, 24, 31, 32, 40, 64, 100])); Cout = int(rng.choice([2, 9, 12, 16, 20, 32, 33, 48, 64]))
        D = int(rng.randint(1, 5)); H = int(rng.randint(1, 30)); W = (8 if mode == 0 else 4) * int(rng.randint(1, 12))
        g = torch.Generator().manual_seed(2000 + case)
        if mode == 2:
            D = 1; Cout = min(Cout, 32)
            x = torch.randn(B, Cin, H, W, generator=g).to(dev)
            w = (torch.randn(Cin, Cout, 4, 4, generator=g) / (4 * Cin) ** 0.5).to(dev)
            f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, N.ACT_NONE, True, "deconv2d")
            level = object.__new__(N.NativePrecise)

            def run():
                out = torch.empty((B, Cout, 2 * H, 2 * W), device=dev)
                level._deconv(x, f, out, out.stride(0))
                return out
        else:
            x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
            shape = (Cout, Cin, 1, 3, 3) if mode == 0 else (Cin, Cout, 1, 3, 3)
            w = (torch.randn(*shape, generator=g) / (9 * Cin) ** 0.5).to(dev)
            f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, N.ACT_NONE, mode == 1, "hw")

            def run():
                return N.conv_hw(x, f, 2, 1, transposed=(mode == 1))
        outs = _both(run)
        scale = max(float(outs[False].abs().max()), 1.0)
        err = float((outs[True] - outs[False]).abs().max())
        assert torch.isfinite(outs[True]).all()
        assert err <= 4e-6 * scale, (case, mode, (B, Cin, Cout, D, H, W), err, scale)
