"""GPU, BASELINE full size (configs[1]/[2]: 544x960, D=192): the all-HIP native engine against the nn.Module
path running the FRAMEWORK's convolutions (MIOpen) on the same calibrated synthetic network -- two
implementations that share no convolution / BatchNorm / activation code -- single frame and with a
temporal state (cost memory + local map) carried over from the first frame.

Bar: |EPE(native, gt*) - EPE(module, gt*)| < 1e-3 px with
gt* = module disparity + N(0,1) clipped to (0, D).  Top-k / sort are discrete, so single pixels may
move by O(1) px between two fp32 implementations; the mean may not."""
import os

import pytest
import torch

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
pytestmark = pytest.mark.gpu


def _delta_epe(a, b, seed, max_disp):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    gen = torch.Generator().manual_seed(seed)
    gt = (b + torch.randn(b.shape, generator=gen, dtype=torch.float64)).clamp(0, max_disp)
    return abs(float((a - gt).abs().mean()) - float((b - gt).abs().mean())), float((a - b).abs().mean())


def test_native_vs_framework_convolutions_full_size_single_and_temporal():
    import bench
    import synth
    from temporalstereo_amd import layers
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 3
    B = 2
    net = bench.build_model(dev, seed)
    inputs = bench.make_inputs(dev, seed, B)
    bench.calibrate_batchnorm(net, inputs)
    eng = InferenceEngine(net, backend="native", replay="plan")

    def module_pass(prev):
        layers.set_conv_backend("torch")
        try:
            with torch.no_grad():
                return net(*inputs, prev)
        finally:
            layers.set_conv_backend("hip")

    # ---- frame 0: single-frame mode
    ref = module_pass({})
    got = eng(*inputs, {})
    for i in range(4):
        scale = bench.RUN_W / ref[0][i].shape[-1]
        d, mad = _delta_epe(got[0][i] * scale, ref[0][i] * scale, seed + i, bench.MAX_DISP)
        assert d < 1e-3, "frame 0 disparity %d: dEPE %.3g px" % (i, d)
        assert mad < 2e-2, "frame 0 disparity %d: mean |diff| %.3g px" % (i, mad)
    assert [tuple(c.shape) for c in got[1]] == [(B, 5, 136, 240), (B, 7, 68, 120), (B, 14, 34, 60)]

    # ---- frame 1: temporal state from frame 0 (cost memory as written by the precise level; the last
    # disparity at 1/8 resolution as a one-plane local map, precise.py:98-103 / TemporalStereo.py:386-426)
    mem = {k: v.clone() for k, v in ref[5]["cost_memory"].items()}
    local = torch.nn.functional.interpolate(ref[0][0], size=(68, 120), mode="bilinear", align_corners=True) / 8.0
    prev = {"cost_memory": mem, "use_past_cost": True, "local_map": local.contiguous(), "local_map_size": 1}
    ref1 = module_pass(dict(prev))
    got1 = eng(*inputs, dict(prev))
    assert [tuple(c.shape) for c in got1[1]] == [(B, 5, 136, 240), (B, 8, 68, 120), (B, 14, 34, 60)]
    for i in range(4):
        scale = bench.RUN_W / ref1[0][i].shape[-1]
        d, mad = _delta_epe(got1[0][i] * scale, ref1[0][i] * scale, seed + 10 + i, bench.MAX_DISP)
        assert d < 1e-3, "frame 1 disparity %d: dEPE %.3g px" % (i, d)
        assert mad < 2e-2, "frame 1 disparity %d: mean |diff| %.3g px" % (i, mad)
    # the memory really is used: frame 1 differs from frame 0
    assert float((got1[0][0] - got[0][0]).abs().mean()) > 1e-5
    # second replay of the temporal plan gives the same answer (static buffers, no stale state)
    again = eng(*inputs, dict(prev))
    assert float((again[0][0] - got1[0][0]).abs().max()) == 0.0
