"""GPU: cat_fms / dif_fms (dense siblings of block_cost) against what the real
reference produced (tests/golden/fms_*.npz), against the oracle on other shapes, and the reference-authored
3x4 value test (cat_fms.py:48-69, dif_fms.py:56-76).  Tolerance 2e-5: the warp goes through the same
normalise / de-normalise float sequence as grid_sample, products differ by fma contraction only."""
import numpy as np
import pytest
import torch

import synth
from helpers import load, t

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["fms_0", "fms_1", "fms_2"])
def test_fms_golden(name):
    import temporalstereo_amd as ts
    g = load(name); dev = _dev()
    l, r, d = t(g["left"], dev), t(g["right"], dev), t(g["disp"], dev)
    np.testing.assert_allclose(ts.cat_fms(l, r, d).cpu().numpy(), g["cat"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ts.dif_fms(l, r, d).cpu().numpy(), g["dif"], rtol=1e-5, atol=2e-5)


def test_fms_value_test_known_answer():
    """left = 1..12, right = 13..24 on a 3x4 grid, shifts -2..2 (the prints of cat_fms.py:60-69); the kernels
    work on 8-channel groups, so the single channel is replicated 8 times."""
    import temporalstereo_amd as ts
    g = load("fms_value_test"); dev = _dev()
    l, r, d = (t(g[k], dev) for k in ("left", "right", "disp"))
    cat = ts.cat_fms(l.repeat(1, 8, 1, 1), r.repeat(1, 8, 1, 1), d).cpu().numpy()
    dif = ts.dif_fms(l.repeat(1, 8, 1, 1), r.repeat(1, 8, 1, 1), d).cpu().numpy()
    for c in range(8):
        np.testing.assert_allclose(cat[:, c], g["cat"][:, 0], atol=1e-5)
        np.testing.assert_allclose(cat[:, 8 + c], g["cat"][:, 1], atol=1e-5)
        np.testing.assert_allclose(dif[:, c], g["dif"][:, 0], atol=1e-5)
    # integer shift -2 at the first row: the right row 13,14,15,16 moves two columns, zeros enter
    np.testing.assert_allclose(cat[0, 8, 0, 0], [15, 16, 0, 0], atol=1e-5)


# (1, 8, 3, 480, 4): an ALIGNED map too wide for the run-order form's LDS footprint -- the row-order form must take over
@pytest.mark.parametrize("shape", [(1, 32, 11, 40, 48), (2, 8, 5, 23, 7), (1, 16, 6, 312, 5), (1, 8, 3, 480, 4)])
def test_fms_vs_oracle(shape):
    import oracle.cost_volume as ocv
    import temporalstereo_amd as ts
    B, C, H, W, D = shape
    dev = _dev()
    l = synth.normal(71, "l", (B, C, H, W)); r = synth.normal(72, "r", (B, C, H, W))
    d = synth.uniform(73, "d", (B, D, H, W), -4.0, W / 2.0)
    d[:, 0] = np.round(d[:, 0])
    np.testing.assert_allclose(ts.cat_fms(t(l, dev), t(r, dev), t(d, dev)).cpu().numpy(),
                               ocv.cat_fms(t(l), t(r), t(d)).numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ts.dif_fms(t(l, dev), t(r, dev), t(d, dev)).cpu().numpy(),
                               ocv.dif_fms(t(l), t(r), t(d)).numpy(), rtol=1e-5, atol=2e-5)


def test_fms_is_block_cost_without_the_correlation_channels_and_rejects_bad_input():
    import temporalstereo_amd as ts
    dev = _dev()
    l, r = t(synth.normal(81, "l", (1, 16, 12, 36)), dev), t(synth.normal(82, "r", (1, 16, 12, 36)), dev)
    d = t(synth.uniform(83, "d", (1, 5, 12, 36), 0.0, 20.0), dev)
    assert torch.equal(ts.cat_fms(l, r, d), ts.block_cost(l, r, d, 3)[:, :32])
    with pytest.raises(ValueError):
        ts.cat_fms(l[:, :12], r[:, :12], d)                      # C % 8
    with pytest.raises(RuntimeError):
        ts.dif_fms(l, r, d[:, :1])                                # D == 1 divides by zero in the reference's warp


@pytest.mark.parametrize("shape", [(2, 8, 9, 13, 3), (1, 16, 12, 40, 5), (1, 8, 6, 130, 2)])
def test_fms_backward_vs_oracle_autograd(shape):
    """Gradients of cat_fms / dif_fms (left, right, candidates) against autograd through the oracle's torch formulation
    (the same ops as the reference).  dif_fms: candidates are kept away from the `warped > 0` threshold."""
    import oracle.cost_volume as ocv
    import temporalstereo_amd as ts
    B, C, H, W, D = shape
    dev = _dev()
    L = synth.normal(31, "L", (B, C, H, W)); R = np.abs(synth.normal(31, "R", (B, C, H, W))) + 0.2      # warped > 0 inside the image
    disp = synth.uniform(31, "d", (B, D, H, W), -2.0, W * 0.5)
    for name, fn_gpu, fn_cpu, cout in (("cat_fms", ts.cat_fms, ocv.cat_fms, 2 * C), ("dif_fms", ts.dif_fms, ocv.dif_fms, C)):
        gout = synth.normal(32, "g" + name, (B, cout, D, H, W))
        lc, rc, dc = (t(v).requires_grad_() for v in (L, R, disp))
        fn_cpu(lc, rc, dc).backward(t(gout))
        lg, rg, dg = (t(v, dev).requires_grad_() for v in (L, R, disp))
        out = fn_gpu(lg, rg, dg)
        assert out.requires_grad
        out.backward(t(gout, dev))
        for got, want, what in ((lg.grad, lc.grad, "left"), (rg.grad, rc.grad, "right"), (dg.grad, dc.grad, "candidates")):
            scale = float(want.abs().max()) + 1e-6
            err = float((got.cpu() - want).abs().max())
            assert err <= 3e-4 * scale + 1e-5, "%s grad %s: max err %g (scale %g)" % (name, what, err, scale)
