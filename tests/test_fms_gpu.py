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


def test_inverse_warp_3d_value_test_and_oracle():
    """temporalstereo_amd.inverse_warp_3d(img, disp): the function-level seam of layers/inverse_warp_3d.py:4.
    (1) the reference's own output on the 3x4 value-test grid (golden/value_test_warp.npz: inverse_warp_3d(right, -disp) as
    recorded from the imported reference); (2) the oracle's warp on a config geometry (fine level of configs[1]: 68 x 120, five
    candidates, a channel count that is not a multiple of 8); (3) both gradients against fp64 autograd of the oracle's form;
    (4) the 5-D forms (an expanded view, as cat_fms.py:28-31 builds it, and a genuinely per-plane image); (5) what is refused."""
    import temporalstereo_amd as ts
    from oracle import cost_volume as O
    dev = _dev()
    g = load("value_test_warp")
    r, d = t(g["right"], dev), t(g["disp"], dev)
    np.testing.assert_allclose(ts.inverse_warp_3d(r, -d).cpu().numpy(), g["warped"], atol=1e-5)

    B, C, D, H, W = 2, 12, 5, 68, 120
    img = t(synth.normal(11, "img", (B, C, H, W)), dev)
    disp = t(synth.uniform(12, "disp", (B, D, H, W), -9.0, 40.0), dev)
    ref = O.warp_candidates(img.cpu(), (-disp).cpu())                 # the oracle warps by -disp like the reference's callers
    out = ts.inverse_warp_3d(img, disp)
    assert out.shape == (B, C, D, H, W)
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=2e-5)

    # gradients: fp64 autograd through the oracle's form of the same warp
    B, C, D, H, W = 1, 8, 3, 9, 16
    img = t(synth.normal(13, "img", (B, C, H, W)), dev).requires_grad_(True)
    disp = t(synth.uniform(14, "disp", (B, D, H, W), -3.0, 6.0), dev).requires_grad_(True)
    gout = t(synth.normal(15, "g", (B, C, D, H, W)), dev)
    ts.inverse_warp_3d(img, disp).backward(gout)
    i64 = img.detach().double().cpu().requires_grad_(True)
    d64 = disp.detach().double().cpu().requires_grad_(True)
    O.warp_candidates(i64, -d64).backward(gout.double().cpu())
    np.testing.assert_allclose(img.grad.cpu().numpy(), i64.grad.numpy(), rtol=1e-4, atol=1e-5)
    # the disparity gradient of a linear interpolation is the local slope: where a sample sits within 1e-4 of a pixel centre the fp32
    # and fp64 positions can fall on different sides of it -- those elements are left out (they are the warp's own kinks)
    pos = (torch.arange(W).view(1, 1, 1, W) + d64.detach())
    smooth = ((pos - pos.round()).abs() > 1e-3).numpy()
    np.testing.assert_allclose(disp.grad.cpu().numpy()[smooth], d64.grad.numpy()[smooth], rtol=1e-4, atol=1e-4)

    # 5-D images
    img = t(synth.normal(16, "img", (2, 8, 7, 12)), dev)
    disp = t(synth.uniform(17, "disp", (2, 4, 7, 12), -2.0, 5.0), dev)
    a = ts.inverse_warp_3d(img.unsqueeze(2).expand(2, 8, 4, 7, 12), disp)
    np.testing.assert_allclose(a.cpu().numpy(), ts.inverse_warp_3d(img, disp).cpu().numpy(), atol=0)
    vol = t(synth.normal(18, "vol", (2, 8, 4, 7, 12)), dev)
    b5 = ts.inverse_warp_3d(vol, disp)
    for k in range(4):
        np.testing.assert_allclose(b5[:, :, k].cpu().numpy(), ts.inverse_warp_3d(vol[:, :, k].contiguous(), disp)[:, :, k].cpu().numpy(), atol=1e-6)

    with pytest.raises(ValueError):
        ts.inverse_warp_3d(img[0], disp)                                # 3-D image: the reference's own ValueError (inverse_warp_3d.py:31-33)
    with pytest.raises(RuntimeError):
        ts.inverse_warp_3d(img, disp, padding_mode='border')
    with pytest.raises(RuntimeError):
        ts.inverse_warp_3d(img, disp, disp_Y=disp)
