"""GPU: K4 (regression) and K2 (splat / re-projection) through the C ABI vs oracle + golden vectors."""
import numpy as np
import pytest
import torch

import oracle
import synth
from helpers import load, t

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("name,k", [("topk_softargmax", 2), ("topk_softargmax_k3", 3)])
def test_topk_softargmax_golden(name, k):
    import temporalstereo_amd as ts
    g = load(name); dev = _dev()
    d, td, tc = ts.topk_softargmax(t(g["cost"], dev), t(g["samp"], dev), t(g["off"], dev), k=k)
    np.testing.assert_allclose(d.cpu().numpy(), g["disp"], rtol=1e-6, atol=1e-5)
    np.testing.assert_array_equal(tc.cpu().numpy(), g["topk_cost"])            # selection is exact
    np.testing.assert_allclose(td.cpu().numpy(), g["topk_disp"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape,k", [((1, 14, 34, 60), 2), ((2, 7, 68, 120), 2), ((1, 5, 136, 240), 2),
                                     ((1, 3, 5, 7), 1), ((1, 12, 9, 11), 8)])
def test_topk_softargmax_vs_oracle(shape, k):
    import temporalstereo_amd as ts
    dev = _dev()
    cost = synth.normal(1, "c", shape); samp = synth.uniform(1, "s", shape, 0, 48); off = synth.uniform(1, "o", shape, -1, 1)
    e = oracle.topk_softargmax(t(cost), t(samp), t(off), k=k)
    g = ts.topk_softargmax(t(cost, dev), t(samp, dev), t(off, dev), k=k)
    for a, b in zip(g, e):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=1e-6, atol=2e-5)


def test_topk_tie_break_lowest_index_first():
    import temporalstereo_amd as ts
    dev = _dev()
    cost = torch.zeros(1, 6, 2, 3, device=dev)
    samp = torch.arange(6, dtype=torch.float32, device=dev).view(1, 6, 1, 1).expand(1, 6, 2, 3).contiguous()
    d, td, tc = ts.topk_softargmax(cost, samp, torch.zeros_like(cost), k=2)
    assert torch.equal(td[:, 0], samp[:, 0]) and torch.equal(td[:, 1], samp[:, 1])
    np.testing.assert_allclose(d.cpu().numpy(), 0.5)


def test_topk_softargmax_backward():
    import temporalstereo_amd as ts
    dev = _dev()
    shape = (2, 9, 6, 10)
    cost = synth.normal(2, "c", shape); samp = synth.uniform(2, "s", shape, 0, 30); off = synth.uniform(2, "o", shape, -1, 1)
    gd = synth.normal(2, "gd", (2, 1, 6, 10)); gt = synth.normal(2, "gt", (2, 2, 6, 10)); gc = synth.normal(2, "gc", (2, 2, 6, 10))
    ins_c = [t(a).requires_grad_() for a in (cost, samp, off)]
    outs = oracle.topk_softargmax(*ins_c, k=2)
    torch.autograd.backward(outs, [t(gd), t(gt), t(gc)])
    ins_g = [t(a, dev).requires_grad_() for a in (cost, samp, off)]
    outs = ts.topk_softargmax(*ins_g, k=2)
    torch.autograd.backward(outs, [t(gd, dev), t(gt, dev), t(gc, dev)])
    for a, b in zip(ins_g, ins_c):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=1e-5)


def test_soft_argmin_golden_and_argmin():
    import temporalstereo_amd as ts
    g = load("soft_argmin"); dev = _dev()
    c, s = t(g["cost"], dev), t(g["samp"], dev)
    np.testing.assert_allclose(ts.soft_argmin(c, s).cpu().numpy(), g["disp"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ts.soft_argmin(c, s, temperature=2.0).cpu().numpy(), g["disp_t2"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(ts.argmin_select(c, s).cpu().numpy(), g["disp_argmin"])


@pytest.mark.parametrize("shape", [(1, 192, 20, 33), (2, 48, 17, 40), (1, 3, 4, 5), (1, 1, 3, 3)])
@pytest.mark.parametrize("normalize", [True, False])
def test_soft_argmin_vs_oracle_fwd_bwd(shape, normalize):
    """D up to 192 (the dense use of BASELINE.json); fp32 softmax over 192 terms: 1e-4 abs on ~96."""
    import temporalstereo_amd as ts
    dev = _dev()
    cost = synth.normal(3, "c", shape, 2.0); samp = synth.uniform(3, "s", shape, 0, 192)
    gd = synth.normal(3, "g", (shape[0], 1) + shape[2:])
    cc, sc = t(cost).requires_grad_(), t(samp).requires_grad_()
    e = oracle.soft_argmin(cc, sc, temperature=0.7, normalize=normalize); e.backward(t(gd))
    cg, sg = t(cost, dev).requires_grad_(), t(samp, dev).requires_grad_()
    o = ts.soft_argmin(cg, sg, temperature=0.7, normalize=normalize); o.backward(t(gd, dev))
    scale = float(e.detach().abs().max()) + 1.0
    np.testing.assert_allclose(o.detach().cpu().numpy(), e.detach().numpy(), rtol=1e-5, atol=2e-6 * scale)
    np.testing.assert_allclose(cg.grad.cpu().numpy(), cc.grad.numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(sg.grad.cpu().numpy(), sc.grad.numpy(), rtol=1e-4, atol=1e-5 * scale)


# ---------------------------------------------------------------------------------------------- K2
@pytest.mark.parametrize("name", ["project_to_3d", "project_to_3d_k3"])
def test_project_to_3d_golden(name):
    import temporalstereo_amd as ts
    g = load(name); dev = _dev()
    K = t(g["K"], dev)
    inv_K = torch.inverse(K) if K.shape[-1] == 4 else None
    o = ts.project_to_3d(t(g["depth"], dev), K, inv_K, t(g["T"], dev))
    np.testing.assert_allclose(o["triangular_depth"].cpu().numpy(), g["triangular_depth"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(o["optical_flow"].cpu().numpy(), g["optical_flow"], rtol=1e-5, atol=2e-4)
    if "flow_mask" in g:
        assert (o["flow_mask"].cpu().numpy() == g["flow_mask"]).mean() > 0.99     # boundary pixels may flip


def _splat_inputs(seed, B, C, H, W, scale=2.0):
    return (synth.normal(seed, "in", (B, C, H, W)), synth.normal(seed, "fl", (B, 2, H, W), scale),
            synth.normal(seed, "me", (B, 1, H, W)))


def test_splat_known_answers():
    """The analytic cases that pin the oracle, on the GPU kernel itself."""
    import temporalstereo_amd as ts
    dev = _dev()
    x = t(synth.normal(4, "x", (2, 3, 6, 8)), dev)
    z = torch.zeros(2, 2, 6, 8, device=dev)
    assert torch.equal(ts.FunctionSoftsplat(x, z, None, 'summation'), x)              # zero flow = identity
    f = z.clone(); f[:, 0] = 3.0; f[:, 1] = -2.0                                      # integer shift
    out = ts.FunctionSoftsplat(x, f, None, 'summation')
    exp = torch.zeros_like(x); exp[:, :, 0:4, 3:8] = x[:, :, 2:6, 0:5]
    assert torch.equal(out, exp)
    m = torch.full((2, 1, 6, 8), 0.7, device=dev)                                     # softmax(const) = average
    f2 = t(synth.normal(4, "f", (2, 2, 6, 8), 1.5), dev)
    np.testing.assert_allclose(ts.FunctionSoftsplat(x, f2, m, 'softmax').cpu().numpy(),
                               ts.FunctionSoftsplat(x, f2, None, 'average').cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(4, 4, 68, 120), (2, 3, 60, 80), (1, 1, 5, 7), (8, 4, 48, 156)])
@pytest.mark.parametrize("mode", ["summation", "softmax", "average", "linear"])
def test_splat_vs_oracle(shape, mode):
    """fp32 atomics: summation order differs -> rtol 1e-5 of the accumulated magnitude (Appendix B.4)."""
    import temporalstereo_amd as ts
    dev = _dev()
    x, f, m = _splat_inputs(5, *shape)
    if mode == "linear":
        m = np.abs(m) + 0.1
    e = oracle.softsplat(t(x), t(f), None if mode in ("summation", "average") else t(m), mode)
    o = ts.FunctionSoftsplat(t(x, dev), t(f, dev), None if mode in ("summation", "average") else t(m, dev), mode)
    np.testing.assert_allclose(o.cpu().numpy(), e.numpy(), rtol=2e-4, atol=2e-5)


def test_splat_backward_vs_oracle_autograd():
    import temporalstereo_amd as ts
    dev = _dev()
    x, f, m = _splat_inputs(6, 2, 3, 9, 12, scale=1.3)
    f = f + 0.11
    g = synth.normal(6, "g", (2, 3, 9, 12))
    for mode in ("summation", "softmax"):
        xc, fc, mc = t(x).requires_grad_(), t(f).requires_grad_(), t(m).requires_grad_()
        oracle.softsplat(xc, fc, mc if mode == "softmax" else None, mode).backward(t(g))
        xg, fg, mg = t(x, dev).requires_grad_(), t(f, dev).requires_grad_(), t(m, dev).requires_grad_()
        ts.FunctionSoftsplat(xg, fg, mg if mode == "softmax" else None, mode).backward(t(g, dev))
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xc.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(fg.grad.cpu().numpy(), fc.grad.numpy(), rtol=1e-3, atol=1e-4)
        if mode == "softmax":
            np.testing.assert_allclose(mg.grad.cpu().numpy(), mc.grad.numpy(), rtol=1e-3, atol=1e-4)


def test_splat_deterministic_mode_is_bit_reproducible():
    """fp32 atomics make the default splat order-dependent; the fixed-point mode is not.
    Heavy collisions on purpose: every pixel is pushed into a 6x6 window."""
    import temporalstereo_amd as ts
    dev = _dev()
    B, C, H, W = 2, 5, 40, 56
    inp = t(synth.normal(91, "i", (B, C, H, W)), dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    tgt = t(synth.uniform(92, "t", (B, 2, H, W), 10.0, 16.0), dev)
    flow = torch.stack([tgt[:, 0] - xs, tgt[:, 1] - ys], 1).contiguous()
    met = t(synth.normal(93, "m", (B, 1, H, W)), dev)
    first = ts.FunctionSoftsplat(inp, flow, met, "softmax", deterministic=True)
    for _ in range(5):
        assert torch.equal(ts.FunctionSoftsplat(inp, flow, met, "softmax", deterministic=True), first)
    ref = ts.FunctionSoftsplat(inp, flow, met, "softmax")
    np.testing.assert_allclose(first.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)
    s1 = ts.FunctionSoftsplat(inp, flow, None, "summation", deterministic=True)
    np.testing.assert_allclose(s1.cpu().numpy(), ts.FunctionSoftsplat(inp, flow, None, "summation").cpu().numpy(), rtol=1e-4, atol=1e-4)
