"""CPU: analytic known-answer tests that pin the splat oracle (the reference's cupy op cannot run on
CPU: softsplat.py:252,269 -- 'parity unpinned by the reference')."""
import numpy as np
import pytest
import torch

from oracle import splat_sum, softsplat


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def test_zero_flow_is_identity():
    x = _rand(2, 3, 5, 7)
    assert torch.equal(splat_sum(x, torch.zeros(2, 2, 5, 7, dtype=torch.float64)), x)


def test_integer_flow_is_exact_shift_with_mass_dropped():
    x = _rand(1, 2, 6, 8, seed=1)
    flow = torch.zeros(1, 2, 6, 8, dtype=torch.float64)
    flow[:, 0] = 3.0
    flow[:, 1] = -2.0
    out = splat_sum(x, flow)
    exp = torch.zeros_like(x)
    exp[:, :, 0:4, 3:8] = x[:, :, 2:6, 0:5]
    assert torch.equal(out, exp)


def test_fractional_flow_conserves_mass_inside_frame():
    x = _rand(1, 1, 9, 9, seed=2).abs()
    flow = torch.zeros(1, 2, 9, 9, dtype=torch.float64)
    flow[:, 0] = 0.3
    flow[:, 1] = 0.6
    x[:, :, -1, :] = 0          # nothing leaves the frame
    x[:, :, :, -1] = 0
    out = splat_sum(x, flow)
    assert abs(float(out.sum() - x.sum())) < 1e-12


def test_four_weights():
    x = torch.zeros(1, 1, 4, 4, dtype=torch.float64)
    x[0, 0, 1, 1] = 1.0
    flow = torch.zeros(1, 2, 4, 4, dtype=torch.float64)
    flow[0, 0, 1, 1], flow[0, 1, 1, 1] = 0.25, 0.5
    out = splat_sum(x, flow)[0, 0]
    np.testing.assert_allclose(out[1, 1], 0.75 * 0.5)
    np.testing.assert_allclose(out[1, 2], 0.25 * 0.5)
    np.testing.assert_allclose(out[2, 1], 0.75 * 0.5)
    np.testing.assert_allclose(out[2, 2], 0.25 * 0.5)


def test_softmax_constant_metric_equals_average():
    x = _rand(2, 3, 6, 6, seed=3)
    flow = _rand(2, 2, 6, 6, seed=4) * 1.5
    m = torch.full((2, 1, 6, 6), 0.7, dtype=torch.float64)
    a = softsplat(x, flow, m, 'softmax')
    b = softsplat(x, flow, None, 'average')
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-10, atol=1e-12)


def test_linear_mode_and_bad_args():
    x = _rand(1, 2, 5, 5, seed=5)
    flow = _rand(1, 2, 5, 5, seed=6)
    m = _rand(1, 1, 5, 5, seed=7).abs() + 0.1
    lin = softsplat(x, flow, m, 'linear')
    num = splat_sum(x * m, flow)
    den = splat_sum(m, flow)
    np.testing.assert_allclose(lin.numpy(), (num / (den + 1e-22)).numpy(), rtol=1e-12)
    with pytest.raises(ValueError):
        softsplat(x, flow, m, 'bogus')
    with pytest.raises(ValueError):
        softsplat(x, flow, torch.cat([m, m], 1), 'linear')


def test_gradcheck_float64():
    """The backward kernels of the reference (softsplat.py:63-105, :116-176) are the adjoints of the
    forward; autograd through the oracle restatement gives them, and gradcheck validates those."""
    x = _rand(1, 2, 4, 5, seed=8).requires_grad_()
    flow = (_rand(1, 2, 4, 5, seed=9) * 0.8 + 0.13).requires_grad_()    # away from integer kinks
    assert torch.autograd.gradcheck(splat_sum, (x, flow), eps=1e-6, atol=1e-6)
