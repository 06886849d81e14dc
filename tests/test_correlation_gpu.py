"""GPU: native correlation volumes (csrc/correlation.hip) vs the oracle restatement of the sampler's definition, plus
known-answer cases that need no oracle at all (the third-party sampler is absent: parity is unpinned by reference output)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def test_correlation1d_known_answers():
    import temporalstereo_amd as ts
    dev = _dev()
    B, C, H, W, D = 1, 8, 5, 23, 6
    left = torch.from_numpy(synth.normal(1, "l", (B, C, H, W))).to(dev)
    # right = left shifted by 2 px (right[x] = left[x+2]): the plane of disparity 2 (k = D-1-2) holds sum_c left^2 where defined
    right = torch.zeros_like(left)
    right[..., :W - 2] = left[..., 2:]
    out = ts.correlation1d(left, right, D)
    assert tuple(out.shape) == (B, D, H, W)
    k = D - 1 - 2
    want = (left * left).sum(1)
    np.testing.assert_allclose(out[:, k, :, 2:].cpu().numpy(), want[..., 2:].cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert float(out[:, k, :, :2].abs().max()) == 0.0            # x - 2 < 0: outside the image -> 0
    # plane D-1 is disparity 0: plain channel dot product (leaky_relu applied)
    dot = (left * right).sum(1)
    np.testing.assert_allclose(out[:, D - 1].cpu().numpy(), torch.where(dot > 0, dot, 0.1 * dot).cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", range(6))
def test_correlation_vs_oracle_forward_and_backward(case):
    import temporalstereo_amd as ts
    from oracle import correlation as oc
    dev = _dev()
    rng = np.random.RandomState(100 + case)
    B, C = int(rng.choice([1, 2])), int(rng.choice([3, 8, 32]))
    H, W = int(rng.randint(3, 20)), int(rng.randint(5, 70))
    l = torch.from_numpy(synth.normal(200 + case, "l", (B, C, H, W)))
    r = torch.from_numpy(synth.normal(200 + case, "r", (B, C, H, W)))
    if case % 2 == 0:
        D = int(rng.randint(1, 25))
        ours = lambda a, b: ts.correlation1d(a, b, D)
        ref = lambda a, b: oc.correlation1d(a, b, D)
    else:
        p = int(rng.choice([1, 3, 5, 9]))
        ours = lambda a, b: ts.correlation(a, b, p)
        ref = lambda a, b: oc.correlation(a, b, p)
    lr, rr = l.clone().requires_grad_(True), r.clone().requires_grad_(True)
    want = ref(lr, rr)
    g = torch.from_numpy(synth.normal(300 + case, "g", tuple(want.shape)))
    want.backward(g)
    lg, rg = l.to(dev).requires_grad_(True), r.to(dev).requires_grad_(True)
    got = ours(lg, rg)
    got.backward(g.to(dev))
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(rg.grad.cpu().numpy(), rr.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,C,H,W,D", [(1, 32, 6, 240, 48), (2, 64, 3, 120, 192), (1, 20, 4, 77, 17), (1, 8, 2, 64, 241), (1, 4, 3, 16, 1)])
def test_row_correlation_on_the_matrix_cores_vs_oracle(B, C, H, W, D):
    """The MFMA form of correlation1d (a band of the row's Gram matrix, csrc/correlation.hip) at sizes where every tile / strip /
    channel-chunk edge occurs: ragged widths, C not a multiple of the 32-channel chunk, D from 1 to the kernel's limit."""
    import temporalstereo_amd as ts
    from oracle import correlation as oc
    dev = _dev()
    l = torch.from_numpy(synth.normal(400 + D, "l", (B, C, H, W)))
    r = torch.from_numpy(synth.normal(400 + D, "r", (B, C, H, W)))
    want = oc.correlation1d(l.double(), r.double(), D)
    got = ts.correlation1d(l.to(dev), r.to(dev), D)
    scale = float(want.abs().max())
    assert float((got.cpu().double() - want).abs().max()) < 2e-6 * scale * (C ** 0.5)
