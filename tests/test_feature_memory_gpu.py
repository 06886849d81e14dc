"""temporal.exchange_feature_memory (ts_channel_splice_fwd) -- the memory plumbing in front of the reference backbone's residual
blocks (architecture/modeling/backbone/TemporalStereo.py:183-197, :218) -- against the reference's
recorded outputs and the oracle, values and gradients."""
import numpy as np
import pytest
import torch

from helpers import load, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["feature_memory_%d" % i for i in range(4)])
def test_feature_memory_golden(name):
    from temporalstereo_amd import temporal
    dev = torch.device("cuda:0")
    g = load(name)
    inp = t(g["input"], dev)
    mem = t(g["memory"], dev) if int(g["has_memory"]) else None
    x, new_mem = temporal.exchange_feature_memory(inp, mem, float(g["memory_percent"]))
    assert torch.equal(x.cpu(), t(g["out"]) - t(g["input"]))
    assert torch.equal(new_mem.cpu(), t(g["new_memory"]))


def test_feature_memory_gradients_and_errors():
    from oracle import backbone_memory as obm
    from temporalstereo_amd import temporal
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(7)
    for B, C, H, W, pct in ((2, 32, 17, 30, 0.25), (1, 24, 8, 9, 0.5), (3, 16, 5, 5, 0.0625)):
        a = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32))
        m = torch.from_numpy(rng.randn(B, int(C * pct), H, W).astype(np.float32))
        ga = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32))
        gm = torch.from_numpy(rng.randn(B, int(C * pct), H, W).astype(np.float32))
        a1, m1 = a.clone().requires_grad_(True), m.clone().requires_grad_(True)
        a2, m2 = a.to(dev).requires_grad_(True), m.to(dev).requires_grad_(True)
        x1, n1 = obm.exchange(a1, m1, pct)
        x2, n2 = temporal.exchange_feature_memory(a2, m2, pct)
        ((x1 * ga).sum() + (n1 * gm).sum()).backward()
        ((x2 * ga.to(dev)).sum() + (n2 * gm.to(dev)).sum()).backward()
        assert torch.equal(x2.cpu(), x1.detach()) and torch.equal(n2.cpu(), n1.detach())
        assert torch.equal(a2.grad.cpu(), a1.grad) and torch.equal(m2.grad.cpu(), m1.grad)
    with pytest.raises(AssertionError, match="memory shape"):
        temporal.exchange_feature_memory(torch.zeros(1, 16, 4, 4, device=dev), torch.zeros(1, 3, 4, 4, device=dev), 0.25)
    first, mem0 = temporal.exchange_feature_memory(torch.ones(1, 16, 4, 4, device=dev), None, 0.25)       # first frame: nothing to splice
    assert first.shape == (1, 16, 4, 4) and mem0.shape == (1, 4, 4, 4)
