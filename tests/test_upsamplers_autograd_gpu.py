"""GPU: forward + backward of the two softmax-weighted upsamplers (ts_convex_upsample_{fwd,bwd}, ts_unet_upsample_{fwd,bwd}) against
the reference's torch formulation (module.py:337-353 and :468-482, restated in aggregation/blocks.py) evaluated in fp64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth

pytestmark = pytest.mark.gpu


def _convex_ref(logits, disp, r, scale):
    B, C, H, W = disp.shape
    w = torch.softmax(logits.view(B, 1, 9, r, r, H, W), dim=2)
    nb = F.unfold(disp * scale, kernel_size=(3, 3), padding=(1, 1)).view(B, C, 9, 1, 1, H, W)
    return torch.sum(w * nb, dim=2).permute(0, 1, 4, 2, 5, 3).contiguous().reshape(B, C, H * r, W * r)


def _unet_ref(mask, disp):
    mask = F.softmax(mask, dim=1)
    b, _, h, w = mask.shape
    dh, dw = disp.shape[-2:]
    nb = F.unfold(disp, kernel_size=(3, 3), padding=(1, 1)).reshape(b, 9, dh, dw)
    full = F.interpolate(nb * w / dw, size=(h, w), mode='bilinear', align_corners=True)
    return torch.sum(full * mask, dim=1, keepdim=True)


@pytest.mark.parametrize("case", range(4))
def test_convex_upsample_autograd(case):
    from temporalstereo_amd import functional as TF
    dev = torch.device("cuda:0")
    B, H, W, r = [(1, 9, 14, 2), (2, 17, 30, 2), (2, 5, 7, 4), (8, 30, 40, 2)][case]
    logits = torch.from_numpy(synth.normal(500 + case, "m", (B, 9 * r * r, H, W), 2.0)).to(dev)
    disp = torch.from_numpy(synth.uniform(500 + case, "d", (B, 1, H, W), 0.0, 30.0)).to(dev)
    g = torch.from_numpy(synth.normal(500 + case, "g", (B, 1, H * r, W * r))).to(dev)
    la, da = logits.clone().requires_grad_(True), disp.clone().requires_grad_(True)
    ya = TF.convex_upsample(la, da, r, float(r))
    ya.backward(g)
    lb, db = logits.double().requires_grad_(True), disp.double().requires_grad_(True)
    yb = _convex_ref(lb, db, r, float(r))
    yb.backward(g.double())
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(la.grad.cpu().numpy(), lb.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(da.grad.cpu().numpy(), db.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("case", range(4))
def test_unet_upsample_autograd(case):
    from temporalstereo_amd import functional as TF
    dev = torch.device("cuda:0")
    B, h, w, f = [(1, 8, 12, 4), (2, 17, 30, 4), (2, 9, 13, 2), (4, 34, 60, 4)][case]
    Ho, Wo = h * f, w * f
    mask = torch.from_numpy(synth.normal(600 + case, "m", (B, 9, Ho, Wo), 2.0)).to(dev)
    disp = torch.from_numpy(synth.uniform(600 + case, "d", (B, 1, h, w), 0.0, 40.0)).to(dev)
    g = torch.from_numpy(synth.normal(600 + case, "g", (B, 1, Ho, Wo))).to(dev)
    ma, da = mask.clone().requires_grad_(True), disp.clone().requires_grad_(True)
    ya = TF.unet_upsample(ma, da)
    ya.backward(g)
    mb, db = mask.double().requires_grad_(True), disp.double().requires_grad_(True)
    yb = _unet_ref(mb, db)
    yb.backward(g.double())
    np.testing.assert_allclose(ya.detach().cpu().numpy(), yb.detach().cpu().numpy(), rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(ma.grad.cpu().numpy(), mb.grad.cpu().numpy(), rtol=1e-4, atol=2e-3)      # g * p * (bil - out): a difference of values ~160 in fp32
    np.testing.assert_allclose(da.grad.cpu().numpy(), db.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
