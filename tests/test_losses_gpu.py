"""GPU: the loss-side kernels (csrc/losses.hip) against vectors recorded from the reference's own loss objects
(WarssersteinDistanceLoss, DispSmoothL1Loss behind the wrapper's full-resolution rescale) -- values and gradients -- and against
the oracle at the sizes of BASELINE configs[1]/[2]."""
import numpy as np
import pytest
import torch

from helpers import load, t

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["loss_dense", "loss_sparse", "loss_none_valid", "loss_ragged"])
def test_losses_match_reference_values_and_gradients(name):
    from temporalstereo_amd import losses as TL
    dev = _dev()
    g = load(name)
    gt = t(g["gt"], dev)
    md, sparse = float(g["max_disp"]), bool(int(g["sparse"]))
    H, W = gt.shape[-2:]
    nl, ne = int(g["n_levels"]), int(g["n_est"])
    costs = [t(g["cost%d" % i], dev).requires_grad_(True) for i in range(nl)]
    offs = [t(g["off%d" % i], dev).requires_grad_(True) for i in range(nl)]
    samps = [t(g["sample%d" % i], dev).requires_grad_(True) for i in range(nl)]
    ests = [t(g["est%d" % i], dev).requires_grad_(True) for i in range(ne)]
    wd = TL.WarssersteinDistanceLoss(max_disp=md, sparse=sparse)(costs, offs, samps, gt)
    sd = TL.DispSmoothL1Loss(max_disp=md, sparse=sparse, rescale=True)(ests, gt)
    assert sorted(wd) == ["wars_loss_lvl%d" % i for i in range(nl)] and sorted(sd) == ["l1_loss_lvl%d" % i for i in range(ne)]
    (sum(wd.values()) + sum(sd.values())).backward()
    for i in range(nl):
        np.testing.assert_allclose(float(wd["wars_loss_lvl%d" % i]), float(g["wars_loss%d" % i]), rtol=2e-6, atol=1e-7)
        for k, v in (("g_cost", costs[i]), ("g_off", offs[i]), ("g_sample", samps[i])):
            np.testing.assert_allclose(v.grad.cpu().numpy(), g["%s%d" % (k, i)], rtol=2e-5, atol=1e-8, err_msg="%s level %d" % (k, i))
    for i in range(ne):
        np.testing.assert_allclose(float(sd["l1_loss_lvl%d" % i]), float(g["l1_loss%d" % i]), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(ests[i].grad.cpu().numpy(), g["g_est%d" % i], rtol=2e-5, atol=1e-9, err_msg="g_est %d" % i)
        np.testing.assert_allclose(TL.rescale_to_full(ests[i].detach(), (H, W)).cpu().numpy(), g["full%d" % i], rtol=1e-6, atol=2e-6)
    # the reference's contract (already full-resolution disparities) gives the same values
    full = [t(g["full%d" % i], dev) for i in range(ne)]
    sd2 = TL.DispSmoothL1Loss(max_disp=md, sparse=sparse)(full, gt)
    for i in range(ne):
        np.testing.assert_allclose(float(sd2["l1_loss_lvl%d" % i]), float(g["l1_loss%d" % i]), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("B", [1, 4])
def test_losses_at_config_sizes_vs_oracle(B):
    """544x960 ground truth, the three cost levels of configs[1]/[2] (D = 5 / 7 / 14) and the four disparities."""
    import synth
    from oracle import losses as ol
    from temporalstereo_amd import losses as TL
    dev = _dev()
    H, W, md = 544, 960, 192
    seed = synth.SEED0 + 400 + B
    gt = torch.from_numpy(synth.uniform(seed, "gt", (B, 1, H, W), -5.0, 230.0))
    for li, (s, D) in enumerate(((4, 5), (8, 7), (16, 14))):
        h, w = H // s, W // s
        c = torch.from_numpy(synth.normal(seed, "c%d" % li, (B, D, h, w), 3.0))
        o = torch.from_numpy(synth.normal(seed, "o%d" % li, (B, D, h, w), 0.3))
        sm = torch.from_numpy(synth.uniform(seed, "s%d" % li, (B, D, h, w), 0.0, md / s))
        ref_in = [x.clone().requires_grad_(True) for x in (c, o, sm)]
        ref = ol.wasserstein_loss_per_level(*ref_in, gt, md, 0, False)
        ref.backward()
        got_in = [x.to(dev).requires_grad_(True) for x in (c, o, sm)]
        got = TL.wasserstein_loss_per_level(*got_in, gt.to(dev), md, 0, False)
        got.backward()
        np.testing.assert_allclose(float(got), float(ref), rtol=5e-6)
        for a, b in zip(got_in, ref_in):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=5e-5, atol=1e-9)
        e = torch.from_numpy(synth.uniform(seed, "e%d" % li, (B, 1, h, w), 0.0, md / s))
        er = e.clone().requires_grad_(True)
        ref = ol.smooth_l1_loss_per_level(ol.rescale_to_full(er, (H, W)), gt, md, 0)
        ref.backward()
        eg = e.to(dev).requires_grad_(True)
        got = TL.smooth_l1_loss_per_level(eg, gt.to(dev), md, 0)
        got.backward()
        np.testing.assert_allclose(float(got), float(ref), rtol=5e-6)
        np.testing.assert_allclose(eg.grad.cpu().numpy(), er.grad.numpy(), rtol=5e-5, atol=1e-9)
