"""GPU: the COMPOSED backward of the whole aggregator against the reference's autograd.

tests/golden/planted_train_grads.npz (tools/gen_golden.py planted_gradient_case): the reference aggregator in train() mode with
the committed checkpoint on a planted scene, the reference's own loss objects with the sceneflow.yaml weights, loss.backward().
Here: the product's modules (HIP forward + backward: cost volume, conv -> BatchNorm(train) -> activation nodes, resize, pooling,
sort + gather, upsamplers, both fused losses) on the same inputs; every loss term, the gradients of the six feature maps, 41
named weight gradients element by element, and the norm + a seeded projection of ALL 273 parameter gradients."""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
pytestmark = pytest.mark.gpu

import parity_tools as PT  # noqa: E402
import synth  # noqa: E402
from helpers import load  # noqa: E402


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_whole_aggregator_gradients_match_the_reference_autograd():
    import bench
    from temporalstereo_amd.losses import DispSmoothL1Loss, WarssersteinDistanceLoss
    g = load("planted_train_grads")
    dev = torch.device("cuda:0")
    B, H, W, ns, max_disp, seed = (int(g[k]) for k in ("B", "H", "W", "num_sample", "max_disp", "seed"))
    net = bench.build_model(dev, seed, ns)
    net.load_state_dict(PT.load_checkpoint(), strict=True)
    net.train()
    sc = synth.stereo_sequence(seed, B, H, W, frames=1, max_disp=max_disp, fx=float(g["fx"]), baseline=1.0)
    lf, rf, il, ir = sc["frames"][0]
    cs = synth.checksum([lf, rf, il, ir])
    assert abs(cs - float(g["input_checksum"])) <= 1e-9 * abs(cs), "scene generator does not reproduce the fixture's inputs"
    T = lambda a: torch.from_numpy(a).to(dev)
    lf = [T(x).requires_grad_(True) for x in lf]
    rf = [T(x).requires_grad_(True) for x in rf]
    gt = T(sc["gt"][0])
    disps, costs, samples, offs, _, _ = net(lf, rf, T(il), T(ir), {})
    l1 = DispSmoothL1Loss(max_disp=max_disp, rescale=True, weights=[2.0, 1.0, 0.7, 0.5])(disps, gt)
    wd = WarssersteinDistanceLoss(max_disp=max_disp, global_weight=2.0, weights=[1.0, 0.7, 0.5])(costs, offs, samples, gt)
    total = sum(l1.values()) + sum(wd.values())
    total.backward()
    rep = PT.Report()
    try:
        for k, v in list(l1.items()) + list(wd.items()):
            r = abs(float(v) - float(g["loss::" + k])) / max(abs(float(g["loss::" + k])), 1e-12)
            rep.add(what="loss term", key=k, ours=float(v), reference=float(g["loss::" + k]), rel=r)
            assert r < 1e-4, (k, float(v), float(g["loss::" + k]))
        worst = 0.0
        for i in range(3):
            for side, ts in (("left", lf), ("right", rf)):
                gr = ts[i].grad
                r = _rel(gr[:, ::4], torch.from_numpy(g["g_%s_%d" % (side, i)]))
                rn = abs(float(gr.double().norm()) - float(g["g_%s_%d_norm" % (side, i)])) / float(g["g_%s_%d_norm" % (side, i)])
                rep.add(what="feature gradient", key="%s_%d" % (side, i), rel_max=r, rel_norm=rn)
                worst = max(worst, r)
                assert r < 2e-3 and rn < 1e-3, ("feature gradient", side, i, r, rn)
        named = dict(net.named_parameters())
        for k in [str(x) for x in g["picked"]]:
            r = _rel(named[k].grad, torch.from_numpy(g["gw::" + k]))
            rep.add(what="weight gradient", key=k, rel_max=r)
            assert r < 2e-3, ("weight gradient", k, r)
        keys = [str(x) for x in g["all_keys"]]
        assert sorted(k for k in named if named[k].grad is not None) == keys          # same set of parameters receives a gradient
        assert [str(x) for x in g["no_grad_keys"]] == sorted(k for k in named if named[k].grad is None)
        for k, n_ref, p_ref in zip(keys, g["all_norm"], g["all_proj"]):
            gr = named[k].grad.double().cpu()
            n = float(gr.norm())
            p = float((gr.flatten() * torch.from_numpy(synth.normal(seed, "proj" + k, (gr.numel(),))).double()).sum())
            rn = abs(n - n_ref) / max(n_ref, 1e-12)
            rp = abs(p - p_ref) / max(n_ref, 1e-12)             # a projection's scale is the gradient's norm
            rep.add(what="every parameter: gradient norm / projection", key=k, rel_norm=rn, rel_proj=rp)
            assert rn < 2e-3 and rp < 2e-3, (k, n, float(n_ref), p, float(p_ref))
    finally:
        rep.dump("parity_train_grads.json")
