"""GPU: the COMPOSED backward of the whole aggregator against the reference's autograd.

tests/golden/planted_train_grads.npz (tools/gen_golden.py planted_gradient_case): the reference aggregator in train() mode with
the committed checkpoint on a planted scene, the reference's own loss objects with the sceneflow.yaml weights, loss.backward().
Here: the product's modules (HIP forward + backward: cost volume, conv -> BatchNorm(train) -> activation nodes, resize, pooling,
sort + gather, upsamplers, both fused losses) on the same inputs; every loss term, the gradients of the six feature maps, 41
named weight gradients element by element, and the norm + a seeded projection of ALL 273 parameter gradients.  (The fixture also
holds the reference's float64 backward, 'f64::*': its float32 backward is within 5e-6 of it everywhere, so float32 is the arbiter.)"""
import os

import numpy as np
import pytest
import torch

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
pytestmark = pytest.mark.gpu

import parity_tools as PT  # noqa: E402
import synth  # noqa: E402
from helpers import load  # noqa: E402


SOFT = bool(int(os.environ.get("TS_PARITY_SOFT", "0")))          # collect the report without asserting (exploration)


def _rel(a, b):
    """-> (max |a-b| / max |b|,  ||a-b|| / ||b||,  fraction of elements further than 1e-3 max |b| apart)."""
    a, b = a.double().cpu(), b.double().cpu()
    d, m = (a - b).abs(), b.abs().max().clamp_min(1e-30)
    return float(d.max() / m), float(d.norm() / b.norm().clamp_min(1e-30)), float((d > 1e-3 * m).double().mean())


def _check(cond, msg):
    if not SOFT:
        assert cond, msg


@pytest.mark.parametrize("fixture", ["planted_train_grads", "planted_train_grads_c1"])
def test_whole_aggregator_gradients_match_the_reference_autograd(fixture):
    """planted_train_grads: 128x192, B=2, element by element; planted_train_grads_c1 (round 5): BASELINE configs[1]'s own size,
    544x960, D=192, B=1 -- the reference's train-mode forward + backward run there on the CPU, stored as loss terms, norms and
    seeded projections of every gradient (tools/gen_golden.py --only-planted-grads-c1)."""
    import bench
    from temporalstereo_amd.losses import DispSmoothL1Loss, WarssersteinDistanceLoss
    g = load(fixture)
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
            r = abs(float(v.detach()) - float(g["loss::" + k])) / max(abs(float(g["loss::" + k])), 1e-12)
            rep.add(what="loss term", key=k, ours=float(v.detach()), reference=float(g["loss::" + k]), rel=r)
            _check(r < 1e-5, (k, float(v.detach()), float(g["loss::" + k])))
        # Element by element.  A discrete step of the forward pass (top-2 selection, candidate order) that falls the other way at a
        # near-tie would change the gradient of the few pixels behind it completely, in any two fp32 implementations; so the bar is
        # stated robustly: almost every element within 1e-3 of the tensor's scale, and the tensor as a whole within 1e-3 in the L2
        # sense (measured: 1e-5 -- the trained network has no such near-ties on this scene).
        for i in range(3):
            for side, ts in (("left", lf), ("right", rf)):
                gr = ts[i].grad
                n_ref = float(g["g_%s_%d_norm" % (side, i)])
                rn = abs(float(gr.double().norm()) - n_ref) / n_ref
                if "g_%s_%d" % (side, i) in g:
                    rmax, rl2, bad = _rel(gr[:, ::8], torch.from_numpy(g["g_%s_%d" % (side, i)]))
                    rep.add(what="feature gradient", key="%s_%d" % (side, i), rel_max=rmax, rel_l2=rl2, frac_off=bad, rel_norm=rn)
                    _check(rl2 < 1e-3 and bad < 1e-3 and rn < 1e-3, ("feature gradient", side, i, rmax, rl2, bad, rn))
                else:       # full size: norm and a seeded projection (scale of a projection = the gradient's norm)
                    pr = torch.from_numpy(synth.normal(seed, "projf%s%d" % (side, i), tuple(gr.shape))).double()
                    rp = abs(float((gr.double().cpu() * pr).sum()) - float(g["g_%s_%d_proj" % (side, i)])) / n_ref
                    rep.add(what="feature gradient (norm / projection)", key="%s_%d" % (side, i), rel_norm=rn, rel_proj=rp)
                    _check(rn < 1e-3 and rp < 1e-3, ("feature gradient", side, i, rn, rp))
        named = dict(net.named_parameters())
        top = float(g["f64::all_norm"].max())
        exactly_zero = {str(k) for k, n64 in zip(g["all_keys"], g["f64::all_norm"]) if n64 < 1e-10 * top}     # decided in float64
        for k in [str(x) for x in g["picked"]]:
            rmax, rl2, bad = _rel(named[k].grad, torch.from_numpy(g["gw::" + k]))
            rep.add(what="weight gradient", key=k, rel_max=rmax, rel_l2=rl2, frac_off=bad)
            if k in exactly_zero:        # a bias in front of BatchNorm: rounding noise on both sides
                _check(float(named[k].grad.norm()) < 1e-5 * top, ("weight gradient that is exactly zero", k, float(named[k].grad.norm())))
            else:
                _check(rl2 < 1e-3 and rmax < 2e-3, ("weight gradient", k, rmax, rl2, bad))
        keys = [str(x) for x in g["all_keys"]]
        assert sorted(k for k in named if named[k].grad is not None) == keys          # same set of parameters receives a gradient
        assert [str(x) for x in g["no_grad_keys"]] == sorted(k for k in named if named[k].grad is None)
        # (weights: a sum over all pixels, so a flipped pixel is diluted; 1 % of the gradient's norm in norm and projection)
        # At 544x960 a gradient is a sum over half a million pixels and the reference's OWN float32 backward sits up to 1.1e-3 (of the
        # gradient's norm) from its float64 one (precise.init3d.1.conv4.conv.0.norm.bias; 1.0e-3 on precise.refinement.conv4.0.weight):
        # there the arbiter is the float64 run and a key's bar is max(1e-3, 3 x the reference's own float32 deviation on that key).
        full_size = fixture.endswith("_c1")
        for i, (k, n_ref, p_ref) in enumerate(zip(keys, g["all_norm"], g["all_proj"])):
            gr = named[k].grad.double().cpu()
            n = float(gr.norm())
            p = float((gr.flatten() * torch.from_numpy(synth.normal(seed, "proj" + k, (gr.numel(),))).double()).sum())
            bar = 1e-3
            if full_size:
                n64, p64 = float(g["f64::all_norm"][i]), float(g["f64::all_proj"][i])
                own = max(abs(float(n_ref) - n64), abs(float(p_ref) - p64)) / max(n64, 1e-12)
                # (measured worst case of the product path: 1.05e-3, the projection of precise.refinement.conv2.0.norm.weight with the
                # split first layer -- 0.9e-3 without it; the reference's own float32 run reaches 1.1e-3 on another key)
                bar, n_ref, p_ref = max(2e-3, 3.0 * own), n64, p64
            rn = abs(n - n_ref) / max(n_ref, 1e-12)
            rp = abs(p - p_ref) / max(n_ref, 1e-12)             # a projection's scale is the gradient's norm
            rep.add(what="every parameter: gradient norm / projection", key=k, rel_norm=rn, rel_proj=rp, bar=bar)
            if k in exactly_zero:
                _check(n < 1e-5 * top, (k, n))
            else:
                _check(rn < bar and rp < bar, (k, n, float(n_ref), p, float(p_ref), bar))
    finally:
        rep.dump("parity_train_grads.json" if fixture == "planted_train_grads" else "parity_train_grads_c1.json")
