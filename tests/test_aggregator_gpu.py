"""GPU: the whole aggregation pyramid (HIP cost volume / regression + the conv blocks) against the
outputs the REAL reference produced on the same seeded inputs (tests/golden/agg_*.npz), and the
temporal update against the oracle.

Tolerance: |dEPE| < 1e-3 px is the bar of BASELINE.json.  EPE-style metrics (mean |a-b|) are used
for the disparity maps because top-k / sort are discrete: a near-tie that flips between two fp32
implementations moves single pixels by O(1) without moving the mean.
"""
import numpy as np
import pytest
import torch

import synth
from helpers import load, t, dims_from_golden, synth_state, aggregator_inputs, epe

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _build(dims, seed, dev, train=False, golden=None):
    import temporalstereo_amd as ts
    net = ts.TEMPORALSTEREO(
        coarse=ts.CoarseAggregation(dims['coarse']['in_planes'], dims['coarse']['C'], dims['coarse']['num_sample']),
        fine=ts.FineAggregation(dims['fine']['in_planes'], dims['fine']['C'], 5),
        precise=ts.PreciseAggregation(dims['precise']['in_planes'], dims['precise']['C'], 5))
    net.load_state_dict(synth_state(dims, seed, golden=golden), strict=True)
    return net.to(dev).train(train)


def _check_against_golden(g, outs, tol_epe=1e-3):
    disps, costs, samples, offs, ranges, info = outs
    for i, nm in enumerate(("full", "precise", "fine_up", "coarse_up")):
        scale = g["W"] / disps[i].shape[-1]                       # EPE at full resolution
        assert epe(disps[i].cpu(), t(g["disp_" + nm])) * scale < tol_epe, nm
    for i, nm in enumerate(("precise", "fine", "coarse")):
        assert epe(costs[i].cpu(), t(g["cost_" + nm])) < 1e-3
        assert epe(samples[i].cpu(), t(g["samp_" + nm])) < 1e-3
        assert epe(offs[i].cpu(), t(g["off_" + nm])) < 1e-4
        assert costs[i].shape == g["cost_" + nm].shape
    assert epe(info["cost_memory"]["disp_sample"].cpu(), t(g["mem_out_disp_sample"])) < 1e-3
    assert epe(info["cost_memory"]["cost_volume"].cpu(), t(g["mem_out_cost_volume"])) < 1e-3
    assert epe(info["prev_disp"][:, :, ::4, ::4].cpu(), t(g["prev_disp_sub"])) < tol_epe


@pytest.mark.parametrize("name", ["agg_tiny_single", "agg_tiny_temporal"])
def test_aggregator_eval_matches_reference(name):
    g = load(name); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    with torch.no_grad():
        outs = net(lf, rf, il, ir, prev)
    _check_against_golden(g, outs)


def test_aggregator_train_mode_matches_reference_and_backprops():
    """BatchNorm on batch statistics (train()), gradients flow to features and weights."""
    g = load("agg_tiny_train"); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, train=True, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    lf = [x.requires_grad_() for x in lf]
    outs = net(lf, rf, il, ir, prev)
    _check_against_golden(g, [[d.detach() for d in outs[0]], [c.detach() for c in outs[1]],
                              [s.detach() for s in outs[2]], [o.detach() for o in outs[3]], outs[4], outs[5]])
    loss = sum(d.mean() for d in outs[0]) + sum(c.square().mean() for c in outs[1])
    loss.backward()
    assert all(x.grad is not None and torch.isfinite(x.grad).all() for x in lf)
    gw = net.coarse.init3d[0].conv[0].weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0


def test_aggregator_config1_256x512():
    """BASELINE.json configs[0] (256x512, D=48) with the sceneflow channel dims."""
    g = load("agg_config1_256x512"); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    with torch.no_grad():
        disps, costs, samples, offs, ranges, info = net(lf, rf, il, ir, prev)
    assert epe(disps[0][:, :, ::4, ::4].cpu(), t(g["disp_full_sub4"])) < 1e-3
    assert epe(disps[1].cpu(), t(g["disp_precise"])) * 4 < 1e-3
    assert epe(disps[3].cpu(), t(g["disp_coarse_up"])) * 8 < 1e-3
    assert abs(float(disps[0].double().mean()) - float(g["disp_full_mean"])) < 1e-3


def test_temporal_update_vs_oracle():
    """update_map (re-projection + softmax splat) vs the oracle restatement, config-3 geometry."""
    import temporalstereo_amd as ts
    from oracle import temporal as otemporal
    dev = _dev()
    B, H, W = 2, 544, 960
    h, w = H // 8, W // 8
    seed = synth.SEED0 + 3
    prev_disp = synth.uniform(seed, "pd", (B, 1, H, W), 5.0, 150.0)
    base = synth.uniform(seed, "mb", (B, 1, h, w), 1.0, 18.0)
    mem = {'disp_sample': np.concatenate([base + 0.3, base - 0.4], 1).astype(np.float32),
           'cost_volume': synth.normal(seed, "mc", (B, 2, h, w))}
    lm = np.concatenate([base * 1.02, base * 0.97 + 0.2], 1).astype(np.float32)
    K = synth.sceneflow_intrinsics(B, H, W)
    T = synth.small_motion(seed, B)
    eye = np.broadcast_to(np.eye(4, dtype=np.float32), (B, 4, 4)).copy()

    def run(mod, to):
        info = {'prev_disp': to(prev_disp), 'cost_memory': {k: to(v) for k, v in mem.items()}, 'local_map': to(lm)}
        return mod.update_map(info, to(K), to(T), to(eye), 1.0, H, W, use_past_cost=True, local_map_size=3)
    ref = run(otemporal, lambda a: t(a))
    got = run(ts.temporal, lambda a: t(a, dev))
    for key in ("disp_sample", "cost_volume"):
        a, b = got['cost_memory'][key].cpu(), ref['cost_memory'][key]
        assert a.shape == b.shape
        assert epe(a, b) < 1e-4 and float((a - b).abs().max()) < 5e-2
    assert got['local_map'].shape == ref['local_map'].shape == (B, 3, h, w)
    assert epe(got['local_map'].cpu(), ref['local_map']) < 1e-4
