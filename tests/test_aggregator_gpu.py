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


@pytest.mark.parametrize("case", range(8))
def test_temporal_update_fuzz_vs_oracle(case):
    """The fused update (ts_reproject_memory_fwd) against the oracle's op-by-op restatement over odd sizes,
    larger motions (taps leaving the frame), absent memory / local maps, per-item baselines, 3x3 intrinsics,
    a pre-composed pose and local_map_size cropping."""
    import temporalstereo_amd as ts
    from oracle import temporal as otemporal
    dev = _dev()
    rng = np.random.RandomState(700 + case)
    B = int(rng.randint(1, 4))
    H, W = int(rng.randint(9, 40)) * 8 + int(rng.randint(0, 8)), int(rng.randint(12, 60)) * 8 + int(rng.randint(0, 8))
    h, w = H // 8, W // 8
    k = int(rng.randint(1, 4))
    n_in = int(rng.randint(0, 4))
    size = int(rng.randint(0, 5)) if case != 1 else 3
    with_memory = case != 2
    use_past = case != 3
    prev_disp = rng.uniform(2.0, 60.0, (B, 1, H, W)).astype(np.float32)
    base = rng.uniform(1.0, 12.0, (B, 1, h, w)).astype(np.float32)
    mem = {'disp_sample': (base + rng.uniform(-0.5, 0.5, (B, k, h, w))).astype(np.float32),
           'cost_volume': rng.randn(B, k, h, w).astype(np.float32)}
    lm = (base * rng.uniform(0.9, 1.1, (B, n_in, h, w))).astype(np.float32) if n_in else None
    K = synth.sceneflow_intrinsics(B, H, W)
    K[:, 0, 2] = W / 2.0 + rng.uniform(-3, 3, B); K[:, 1, 2] = H / 2.0 + rng.uniform(-3, 3, B)
    if case == 4:
        K = K[:, :3, :3].copy()
    T = synth.small_motion(700 + case, B)
    T[:, :3, 3] *= (1.0 if case % 2 else 6.0)                 # bigger translations: some pixels leave the frame
    T2 = np.linalg.inv(synth.small_motion(900 + case, B)).astype(np.float32)
    baseline = rng.uniform(0.3, 1.2, (B, 1, 1, 1)).astype(np.float32) if case in (5, 6) else 0.54

    def run(mod, to):
        info = {'prev_disp': to(prev_disp)}
        if with_memory:
            info['cost_memory'] = {kk: to(v) for kk, v in mem.items()}
        if lm is not None:
            info['local_map'] = to(lm)
        if case == 7:
            info['T_past_to_now'] = to(np.matmul(T, T2))
        bl = to(baseline) if isinstance(baseline, np.ndarray) else baseline
        return mod.update_map(info, to(K), to(T), to(T2), bl, H, W, use_past_cost=use_past, local_map_size=size)
    ref = run(otemporal, lambda a: t(a))
    got = run(ts.temporal, lambda a: t(a, dev))
    tag = "B=%d %dx%d k=%d n_in=%d size=%d mem=%s use=%s" % (B, H, W, k, n_in, size, with_memory, use_past)

    def close(a, b, what):
        a = a.cpu()
        assert a.shape == b.shape, (tag, what, a.shape, b.shape)
        d = (a - b).abs()
        # splat targets that receive almost no weight amplify rounding (x / (den + 1e-22)): robust criterion
        assert float(d.mean()) < 2e-4 and float((d > 1e-2 * (1 + b.abs())).float().mean()) < 2e-3, \
            (tag, what, float(d.mean()), float(d.max()))
    if ref.get('cost_memory') is None:
        assert got.get('cost_memory') is None, tag
    else:
        for key in ("disp_sample", "cost_volume"):
            close(got['cost_memory'][key], ref['cost_memory'][key], key)
    if size > 0:
        close(got['local_map'], ref['local_map'], "local_map")
        assert got['local_map_size'] == size
    else:
        assert ('local_map' in got) == ('local_map' in ref)
    assert got['use_past_cost'] == ref['use_past_cost']


@pytest.mark.parametrize("case", range(5))
def test_temporal_update_vs_reference_vectors(case):
    """The fused HIP update (ts_reproject_memory_fwd) against vectors made by the reference's own update_map."""
    import temporalstereo_amd as ts
    from helpers import temporal_update_from_golden, check_temporal_update
    g = load("temporal_update_%d" % case)
    info = temporal_update_from_golden(g, ts.temporal, _dev())
    check_temporal_update(g, info, 2e-4, 2e-3)
