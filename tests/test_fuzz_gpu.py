"""GPU: seeded random-shape sweep of the cost-volume kernels (block_cost int / sampled / warped, cat_fms, dif_fms)
against the oracle: batch 1-3, 8-128 channels, ragged and aligned widths up to 320, 2-12 candidates, 1-3 scales.
This is the kind of sweep that exposed the store-data hazard of DESIGN.md section 7 (one channel, element .x, only
for batch >= 2 with 7-8 candidates).  Each op is run twice: the hazards seen so far were timing dependent.

dif_fms thresholds the interpolated value at > 0 (dif_fms.py:40): an element whose warped value is ~1e-7 may fall
on the other side in two fp32 implementations and then holds the fill value instead of the difference.  Those
(measure-zero) elements are excluded; everything else must agree to 1e-4 absolute + 2e-5 relative (SURVEY.md Appendix
B.1; the kernels reproduce the reference's coordinate normalise / un-normalise float sequence, so tap positions round
identically and what is left is summation order)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


def test_cost_volume_kernels_random_shapes():
    import oracle
    import oracle.cost_volume as ocv
    import temporalstereo_amd as ts
    from temporalstereo_amd import functional as TF
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(20260928)
    for it in range(28):
        B = int(rng.choice([1, 2, 3])); C = int(rng.choice([8, 16, 32, 64, 128])); H = int(rng.randint(4, 48))
        W = int(rng.choice([rng.randint(4, 130), 4 * rng.randint(1, 80)]))
        D = int(rng.randint(2, 13)); sc = int(rng.choice([1, 2, 3]))
        l = torch.from_numpy(synth.normal(500 + it, "l", (B, C, H, W))); r = torch.from_numpy(synth.normal(500 + it, "r", (B, C, H, W)))
        d = torch.from_numpy(synth.uniform(500 + it, "d", (B, D, H, W), -3.0, W * 0.6))
        lg, rg, dg = l.to(dev), r.to(dev), d.to(dev)
        tag = "case %d %s" % (it, (B, C, H, W, D, sc))
        exp = oracle.block_cost(l, r, d, sc)
        for _ in range(2):
            np.testing.assert_allclose(ts.block_cost(lg, rg, dg, sc).cpu().numpy(), exp.numpy(), rtol=2e-5, atol=1e-4, err_msg=tag + " sampled")
            np.testing.assert_allclose(TF.block_cost_warped(lg, rg, dg, sc).cpu().numpy(), exp[:, C:].numpy(), rtol=2e-5, atol=1e-4, err_msg=tag + " warped")
        np.testing.assert_allclose(ts.block_cost(lg, rg, D, sc).cpu().numpy(), oracle.block_cost(l, r, D, sc).numpy(), rtol=2e-5, atol=1e-4, err_msg=tag + " int")
        np.testing.assert_allclose(ts.cat_fms(lg, rg, dg).cpu().numpy(), ocv.cat_fms(l, r, d).numpy(), rtol=2e-5, atol=1e-4, err_msg=tag + " cat")
        tgt = ocv.warp_candidates(r, d)
        got, want = ts.dif_fms(lg, rg, dg).cpu(), ocv.dif_fms(l, r, d)
        off = ((got - want).abs() > 1e-4 + 2e-5 * want.abs()) & (tgt.abs() > 1e-5)
        assert int(off.sum()) == 0, tag + " dif: %d elements differ away from the threshold" % int(off.sum())
        assert int((((got - want).abs() > 1e-4 + 2e-5 * want.abs())).sum()) <= 4, tag + " dif: too many threshold flips"


def _rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


def test_convolution_kernels_random_shapes():
    """Forward (folded scale/shift/activation form), backward-data and backward-weight of both families on random
    geometries -- ragged channel counts, 1-pixel images, odd sizes under stride 2, every K-chunk cap -- against the
    framework's conv3d / conv_transpose3d autograd."""
    import torch.nn.functional as F
    from temporalstereo_amd import _lib
    from temporalstereo_amd import functional as TF
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(77)
    try:
        for it in range(36):
            cap = int(rng.choice([8, 16, 32]))
            _lib.check(_lib.lib().ts_conv_set_chunk_cap(cap), "cap")
            B = int(rng.choice([1, 2])); cin = int(rng.choice([1, 3, 8, 13, 32, 70, 176])); cout = int(rng.choice([1, 2, 8, 24, 32, 64]))
            D = int(rng.randint(1, 8)); H = int(rng.randint(1, 40)); W = int(rng.randint(1, 75))
            fam = rng.choice(["hw", "d", "hwT", "dT"])
            x = torch.from_numpy(synth.normal(900 + it, "x", (B, cin, D, H, W))).to(dev)
            if fam == "hw":
                s, dl = [(1, 1), (2, 1), (1, 2)][rng.randint(3)]
                w = torch.from_numpy(synth.normal(900 + it, "w", (cout, cin, 1, 3, 3), 0.2)).to(dev)
                args = ((1, s, s), (0, dl, dl), (1, dl, dl))
                ours = lambda a, b: TF.conv3d(a, b, None, *args); ref = lambda a, b: F.conv3d(a, b, None, *args)
            elif fam == "d":
                k = int(rng.choice([1, 3, 5])); s = int(rng.choice([1, 2])) if k == 3 else 1
                dl = 1 if s == 2 else int(rng.choice([1, 2])); pad = 1 if s == 2 else int(rng.choice([0, dl * (k - 1) // 2]))
                if D + 2 * pad - dl * (k - 1) < 1:
                    continue
                w = torch.from_numpy(synth.normal(900 + it, "w", (cout, cin, k, 1, 1), 0.2)).to(dev)
                args = ((s, 1, 1), (pad, 0, 0), (dl, 1, 1))
                ours = lambda a, b: TF.conv3d(a, b, None, *args); ref = lambda a, b: F.conv3d(a, b, None, *args)
            elif fam == "hwT":
                cin = min(cin, 64)
                x = x[:, :cin].contiguous()
                w = torch.from_numpy(synth.normal(900 + it, "w", (cin, cout, 1, 3, 3), 0.2)).to(dev)
                args = ((1, 2, 2), (0, 1, 1), (0, 1, 1))
                ours = lambda a, b: TF.conv_transpose3d(a, b, None, *args); ref = lambda a, b: F.conv_transpose3d(a, b, None, *args)
            else:
                cin = min(cin, 64)
                x = x[:, :cin].contiguous()
                w = torch.from_numpy(synth.normal(900 + it, "w", (cin, cout, 3, 1, 1), 0.2)).to(dev)
                args = ((2, 1, 1), (1, 0, 0), (1, 0, 0))
                ours = lambda a, b: TF.conv_transpose3d(a, b, None, *args); ref = lambda a, b: F.conv_transpose3d(a, b, None, *args)
            tag = "case %d %s cin %d cout %d %s cap %d" % (it, fam, cin, cout, (B, D, H, W), cap)
            x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y1, y2 = ours(x1, w1), ref(x2, w2)
            assert y1.shape == y2.shape, tag
            assert _rel_err(y1, y2) < 3e-4, tag + " forward %.2e" % _rel_err(y1, y2)
            g = torch.from_numpy(synth.normal(900 + it, "g", tuple(y2.shape))).to(dev)
            y1.backward(g); y2.backward(g)
            assert _rel_err(x1.grad, x2.grad) < 3e-4, tag + " grad input %.2e" % _rel_err(x1.grad, x2.grad)
            assert _rel_err(w1.grad, w2.grad) < 3e-4, tag + " grad weight %.2e" % _rel_err(w1.grad, w2.grad)
    finally:
        _lib.check(_lib.lib().ts_conv_set_chunk_cap(32), "cap")


def test_element_stage_kernels_random_shapes():
    """resize+add+SiLU, 5^3 avg/max pooling and stable sort+gather (forward and backward) on random sizes, including
    non-matching up-sampling ratios and planes smaller than the pooling window (refused, like the framework's)."""
    import torch.nn.functional as F
    from temporalstereo_amd import functional as TF
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(123)
    for it in range(16):
        B, C = int(rng.choice([1, 2])), int(rng.choice([1, 5, 16]))
        D, H, W = int(rng.randint(1, 15)), int(rng.randint(1, 40)), int(rng.randint(1, 70))
        Da, Ha, Wa = max(1, D - int(rng.randint(0, 3))), int(rng.randint(1, H + 3)), int(rng.randint(1, W + 3))
        tag = "case %d %s <- %s" % (it, (B, C, D, H, W), (Da, Ha, Wa))
        a = torch.from_numpy(synth.normal(700 + it, "a", (B, C, Da, Ha, Wa))).to(dev)
        b = torch.from_numpy(synth.normal(700 + it, "b", (B, C, D, H, W))).to(dev)
        a1, b1, a2, b2 = (v.clone().requires_grad_(True) for v in (a, b, a, b))
        y1 = TF.resize_add_silu(a1, b1)
        y2 = F.silu(F.interpolate(a2, size=(D, H, W), mode="trilinear", align_corners=True) + b2)
        g = torch.from_numpy(synth.normal(700 + it, "g", (B, C, D, H, W))).to(dev)
        y1.backward(g); y2.backward(g)
        assert _rel_err(y1, y2) < 1e-4 and _rel_err(a1.grad, a2.grad) < 3e-4 and _rel_err(b1.grad, b2.grad) < 1e-4, tag + " resize"
        if min(D, H, W) < 5:       # F.avg_pool3d refuses planes smaller than its window, and so does ours
            with pytest.raises(RuntimeError, match="smaller than kernel size"):
                TF.pool5_avgmax(b)
            continue
        x1, x2 = b.clone().requires_grad_(True), b.clone().requires_grad_(True)
        av1, mx1 = TF.pool5_avgmax(x1)
        av2, mx2 = F.avg_pool3d(x2, 5, 1, 2), F.max_pool3d(x2, 5, 1, 2)
        (av1 * g + mx1 * g.flip(-1)).sum().backward(); (av2 * g + mx2 * g.flip(-1)).sum().backward()
        assert _rel_err(av1, av2) < 1e-5 and torch.equal(mx1, mx2) and _rel_err(x1.grad, x2.grad) < 3e-4, tag + " pool"
        DT = int(rng.randint(2, 15))
        vol = torch.from_numpy(synth.normal(700 + it, "v", (B, C, DT, H, W))).to(dev)
        smp = torch.from_numpy(np.round(synth.uniform(700 + it, "s", (B, DT, H, W), 0.0, 6.0))).to(dev)      # many ties
        v1, s1, v2, s2 = (t.clone().requires_grad_(True) for t in (vol, smp, vol, smp))
        ov1, os1 = TF.sort_gather(v1, s1)
        os2, order = torch.sort(s2, dim=1, stable=True)
        ov2 = torch.gather(v2, 2, order.unsqueeze(1).expand(-1, C, -1, -1, -1))
        assert torch.equal(ov1, ov2) and torch.equal(os1, os2), tag + " sort"
        gv = torch.from_numpy(synth.normal(700 + it, "gv", tuple(vol.shape))).to(dev)
        (ov1 * gv).sum().backward(); (ov2 * gv).sum().backward()
        assert torch.equal(v1.grad, v2.grad), tag + " sort backward"


def test_native_engine_random_geometries_and_temporal_states():
    """The all-HIP engine (folded BatchNorm, fused kernels, launch-plan replay, three streams) against the nn.Module
    path on random image sizes (odd quotients at 1/16, 1/8, 1/4), batch 1-3, with and without a temporal state
    carrying 0-3 local maps.  Criterion: bulk agreement (median) and rarity of flipped pixels -- see
    tests/test_fullsize_gpu.py for why the mean alone is not a robust statistic on random-weight networks."""
    import bench
    import temporalstereo_amd as ts
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(4242)
    dims = dict(coarse=dict(in_planes=32, C=8, num_sample=4), fine=dict(in_planes=16, C=8), precise=dict(in_planes=8, C=8))
    for it in range(6):
        B = int(rng.choice([1, 2, 3])); H = 16 * int(rng.randint(5, 12)); W = 16 * int(rng.randint(5, 16)); ns = int(rng.choice([3, 4, 6]))
        nl = int(rng.randint(0, 4)); temporal = bool(it % 2)
        seed = 3000 + it
        net = ts.TEMPORALSTEREO(coarse=ts.CoarseAggregation(32, 8, ns), fine=ts.FineAggregation(16, 8, 5), precise=ts.PreciseAggregation(8, 8, 5))
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_values(shapes, seed).items()}, strict=True)
        net = net.to(dev)
        lf, rf = synth.feature_pyramid(seed, B, H, W, chans=(8, 16, 32))
        il, ir = synth.images(seed, B, H, W)
        inputs = ([torch.from_numpy(x).to(dev) for x in lf], [torch.from_numpy(x).to(dev) for x in rf], torch.from_numpy(il).to(dev), torch.from_numpy(ir).to(dev))
        bench.calibrate_batchnorm(net, inputs)
        prev = {}
        if temporal:
            with torch.no_grad():
                first = net(*inputs, {})
            prev = {"cost_memory": {k: v.clone() for k, v in first[5]["cost_memory"].items()}, "use_past_cost": True}
            if nl:
                lm = torch.nn.functional.interpolate(first[0][0], size=(H // 8, W // 8), mode="bilinear", align_corners=True) / 8.0
                prev.update(local_map=torch.cat([lm + 0.6 * k for k in range(nl)], 1).contiguous(), local_map_size=nl)
        with torch.no_grad():
            ref = net(*inputs, dict(prev))
        got = InferenceEngine(net, backend="native", replay="plan")(*inputs, dict(prev))
        tag = "case %d B=%d %dx%d samples=%d temporal=%s local=%d" % (it, B, H, W, ns, temporal, nl)
        assert [tuple(c.shape) for c in got[1]] == [tuple(c.shape) for c in ref[1]], tag
        for i in range(4):
            diff = (got[0][i] - ref[0][i]).abs() * (W / ref[0][i].shape[-1])
            med, far = float(diff.median()), float((diff > 0.1).double().mean())
            assert med < 2e-3 and far < 0.02, tag + " disparity %d: median %.3g px, %.2f%% beyond 0.1 px" % (i, med, 100 * far)


def test_regression_and_splat_kernels_random_shapes():
    """top-k soft-argmax (k = 1..4), full soft-argmin / argmin over D up to 192, and the four splat modes on random
    sizes against the oracle."""
    import oracle.regress as oreg
    import oracle.splat as osp
    import temporalstereo_amd as ts
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(31337)
    for it in range(14):
        B = int(rng.choice([1, 2, 3])); D = int(rng.choice([2, 5, 7, 14, 20, 48, 192])); H = int(rng.randint(1, 40)); W = int(rng.randint(1, 70))
        cost = torch.from_numpy(synth.normal(1100 + it, "c", (B, D, H, W), 2.0))
        samp = torch.from_numpy(np.sort(synth.uniform(1100 + it, "s", (B, D, H, W), 0.0, 60.0), axis=1))
        off = torch.from_numpy(synth.uniform(1100 + it, "o", (B, D, H, W), -1.0, 1.0))
        tag = "case %d %s" % (it, (B, D, H, W))
        if D <= 20:
            k = int(rng.randint(1, min(D, 4) + 1))
            got = ts.topk_softargmax(cost.to(dev), samp.to(dev), off.to(dev), k=k)
            want = oreg.topk_softargmax(cost, samp, off, k=k)
            for g, w_, nm in zip(got, want, ("disp", "topk_disp", "topk_cost")):
                np.testing.assert_allclose(g.cpu().numpy(), w_.numpy(), rtol=1e-5, atol=2e-4, err_msg=tag + " topk k=%d %s" % (k, nm))
        for temp in (1.0, 2.5):
            got = ts.soft_argmin(cost.to(dev), samp.to(dev), temperature=temp, normalize=True)
            np.testing.assert_allclose(got.cpu().numpy(), oreg.soft_argmin(cost, samp, temp, True).numpy(), rtol=1e-5, atol=2e-4, err_msg=tag + " soft_argmin")
        np.testing.assert_allclose(ts.argmin_select(cost.to(dev), samp.to(dev)).cpu().numpy(), oreg.argmin_select(cost, samp).numpy(), rtol=0, atol=0,
                                   err_msg=tag + " argmin")
        C = int(rng.choice([1, 2, 5]))
        inp = torch.from_numpy(synth.normal(1100 + it, "i", (B, C, H, W)))
        flow = torch.from_numpy(synth.normal(1100 + it, "f", (B, 2, H, W), 3.0))
        met = torch.from_numpy(synth.normal(1100 + it, "m", (B, 1, H, W)))
        for mode in ("summation", "average", "linear", "softmax"):
            m = None if mode == "summation" else (met.abs() + 0.1 if mode == "linear" else met)
            got = ts.FunctionSoftsplat(inp.to(dev), flow.to(dev), None if m is None else m.to(dev), mode)
            np.testing.assert_allclose(got.cpu().numpy(), osp.softsplat(inp, flow, m, mode).numpy(), rtol=1e-4, atol=1e-4, err_msg=tag + " splat " + mode)


def test_three_frame_sequence_engine_vs_module_path():
    """A whole temporal sequence the way the reference's wrapper runs it (projects/TemporalStereo/TemporalStereo.py:
    282-324): frame t's aggregation writes prev_disp / cost memory, update_map moves them (and the growing local map)
    into frame t+1.  Per frame: native engine on the fused HIP update against the nn.Module path on the oracle's
    op-by-op update, both starting from the SAME previous state (the module path's) -- a random-weight network
    amplifies the few flipped pixels of one frame through the splat into the next, so two pipelines that each
    carry their own state drift apart by frame 2 (27 % of the pixels beyond 0.1 px) without either being wrong."""
    import bench
    import temporalstereo_amd as ts
    from oracle import temporal as otemporal
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    B, H, W, ns, size = 2, 128, 192, 4, 3
    seed = 5150
    net = ts.TEMPORALSTEREO(coarse=ts.CoarseAggregation(32, 8, ns), fine=ts.FineAggregation(16, 8, 5), precise=ts.PreciseAggregation(8, 8, 5))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_values(shapes, seed).items()}, strict=True)
    net = net.to(dev)
    frames = []
    for f in range(3):
        lf, rf = synth.feature_pyramid(seed + f, B, H, W, chans=(8, 16, 32))
        il, ir = synth.images(seed + f, B, H, W)
        frames.append(([torch.from_numpy(x).to(dev) for x in lf], [torch.from_numpy(x).to(dev) for x in rf],
                       torch.from_numpy(il).to(dev), torch.from_numpy(ir).to(dev)))
    bench.calibrate_batchnorm(net, frames[0])
    K = torch.from_numpy(synth.sceneflow_intrinsics(B, H, W))
    poses = [torch.from_numpy(synth.small_motion(seed + 10 * f, B)) for f in range(3)]
    eng = InferenceEngine(net, backend="native", replay="plan")

    def moved(v, where):
        if isinstance(v, dict):
            return {k: moved(x, where) for k, x in v.items()}
        return v.detach().clone().to(where) if torch.is_tensor(v) else v

    state = {}
    for f in range(3):
        info_e, info_m = {}, {}
        if f:
            inv_past = torch.inverse(poses[f - 1])
            info_e = ts.temporal.update_map(moved(state, dev), K.to(dev), poses[f].to(dev), inv_past.to(dev), 0.54, H, W,
                                            use_past_cost=True, local_map_size=size)
            info_m = moved(otemporal.update_map(moved(state, "cpu"), K, poses[f], inv_past, 0.54, H, W,
                                                use_past_cost=True, local_map_size=size), dev)
            assert info_e["local_map"].shape == info_m["local_map"].shape == (B, min(f, size), H // 8, W // 8)
            for a, b in ((info_e["local_map"], info_m["local_map"]),
                         (info_e["cost_memory"]["disp_sample"], info_m["cost_memory"]["disp_sample"]),
                         (info_e["cost_memory"]["cost_volume"], info_m["cost_memory"]["cost_volume"])):
                d = (a - b).abs()
                assert float(d.mean()) < 2e-4 and float((d > 1e-2 * (1 + b.abs())).double().mean()) < 2e-3, "frame %d state" % f
        out_e = eng(*frames[f], dict(info_e))
        with torch.no_grad():
            out_m = net(*frames[f], dict(info_m))
        tag = "frame %d" % f
        assert [tuple(c.shape) for c in out_e[1]] == [tuple(c.shape) for c in out_m[1]], tag
        for i in range(4):
            diff = (out_e[0][i] - out_m[0][i]).abs() * (W / out_m[0][i].shape[-1])
            med, far = float(diff.median()), float((diff > 0.1).double().mean())
            assert med < 2e-3 and far < 0.02, tag + " disparity %d: median %.3g px, %.2f%% beyond 0.1 px" % (i, med, 100 * far)
        for key in ("prev_disp", "cost_memory"):
            assert key in out_e[5] and key in out_m[5]
        state = moved(out_m[5], dev)
