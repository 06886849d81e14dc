"""GPU, BASELINE configurations at their STATED batches (configs[1] B=1, configs[2] B=4 T=2, configs[3] B=8 four-frame
sequence, configs[4] B=2 temporal): the all-HIP native path against the CPU oracle, three ways (tests/parity_tools.py).

 1. per-op teacher forcing  -- every HIP stage (cost volume, init3d, candidate merge, fusion, heads, top-k regression,
    upsamplers, candidate generation, memory resize) gets the ORACLE's stage input and must reproduce the oracle's
    stage output to the stated fp32 tolerance: nothing compounds, no discrete step decides on our numbers;
 2. per-level teacher forcing -- every pyramid level gets the oracle's level inputs; its low-resolution disparity may
    differ from the oracle's ONLY at pixels where the oracle's own top-k selection (or candidate order) is a near-tie
    closer than twice the measured cost error at that pixel;
 3. end to end -- whole sequences, nothing forced: |dEPE| < 1e-3 px (BASELINE.json) on the full-resolution disparity
    asserted for EVERY seed and EVERY frame, on inputs that are stereo (planted disparity, rigid scene under the poses:
    synth.stereo_sequence) with the committed trained checkpoint (tests/golden/ckpt_planted.npz), (a) against fixtures the
    imported REFERENCE produced with that checkpoint at the stated batches (tests/golden/planted_*.npz) and (b) against
    the CPU oracle on further seeds.  The random-weight / independent-noise protocol of rounds 1-2 is kept as a
    labelled stress test: an untrained pyramid is not contractive, so only its bulk can be asserted.
"""
import os

import pytest
import torch

os.environ.setdefault("MIOPEN_FIND_MODE", "2")
pytestmark = pytest.mark.gpu

import parity_tools as PT  # noqa: E402

SOFT = bool(int(os.environ.get("TS_PARITY_SOFT", "0")))          # collect the report without asserting (exploration)


def _cmp(rep, what, ours, ref, atol, rtol=0.0, **ctx):
    try:
        PT.compare(rep, what, ours, ref, atol, rtol, **ctx)
    except AssertionError:
        if not SOFT:
            raise


def _lowres(cost, samp, off, k=2):
    from temporalstereo_amd import functional as TF
    return TF.topk_softargmax(cost.contiguous(), samp.contiguous(), off.contiguous(), k)[0]


def _audit(rep, what, ours, ref, **ctx):
    """ours / ref = (cost, off, candidates) of one level.  Every pixel whose low-resolution disparity moved by more than
    1e-3 px must be a near-tie of the ORACLE: top-2 / third-best margin below twice our cost error at that pixel, or a
    candidate order decided by keys closer than 1e-4."""
    c1, o1, s1 = (x.detach().double().cpu() for x in ours)
    c0, o0, s0 = (x.detach().double().cpu() for x in ref)
    same_order = ((s1 - s0).abs() <= 1e-4).all(dim=1)                              # candidate planes line up
    key_gap = (s0[:, 1:] - s0[:, :-1]).abs()
    key_gap = torch.where(key_gap == 0, torch.full_like(key_gap, 1e9), key_gap).min(dim=1).values   # exact ties are stable: excluded
    eps = torch.where(same_order, (c1 - c0).abs().max(dim=1).values, torch.zeros_like(key_gap))
    d1 = _lowres(ours[0], ours[2], ours[1]).double().cpu()[:, 0]
    d0 = _lowres(ref[0].to(ours[0].device).float(), ref[2].to(ours[0].device).float(), ref[1].to(ours[0].device).float()).double().cpu()[:, 0]
    moved = (d1 - d0).abs() > 1e-3
    margin = PT.top_margin(c0, 2)
    near_tie = (margin <= 2 * eps + 1e-7) | (~same_order & (key_gap < 1e-4))
    unexplained = int((moved & ~near_tie).sum())
    rep.add(what=what, pixels=int(moved.numel()), moved=int(moved.sum()), unexplained=unexplained,
            reordered=int((~same_order).sum()), cost_err_max=float(eps.max()), cost_err_mean=float(eps.mean()),
            still_max=float((d1 - d0).abs()[~moved].max()), **ctx)
    if not SOFT:
        assert unexplained == 0, "%s: %d pixels moved by > 1e-3 px away from any near-tie of the oracle" % (what, unexplained)
        assert float(moved.double().mean()) < 0.01, "%s: %.2f%% of the pixels flipped" % (what, 100 * float(moved.double().mean()))


def _frame_checks(case, agg, t, trace, out_o, prev_o, rep, name):
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.aggregation import native as N
    dev, c = case.dev, case.c
    g = lambda k: trace[k].to(dev).float().contiguous()
    ctx = dict(config=name, frame=t, seed=case.seed)
    prev = PT.to_dev(PT.state_for_aggregation(prev_o), dev)
    (l4, l8, l16), (r4, r8, r16), il, ir = case.frames_gpu[t]
    co, fi, pr = agg.coarse, agg.fine, agg.precise
    B = c["B"]
    nl = prev["local_map"].shape[1] if (prev.get("local_map") is not None and prev.get("local_map_size", 0) > 0) else 0

    # ------------------------------------------------------------------------------------------ coarse level, op by op
    _cmp(rep, "coarse K1 block_cost(int)", TF.block_cost(l16, r16, c["num_sample"], 3), trace["coarse_raw"], 1e-4, 2e-5, **ctx)
    _cmp(rep, "coarse init3d", co.init3d(g("coarse_raw")), trace["coarse_init"], 1e-4, 1e-4, **ctx)
    cat4, samp = co.merge(g("coarse_init"), None, prev, True)
    _cmp(rep, "coarse merged candidates", samp, trace["coarse_ds"], 1e-4, 1e-5, **ctx)
    ok = ((samp.cpu().double() - trace["coarse_ds"].double()).abs() <= 1e-4).all(dim=1)            # our memory resize feeds the sort keys
    msk = ok[:, None, None].to(dev)
    _cmp(rep, "coarse merged volume", cat4[:, :co.C] * msk, g("coarse_merged") * msk, 1e-4, 1e-5, reordered=int((~ok).sum()), **ctx)
    cat4[:, :co.C] = g("coarse_merged")
    _cmp(rep, "coarse PyramidFusion", co.fuse(cat4), trace["coarse_fused"], 1e-4, 1e-4, **ctx)
    cost, off = co.heads(g("coarse_fused"))
    _cmp(rep, "coarse head cost", cost, trace["coarse_cost"], 1e-4, 1e-4, **ctx)
    _cmp(rep, "coarse head offset", off, trace["coarse_off"], 1e-5, 1e-4, **ctx)
    _cmp(rep, "coarse top-2 soft-argmax", _lowres(g("coarse_cost"), g("coarse_ds"), g("coarse_off")), trace["coarse_disp_lowres"], 1e-4, 1e-5, **ctx)
    up, low, high, cand = co.up.with_candidates(l16, g("coarse_disp_lowres"), None, 4, nl)
    _cmp(rep, "coarse ConvexUpsample", up, trace["coarse_up"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "fine search range low", low, trace["fine_low"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "fine candidates", cand[:, nl:], trace["fine_ds0"][:, nl:], 1e-4, 1e-5, **ctx)
    if nl:
        lm = prev["local_map"]
        N.resize_bilinear(lm, cand.shape[-2:], cand.shape[-1] / lm.shape[-1], out=cand[:, :nl])
        _cmp(rep, "fine local-map candidates", cand[:, :nl], trace["fine_ds0"][:, :nl], 1e-4, 1e-5, **ctx)

    # ------------------------------------------------------------------------------------------ fine level, op by op
    ds0 = g("fine_ds0")
    Cf = l8.shape[1]
    _cmp(rep, "fine K1 block_cost(sampled)", TF.block_cost(l8, r8, ds0, 3), trace["fine_raw"], 1e-4, 2e-5, **ctx)
    _cmp(rep, "fine K1 warped variant", TF.block_cost_warped(l8, r8, ds0, 3), trace["fine_raw"][:, Cf:], 1e-4, 2e-5, **ctx)
    _cmp(rep, "fine init3d", fi.init3d(g("fine_raw")[:, Cf:].contiguous(), fi.left_term(l8)), trace["fine_init"], 1e-4, 1e-4, **ctx)
    # the pre-contracted first layer (ts_conv3d_hw_warp_fwd): correlation blocks alone, then init3d from [corr | Q | left term]
    _cmp(rep, "fine K1 correlation blocks only", N.block_cost_corr(l8, r8, ds0, 3), trace["fine_raw"][:, 2 * Cf:], 1e-4, 2e-5, **ctx)
    _cmp(rep, "fine init3d, warped half pre-contracted", fi.init3d_from(fi.first_layer_fused(l8, r8, ds0, fi.left_term(l8), fi.right_term(r8))),
         trace["fine_init"], 1e-4, 1e-4, **ctx)
    cat4, samp = fi.merge(g("fine_init"), ds0, prev, False)
    _cmp(rep, "fine merged candidates", samp, trace["fine_ds"], 0.0, **ctx)                      # keys are the oracle's bits: exact
    _cmp(rep, "fine merged volume", cat4[:, :fi.C], trace["fine_merged"], 1e-5, 1e-5, **ctx)
    cat4[:, :fi.C] = g("fine_merged")
    _cmp(rep, "fine PyramidFusion", fi.fuse(cat4), trace["fine_fused"], 1e-4, 1e-4, **ctx)
    cost, off = fi.heads(g("fine_fused"))
    _cmp(rep, "fine head cost", cost, trace["fine_cost"], 1e-4, 1e-4, **ctx)
    _cmp(rep, "fine head offset", off, trace["fine_off"], 1e-5, 1e-4, **ctx)
    _cmp(rep, "fine top-2 soft-argmax", _lowres(g("fine_cost"), g("fine_ds"), g("fine_off")), trace["fine_disp_lowres"], 1e-4, 1e-5, **ctx)
    up, low, high, cand = fi.up.with_candidates(l8, g("fine_disp_lowres"), None, 4, 0)
    _cmp(rep, "fine ConvexUpsample", up, trace["fine_up"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "precise candidates", cand, trace["precise_ds"], 1e-4, 1e-5, **ctx)

    # ------------------------------------------------------------------------------------------ precise level, op by op
    both, (mask, lterm) = pr.unet_features(l4, r4, il, ir)
    _cmp(rep, "precise [feature | spx4] left", both[:B], trace["precise_left"], 1e-4, 1e-4, **ctx)
    _cmp(rep, "precise [feature | spx4] right", both[B:], trace["precise_right"], 1e-4, 1e-4, **ctx)
    both_o = torch.cat([g("precise_left"), g("precise_right")], 0).contiguous()
    dsp = g("precise_ds")
    Cp = both_o.shape[1]
    _cmp(rep, "precise K1 block_cost(sampled)", TF.block_cost(both_o[:B], both_o[B:], dsp, 3), trace["precise_raw"], 1e-4, 2e-5, **ctx)
    # the arbiter for the sampled cost volume is exact arithmetic: the reference's fp32 grid_sample round-trips the tap
    # position through normalised coordinates, which costs IT up to 1e-3 on these features
    # (|value| <= 180); ours must be no further from the fp64 oracle than the fp32 oracle is
    import oracle
    k64 = oracle.block_cost(trace["precise_left"].double(), trace["precise_right"].double(), trace["precise_ds"].double(), 3)
    e_ref = float((trace["precise_raw"].double() - k64).abs().max())
    e_ours = float((TF.block_cost(both_o[:B], both_o[B:], dsp, 3).cpu().double() - k64).abs().max())
    rep.add(what="precise K1 vs the fp64 oracle", ours_max_abs=e_ours, oracle_fp32_max_abs=e_ref, **ctx)
    assert SOFT or e_ours <= 1.25 * e_ref + 1e-5, "precise K1: %.3g from exact, the fp32 reference %.3g" % (e_ours, e_ref)
    _cmp(rep, "precise K1 warped variant", TF.block_cost_warped(both_o[:B], both_o[B:], dsp, 3), trace["precise_raw"][:, Cp:], 1e-4, 2e-5, **ctx)
    _cmp(rep, "precise init3d", pr.init3d(g("precise_raw")[:, Cp:].contiguous(), pr.left_term(both_o[:B])), trace["precise_init"], 1e-4, 1e-4, **ctx)
    _cmp(rep, "precise K1 correlation blocks only", N.block_cost_corr(both_o[:B], both_o[B:], dsp, 3), trace["precise_raw"][:, 2 * Cp:], 1e-4, 2e-5, **ctx)
    _cmp(rep, "precise init3d, warped half pre-contracted",
         pr.init3d_from(pr.first_layer_fused(both_o[:B], both_o[B:], dsp, pr.left_term(both_o[:B]), pr.right_term(both_o[B:]))),
         trace["precise_init"], 1e-4, 1e-4, **ctx)
    cost, off = pr.heads(g("precise_init"))
    _cmp(rep, "precise head cost", cost, trace["precise_cost"], 1e-4, 1e-4, **ctx)
    _cmp(rep, "precise head offset", off, trace["precise_off"], 1e-5, 1e-4, **ctx)
    d, ms, mc = TF.topk_softargmax(g("precise_cost"), dsp, g("precise_off"), 2)
    _cmp(rep, "precise top-2 soft-argmax", d, trace["precise_disp_lowres"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "precise memory candidates", ms, trace["precise_mem_s"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "precise memory costs", mc, trace["precise_mem_v"], 1e-4, 1e-4, **ctx)

    # ------------------------------------------------------------------------------------------ level by level
    out = ([], [], [], [], [])
    agg._coarse_level(l16, r16, dict(prev), out)
    _audit(rep, "coarse level (teacher-forced inputs)", (out[1][0], out[2][0], out[3][0]),
           (trace["coarse_cost"], trace["coarse_off"], trace["coarse_ds"]), **ctx)
    out = ([], [], [], [], [])
    agg._fine_level(l8, r8, ds0, dict(prev), out)
    _audit(rep, "fine level (teacher-forced inputs)", (out[1][0], out[2][0], out[3][0]),
           (trace["fine_cost"], trace["fine_off"], trace["fine_ds"]), **ctx)
    info = {}
    full, d, cost, off, _ = pr(both, (mask, lterm), dsp, info)
    _audit(rep, "precise level (teacher-forced candidates)", (cost, off, dsp), (trace["precise_cost"], trace["precise_off"], trace["precise_ds"]), **ctx)
    # the final x4 upsampling on the oracle's 1/4 disparity (our own decoder mask: continuous in the features)
    full_t = torch.empty_like(full)
    from temporalstereo_amd import _lib
    H4, W4 = dsp.shape[-2:]
    _lib.check(_lib.lib().ts_unet_upsample_fwd(_lib.ptr(mask), _lib.ptr(g("precise_disp_lowres")), _lib.ptr(full_t), B, H4, W4, 4 * H4, 4 * W4,
                                               N._stream()), "ts_unet_upsample_fwd")
    _cmp(rep, "precise UNet x4 upsampling", full_t, trace["precise_full"], 2e-4, 1e-5, **ctx)
    m_s, m_c = N.resize_bilinear_pair(g("precise_mem_s"), g("precise_mem_v"), (H4 // 2, W4 // 2), 0.5, 1.0)
    _cmp(rep, "next frame's memory candidates", m_s, out_o[5]["cost_memory"]["disp_sample"], 1e-4, 1e-5, **ctx)
    _cmp(rep, "next frame's memory costs", m_c, out_o[5]["cost_memory"]["cost_volume"], 1e-4, 1e-4, **ctx)


@pytest.mark.parametrize("name", list(PT.CONFIGS))
def test_every_stage_and_level_teacher_forced_at_stated_batch(name):
    """Per-op and per-level teacher forcing on every frame of the configuration's sequence (oracle state carried)."""
    import synth
    from temporalstereo_amd.aggregation.native import NativeAggregator
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    c = PT.CONFIGS[name]
    case = PT.Case(c, synth.SEED0 + 11, dev)
    agg = NativeAggregator(case.net.eval())
    agg.overlap = False
    rep = PT.Report()
    info = {}
    try:
        for t in range(c["frames"]):
            out_o, trace, prev_o = case.oracle_frame(t, info)
            _frame_checks(case, agg, t, trace, out_o, prev_o, rep, name)
            info = out_o[5]
    finally:
        rep.dump("parity_stagewise.json")


def _clone_info(info):
    return {k: (v.clone() if torch.is_tensor(v) else ({a: b.clone() for a, b in v.items()} if isinstance(v, dict) else v)) for k, v in info.items()}


def _native_sequence(case):
    """The product path end to end: launch-plan engine per frame, update_map between frames.  -> list of full-resolution maps."""
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    eng = InferenceEngine(case.net, backend="native", replay="plan")
    info, outs = {}, []
    for t in range(case.c["frames"]):
        if t > 0:
            info = case.native_update(t, info)
        on = eng(*case.frames_gpu[t], dict(info))
        info = _clone_info(on[5])
        outs.append((on[0][0].detach().clone(), on[0][1].detach().clone()))
    return outs, info


def _timed_form_sequence(case):
    """The same frames through the forms bench.py TIMES: single-frame configurations through
    InferenceEngine(pipeline=3, inputs='bind') -- four calls on the bound tensors, i.e. all three buffer sets of the pipeline and the
    first one again, every call's output checked --; temporal configurations through the two-phase begin / finish schedule of
    tools/sequence_bench.py (frame t+1's state-independent half issued before frame t's state update).
    -> list over frames of lists of (full, quarter) maps (one entry per call of that frame)."""
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    frames = case.c["frames"]
    if frames == 1:
        eng = InferenceEngine(case.net, backend="native", replay="plan", inputs="bind", pipeline=3)
        outs = []
        for _ in range(4):
            on = eng(*case.frames_gpu[0], {})
            outs.append((on[0][0].detach().clone(), on[0][1].detach().clone()))
        torch.cuda.synchronize()
        return [outs], _clone_info(on[5])
    eng = InferenceEngine(case.net, backend="native", replay="plan", inputs="bind")
    on = eng.finish(eng.begin(*case.frames_gpu[0]), {})
    info = _clone_info(on[5])
    outs = [[(on[0][0].detach().clone(), on[0][1].detach().clone())]]
    for t in range(1, frames):
        h = eng.begin(*case.frames_gpu[t])                 # before the state update of frame t-1, as the timed schedule issues it
        info = case.native_update(t, info)
        on = eng.finish(h, dict(info))
        info = _clone_info(on[5])
        outs.append([(on[0][0].detach().clone(), on[0][1].detach().clone())])
    return outs, info


_PLANTED = ("planted_c1_s0", "planted_c1_s1", "planted_c1_s2", "planted_c2_s0", "planted_c3_s0", "planted_c4_s0",
            "planted_c2_s1", "planted_c3_s1", "planted_c4_s1")


@pytest.mark.parametrize("fixture", _PLANTED)
def test_end_to_end_against_reference_fixtures_every_frame(fixture):
    """|dEPE| < 1e-3 px per frame against what the imported REFERENCE computed (tools/gen_golden.py planted_cases: its aggregator
    with the committed checkpoint, its own update_map between frames, BASELINE configurations at their stated batches).  EPE is
    over all pixels against the planted ground truth; the stored sub-sampled maps also bound the per-pixel difference."""
    import numpy as np
    import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz"))
    name = str(g["config"])
    c = PT.CONFIGS[name]
    assert (c["B"], c["H"], c["W"], c["frames"], c["n_local"]) == tuple(int(g[k]) for k in ("B", "H", "W", "frames", "n_local"))
    dev = torch.device("cuda:0")
    case = PT.PlantedCase(c, int(g["seed"]), dev)
    for t in range(c["frames"]):          # the inputs are regenerated from the seed: bit for bit, or the comparison means nothing
        assert abs(case.input_checksum[t] - float(g["input_checksum_%d" % t])) <= 1e-9 * abs(float(g["input_checksum_%d" % t])), \
            "frame %d: synthetic inputs were not regenerated identically on this host" % t
    outs, info = _native_sequence(case)
    rep = PT.Report()
    sub = int(g["sub"])
    try:
        for t, (full, quarter) in enumerate(outs):
            e = PT.epe(full, case.gt[t], case.max_disp)
            ref = torch.from_numpy(g["disp_full_sub_%d" % t]).double()
            d = (full[:, :, ::sub, ::sub].cpu().double() - ref).abs()
            dq = (quarter[:, :, ::max(sub // 2, 1), ::max(sub // 2, 1)].cpu().double() - torch.from_numpy(g["disp_precise_sub_%d" % t]).double()).abs()
            rep.add(what="end to end vs reference fixture", fixture=fixture, config=name, frame=t, epe=e, epe_reference=float(g["epe_%d" % t]),
                    delta_epe=abs(e - float(g["epe_%d" % t])), mean_abs=float(d.mean()), max_abs=float(d.max()), quarter_max_abs=float(dq.max()))
            assert abs(e - float(g["epe_%d" % t])) < 1e-3, "%s frame %d: EPE %.6f vs the reference's %.6f" % (fixture, t, e, float(g["epe_%d" % t]))
            assert float(d.mean()) < 1e-3, "%s frame %d: mean |disparity - reference| %.3g px" % (fixture, t, float(d.mean()))
            # (temporal frames: candidates from the local map / memory / search range nearly coincide, a few pixels per ten thousand
            # take the other side of such a near-tie in any two fp32 implementations; single frames have none)
            assert float((d > 0.05).double().mean()) < (1e-3 if t else 1e-5), "%s frame %d: %.4f%% of the pixels off by > 0.05 px" % (fixture, t, 100 * float((d > 0.05).double().mean()))
        dm = (info["cost_memory"]["disp_sample"].cpu().double() - torch.from_numpy(g["mem_out_disp_sample"]).double()).abs()
        rep.add(what="final cost memory vs reference fixture", fixture=fixture, mean_abs=float(dm.mean()), max_abs=float(dm.max()))
        assert float(dm.mean()) < 1e-3
    finally:
        rep.dump("parity_end_to_end_planted.json")


@pytest.mark.parametrize("fixture", _PLANTED)
def test_end_to_end_fixtures_through_the_forms_bench_times(fixture):
    """The nine reference-made fixtures once more, through what bench.py reports: the three-deep pipelined engine on bound inputs
    (its `value`) for the single-frame configurations, the two-phase begin / finish schedule (its `sequence` object) for the
    temporal ones.  Same bars as the plain engine above: |dEPE| < 1e-3 px per frame, mean |d| < 1e-3 px."""
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture + ".npz"))
    name = str(g["config"])
    c = PT.CONFIGS[name]
    dev = torch.device("cuda:0")
    case = PT.PlantedCase(c, int(g["seed"]), dev)
    for t in range(c["frames"]):
        assert abs(case.input_checksum[t] - float(g["input_checksum_%d" % t])) <= 1e-9 * abs(float(g["input_checksum_%d" % t]))
    outs, info = _timed_form_sequence(case)
    plain, _ = _native_sequence(case)
    rep = PT.Report()
    sub = int(g["sub"])
    try:
        for t, calls in enumerate(outs):
            ref = torch.from_numpy(g["disp_full_sub_%d" % t]).double()
            for k, (full, quarter) in enumerate(calls):
                e = PT.epe(full, case.gt[t], case.max_disp)
                d = (full[:, :, ::sub, ::sub].cpu().double() - ref).abs()
                same = bool(torch.equal(full, plain[t][0]))
                rep.add(what="timed form vs reference fixture", fixture=fixture, config=name, frame=t, call=k, epe=e,
                        epe_reference=float(g["epe_%d" % t]), delta_epe=abs(e - float(g["epe_%d" % t])), mean_abs=float(d.mean()),
                        max_abs=float(d.max()), bit_identical_to_plain_engine=same)
                assert abs(e - float(g["epe_%d" % t])) < 1e-3, "%s frame %d call %d: EPE %.6f vs %.6f" % (fixture, t, k, e, float(g["epe_%d" % t]))
                assert float(d.mean()) < 1e-3
                assert float((d > 0.05).double().mean()) < (1e-3 if t else 1e-5)
                if c["frames"] == 1:
                    assert same, "the pipelined engine's output differs from the plain engine's on the same inputs"
        dm = (info["cost_memory"]["disp_sample"].cpu().double() - torch.from_numpy(g["mem_out_disp_sample"]).double()).abs()
        assert float(dm.mean()) < 1e-3
    finally:
        rep.dump("parity_end_to_end_timed_forms.json")


# seeds per configuration for the oracle-side sweep (the oracle is a CPU pass per frame: configs[3] is 32 of them per seed)
_SEEDS = dict(zip(PT.CONFIGS, (6, 2, 1, 2)))


@pytest.mark.parametrize("name", list(PT.CONFIGS))
def test_end_to_end_delta_epe_every_seed_every_frame(name):
    """Whole sequences through the product path (engine + update_map) against the CPU oracle (fp32) carrying its own state,
    nothing forced, every pixel counted, planted scenes + trained checkpoint: |dEPE| < 1e-3 px for EVERY seed and EVERY frame."""
    import synth
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    c = PT.CONFIGS[name]
    rep = PT.Report()
    try:
        for k in range(_SEEDS[name]):
            seed = synth.SEED0 + 100 + 7 * k
            case = PT.PlantedCase(c, seed, dev)
            outs, _ = _native_sequence(case)
            io32 = {}
            for t in range(c["frames"]):
                o32 = case.oracle_frame(t, io32)[0]; io32 = o32[5]
                e_n, e_o = PT.epe(outs[t][0], case.gt[t], case.max_disp), PT.epe(o32[0][0], case.gt[t], case.max_disp)
                d = (outs[t][0].cpu().double() - o32[0][0].double()).abs()
                rep.add(what="end to end vs oracle, planted scene + checkpoint", config=name, seed=seed, frame=t, epe=e_n, epe_oracle=e_o,
                        delta_epe=abs(e_n - e_o), mean_abs=float(d.mean()), max_abs=float(d.max()))
                assert abs(e_n - e_o) < 1e-3, "%s seed %d frame %d: EPE %.6f vs the oracle's %.6f" % (name, seed, t, e_n, e_o)
                assert float(d.mean()) < 1e-3, "%s seed %d frame %d: mean |disparity - oracle| %.3g px" % (name, seed, t, float(d.mean()))
    finally:
        rep.dump("parity_end_to_end_planted.json")


# ---------------------------------------------------------------------------------------------------------------------------
# STRESS (not the parity bar): random weights on independent noise features, the protocol of rounds 1-2.  An untrained pyramid
# is not contractive -- the CPU oracle in fp32 vs the same oracle in fp64 moves single frames by up to 4.6e-3 px and temporal
# frames by 0.01-0.2 px -- so only the bulk is asserted here; the 1e-3 px bar is asserted above, per seed and per frame.
_STRESS_SEEDS = dict(zip(PT.CONFIGS, (3, 1, 0, 0)))


@pytest.mark.parametrize("name", [n for n in PT.CONFIGS if _STRESS_SEEDS[n]])
def test_stress_random_weights_end_to_end_bulk(name):
    import synth
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    c = PT.CONFIGS[name]
    rep = PT.Report()
    single, floor_single, temporal, floor_temporal = [], [], [], []
    dev_hip, dev_oracle = [], []          # per frame: mean |disparity - fp64 oracle| of the HIP path and of the fp32 oracle (px)
    try:
        for k in range(_STRESS_SEEDS[name]):
            seed = synth.SEED0 + 100 + 7 * k
            case = PT.Case(c, seed, dev)
            eng = InferenceEngine(case.net, backend="native", replay="plan")
            io32, io64, inat = {}, {}, {}
            for t in range(c["frames"]):
                o32 = case.oracle_frame(t, io32)[0]; io32 = o32[5]
                o64 = case.oracle_frame(t, io64, torch.float64)[0]; io64 = o64[5]
                if t > 0:
                    inat = case.native_update(t, inat)
                on = eng(*case.frames_gpu[t], dict(inat))
                inat = _clone_info(on[5])
                d, mad = PT.delta_epe(on[0][0], o32[0][0], seed + 10 * t, case.max_disp)
                f, fmad = PT.delta_epe(o32[0][0], o64[0][0], seed + 10 * t, case.max_disp)
                d64, mad64 = PT.delta_epe(on[0][0], o64[0][0], seed + 10 * t, case.max_disp)
                rep.add(what="STRESS random weights, end to end", config=name, seed=seed, frame=t, delta_epe=d, mean_abs=mad,
                        oracle_fp32_vs_fp64_delta_epe=f, oracle_fp32_vs_fp64_mean_abs=fmad, vs_fp64_delta_epe=d64, vs_fp64_mean_abs=mad64)
                (single if t == 0 else temporal).append(min(d, d64))
                (floor_single if t == 0 else floor_temporal).append(f)
                dev_hip.append((seed, t, mad64, d64)); dev_oracle.append((seed, t, fmad, f))
    finally:
        rep.dump("parity_end_to_end_stress.json")
    if SOFT:
        return
    assert max(single) < max(1e-3, 2 * max(floor_single), 5e-3), "%s: worst seed |dEPE| %.3g px (oracle fp32-vs-fp64 worst %.3g)" % (name, max(single), max(floor_single))
    if temporal:
        assert max(temporal) < max(3 * max(floor_temporal), 0.05), \
            "%s: temporal frames |dEPE| %.3g px vs the oracle's own fp32-vs-fp64 %.3g" % (name, max(temporal), max(floor_temporal))
    # "Within the reference's own fp32 noise" as a test, under SURVEY section 8(d)'s own protocol (reference
    # initialiser, random BatchNorm statistics: a network that is NOT contractive).  The fp64 oracle is the truth; the fp32 oracle --
    # the reference's arithmetic -- misses it by its own rounding noise, a different amount at every seed (5.9e-4 ... 6.6e-3 px mean
    # |difference| over three seeds).  The HIP path must not miss it by more than 1.5x that: per frame against the LARGER of that frame's
    # own oracle noise and the median of the oracle's noise over the frames of its kind (one frame's noise is one draw of a
    # heavy-tailed quantity; the HIP path's deviation is another draw, so the same-frame value alone is not a bound).
    import statistics
    for kind, sel in (("single", lambda t: t == 0), ("temporal", lambda t: t > 0)):
        noise = [m for (_, t, m, _) in dev_oracle if sel(t)]
        if not noise:
            continue
        med = statistics.median(noise)
        for (seed, t, m_hip, e_hip), (_, _, m_or, e_or) in zip(dev_hip, dev_oracle):
            if sel(t):
                assert m_hip <= 1.5 * max(m_or, med), \
                    "%s seed %d frame %d: mean |HIP - fp64 oracle| %.3g px exceeds 1.5 x the fp32 oracle's own %.3g (median over %s frames %.3g)" % (
                        name, seed, t, m_hip, m_or, kind, med)


# ---------------------------------------------------------------------------------------------------------------------------
# The temporal near-tie tail, pixel by pixel.  Temporal frames differ from the oracle by up to 0.2 px at a
# few pixels per ten thousand (profiles/r0*_parity_planted.txt); the bars above bound that tail, this test EXPLAINS it: free running
# (own state, nothing forced), every pixel of every level whose disparity moved is either
#   (a) a near-tie of the ORACLE at that level: its top-2 / third-best margin is below twice OUR measured cost error at that pixel
#       (same candidates on both sides) -- the decision the reference's own arithmetic could have taken either way, or
#   (b) downstream of such a pixel: next to a moved pixel of the level above (whose disparity seeds this level's candidates), or
#   (c) next to a pixel where the temporal state ENTERING the frame (memory candidates and costs, local map: update_map's output) differs
#       by more than 1e-4 -- what earlier frames' moved pixels and the splat's collisions leave behind, measured rather than modelled, or
#   (d) within the hourglass's reach of a near-tie that WAS taken the other way (the level's input volume changed there),
#   (e) moved by no more than the top-2 soft-argmax's own sensitivity to the cost error MEASURED at that pixel (no decision involved),
# with the region explained by (a)-(d) among the moved pixels' neighbourhoods bounded to under a tenth of the pixels (measured: 0.2-6 %),
# and every full-resolution pixel off by more than 0.05 px lies over such a 1/4-resolution pixel.
def _dilate(mask, r):
    import torch.nn.functional as F
    m = mask.float().unsqueeze(1)
    return (F.max_pool2d(m, 2 * r + 1, stride=1, padding=r) > 0).squeeze(1)


def _to_size(mask, size):
    import torch.nn.functional as F
    m = mask.float().unsqueeze(1)
    if m.shape[-2] >= size[0]:
        return (F.adaptive_max_pool2d(m, size) > 0).squeeze(1)
    return (F.interpolate(m, size=size, mode="nearest") > 0).squeeze(1)


@pytest.mark.parametrize("name", [n for n in PT.CONFIGS if PT.CONFIGS[n]["frames"] > 1])
def test_temporal_tail_is_explained_pixel_by_pixel(name):
    import synth
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    c = PT.CONFIGS[name]
    case = PT.PlantedCase(c, synth.SEED0 + 100, dev)
    eng = InferenceEngine(case.net, backend="native", replay="plan")
    info, io32 = {}, {}
    rep = PT.Report()
    try:
        for t in range(c["frames"]):
            if t > 0:
                info = case.native_update(t, info)
            entering = _clone_info(info)
            on = eng(*case.frames_gpu[t], dict(info))
            info = _clone_info(on[5])
            costs, samps, offs = [[x.detach().clone() for x in on[i]] for i in (1, 2, 3)]
            full = on[0][0].detach().cpu().double()
            o32, trace, prev_o = case.oracle_frame(t, io32)
            io32 = o32[5]
            up, state = None, None
            # (c) the temporal state as it ENTERS this frame: update_map splats the previous frame's candidates to their new positions
            # (softsplat.py:334-360: float atomics on the reference's side, collisions resolved by a softmax over near-equal metrics) --
            # where two source pixels land on one target the two sides may keep different ones, and the memory volume differs there
            memdiff = None
            if t > 0:
                for key in ("cost_volume", "disp_sample"):
                    a, b_ = entering["cost_memory"][key].detach().cpu().double(), prev_o["cost_memory"][key].double()
                    dm = (a - b_).abs()
                    while dm.dim() > 3:
                        dm = dm.max(dim=1).values
                    memdiff = (dm > 1e-4) if memdiff is None else (memdiff | _to_size(dm > 1e-4, memdiff.shape[-2:]))
                lm1, lm0 = entering.get("local_map"), prev_o.get("local_map")
                if lm1 is not None and lm0 is not None:
                    dm = (lm1.detach().cpu().double() - lm0.double()).abs().max(dim=1).values
                    memdiff = memdiff | _to_size(dm > 1e-4, memdiff.shape[-2:])
                rep.add(what="temporal tail audit, state entering the frame", config=name, frame=t, pixels=int(memdiff.numel()), differing=int(memdiff.sum()))
            for lvl, idx in (("coarse", 2), ("fine", 1), ("precise", 0)):
                c1, s1 = costs[idx].cpu().double(), samps[idx].cpu().double()
                c0, s0 = trace[lvl + "_cost"].double(), trace[lvl + "_ds"].double()
                d1 = _lowres(costs[idx], samps[idx], offs[idx]).cpu().double()[:, 0]
                d0 = trace[lvl + "_disp_lowres"].double()[:, 0]
                moved = (d1 - d0).abs() > 1e-3
                same = ((s1 - s0).abs() <= 1e-4).all(dim=1)
                eps = torch.where(same, (c1 - c0).abs().max(dim=1).values, torch.zeros_like(d0))
                tie = same & (PT.top_margin(c0, 2) <= 2 * eps + 1e-7)
                # (d) the candidate ORDER is itself a decision (fine.py:105-122: sort of [memory | range] candidates): keys of the oracle
                # closer than 1e-4 (exact ties are stable: excluded) may sort either way, and the merged volume's planes trade places
                gap = (s0[:, 1:] - s0[:, :-1]).abs()
                tie = tie | (torch.where(gap == 0, torch.full_like(gap, 1e9), gap).min(dim=1).values < 1e-4)
                # a decision taken the other way changes the level's input volume at that pixel, and the hourglass spreads that over its
                # neighbourhood, where the disparity then moves CONTINUOUSLY with the (measured, larger) cost error: the reach of an event
                tie = _dilate(tie & ((c1 - c0).abs().max(dim=1).values > 1e-3), 4) | tie
                # (e) no decision involved at all: the top-2 soft-argmax is continuous in the costs -- with weights w, 1 - w on candidates
                # s_a, s_b its derivative in either cost is w (1 - w) |s_a - s_b| <= |s_a - s_b| / 4 -- so a pixel whose costs differ by
                # eps (measured here; upstream events reach it through the hourglass) may move by eps |s_a - s_b| / 2 plus its offsets' error
                events = tie.clone()
                top2 = torch.topk(c0, 2, dim=1).indices
                spread = (torch.gather(s0, 1, top2[:, :1]) - torch.gather(s0, 1, top2[:, 1:2])).abs()[:, 0]
                o1, o0 = offs[idx].cpu().double(), trace[lvl + "_off"].double()
                off_err = (o1 - o0).abs().reshape(o1.shape[0], -1, *o1.shape[-2:]).max(dim=1).values
                tie = tie | (same & ((d1 - d0).abs() <= 0.5 * eps * spread + off_err + 1e-4))
                inherited = torch.zeros_like(moved)
                if up is not None:            # the level above seeds this level's candidates through a 3x3 convex upsampling
                    inherited |= _dilate(_to_size(up, moved.shape[-2:]), 3)
                if memdiff is not None:       # the state that entered this frame, where it differs (measured, not modelled)
                    inherited |= _dilate(_to_size(memdiff, moved.shape[-2:]), 2)
                unexplained = moved & ~(tie | inherited)
                reach = events | inherited
                # (by the fourth frame of configs[3] the entering state itself differs by > 1e-4 at 4 % of the memory's pixels: the region grows
                # with the sequence, the MOVED pixels stay at a few per thousand)
                # (the dilation radii are pixels: on KITTI's 24 x 78 coarse map ONE differing memory pixel already covers 25 of 1,872 pixels, and
                # which near-ties of frame 0's splat flip is a draw -- 28 differing memory pixels with the library SiLU in the epilogues, 51 with
                # the v_exp / v_rcp form of round 6, 8.9 % / 10.3 % of that map: maps under 4,000 pixels get 15 %)
                small_map = moved.shape[-2] * moved.shape[-1] < 4000
                assert float(reach.double().mean()) < ((0.15 if small_map else 0.10) if t <= 2 else 0.30), "%s frame %d %s level: the explained region covers %.1f%% of the pixels -- the audit says nothing" % (
                    name, t, lvl, 100 * float(reach.double().mean()))
                details = []
                for b_, y_, x_ in torch.nonzero(unexplained)[:5].tolist():
                    details.append(dict(b=b_, y=y_, x=x_, same=bool(same[b_, y_, x_]), moved_by=float((d1 - d0).abs()[b_, y_, x_]), eps=float(eps[b_, y_, x_]),
                                        spread=float(spread[b_, y_, x_]), off_err=float(off_err[b_, y_, x_]), margin=float(PT.top_margin(c0, 2)[b_, y_, x_]),
                                        cand_diff=float((s1 - s0).abs().max(dim=1).values[b_, y_, x_])))
                rep.add(what="temporal tail audit", config=name, frame=t, level=lvl, unexplained_pixels=details, pixels=int(moved.numel()), moved=int(moved.sum()), explained_region=int(reach.sum()),
                        near_ties=int((moved & tie).sum()), inherited=int((moved & ~tie & inherited).sum()), unexplained=int(unexplained.sum()),
                        cost_err_max=float(eps.max()), max_move=float((d1 - d0).abs().max()))
                assert int(unexplained.sum()) == 0, "%s frame %d %s level: %d moved pixels are neither a near-tie of the oracle nor downstream of one" % (
                    name, t, lvl, int(unexplained.sum()))
                # what the next level inherits: its candidates are this level's disparity (x2, 3x3 convex upsampling), so every pixel that
                # moved at all (beyond 2e-5, fp32 noise at these magnitudes) -- also by LESS than the 1e-3 this audit calls "moved": 9e-4 here is
                # 1.8e-3 in the next level's candidates
                up = (d1 - d0).abs() > 2e-5
                state = up | ~same
            off = (full - o32[0][0].double())[:, 0].abs() > 0.05
            cover = _dilate(_to_size(state, off.shape[-2:]), 6)       # x4 upsampling through a 3x3 mask: 4 px + the mask's reach
            rep.add(what="temporal tail audit, full resolution", config=name, frame=t, off_by_0p05=int(off.sum()), uncovered=int((off & ~cover).sum()))
            assert int((off & ~cover).sum()) == 0, "%s frame %d: %d full-resolution pixels off by > 0.05 px away from every explained 1/4-resolution pixel" % (
                name, t, int((off & ~cover).sum()))
    finally:
        rep.dump("parity_temporal_tail_audit.json")
