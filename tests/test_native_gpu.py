"""GPU: the all-HIP inference path (aggregation.native / InferenceEngine) against the nn.Module path
on the same weights, against the reference's golden outputs, and each K3 op against torch."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import synth
from helpers import load, t, dims_from_golden, aggregator_inputs, epe
from test_aggregator_gpu import _build, _check_against_golden

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _rand(seed, *shape, scale=1.0, dev=None):
    return t(synth.normal(seed, "x%d" % len(shape), shape, scale), dev)


# ------------------------------------------------------------------------------------------- ops
@pytest.mark.parametrize("cin,cout,stride,dil,act", [(24, 8, 1, 1, "SiLU"), (16, 32, 2, 1, "SiLU"), (8, 16, 1, 2, None),
                                                      (40, 64, 1, 1, "ReLU"), (16, 1, 1, 1, None)])
def test_conv_hw_vs_torch(cin, cout, stride, dil, act):
    from temporalstereo_amd.layers import Conv3d
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    torch.manual_seed(1)
    m = Conv3d(cin, cout, (1, 3, 3), (1, stride, stride), (0, dil, dil), (1, dil, dil), bias=True,
               norm=('BN3d', cout), activation=act).to(dev).eval()
    m.norm.running_mean.normal_(0, 0.2); m.norm.running_var.uniform_(0.5, 1.5); m.norm.weight.data.uniform_(0.5, 1.5)
    x = _rand(2, 2, cin, 5, 21, 37, dev=dev)
    with torch.no_grad():
        ref = m(x)
    got = native.conv_hw(x, native.fold_wrapper(m, "hw"), stride, dil)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("k,stride,dil,pad", [(3, 1, 1, 1), (3, 2, 1, 1), (3, 1, 2, 2), (5, 1, 1, 2), (1, 1, 1, 0)])
def test_conv_d_vs_torch(k, stride, dil, pad):
    from temporalstereo_amd.layers import Conv3d
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    torch.manual_seed(2)
    m = Conv3d(16, 16, (k, 1, 1), (stride, 1, 1), (pad, 0, 0), (dil, 1, 1), bias=False, norm=('BN3d', 16),
               activation='SiLU').to(dev).eval()
    m.norm.running_mean.normal_(0, 0.2); m.norm.running_var.uniform_(0.5, 1.5)
    x = _rand(3, 2, 16, 7, 9, 13, dev=dev)
    with torch.no_grad():
        ref = m(x)
    got = native.conv_d(x, native.fold_wrapper(m, "d"), k, stride, dil, pad)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_transposed_separable_vs_torch():
    from temporalstereo_amd.aggregation import native
    from temporalstereo_amd.aggregation.blocks import DepthwiseConvTranspose3D, DepthwiseConv3D
    dev = _dev()
    torch.manual_seed(3)
    for mod, transposed in ((DepthwiseConvTranspose3D(16, 8, 3, 2, 1, 1, activation=None), True),
                            (DepthwiseConv3D(16, 32, 3, 2, 1), False), (DepthwiseConv3D(8, 8, 3, 1, 2, dilation=2), False)):
        mod = mod.to(dev).eval()
        for bn in [m for m in mod.modules() if isinstance(m, nn.BatchNorm3d)]:
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
        x = _rand(4, 2, mod.conv[0].in_channels, 3, 9, 15, dev=dev)
        with torch.no_grad():
            ref = mod(x)
        got = native.SepConv(mod, transposed)(x)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_resize_add_act_pool_and_bilinear_vs_torch():
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    a = _rand(5, 2, 8, 4, 18, 30, dev=dev); b = _rand(6, 2, 8, 3, 17, 30, dev=dev)
    ref = F.silu(F.interpolate(a, size=(3, 17, 30), mode='trilinear', align_corners=True) + b)
    np.testing.assert_allclose(native.resize_add_act(a, b, (3, 17, 30)).cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    x = _rand(7, 2, 6, 7, 20, 45, dev=dev)
    avg = torch.empty_like(x); mx = torch.empty_like(x)
    native.pool5(x, avg, mx)
    np.testing.assert_allclose(avg.cpu().numpy(), F.avg_pool3d(x, 5, 1, 2).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(mx.cpu().numpy(), F.max_pool3d(x, 5, 1, 2).cpu().numpy())
    y = _rand(8, 2, 3, 17, 30, dev=dev)
    ref = F.interpolate(y * 2.5, size=(8, 15), mode='bilinear', align_corners=True)
    np.testing.assert_allclose(native.resize_bilinear(y, (8, 15), 2.5).cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_conv_hw_addend_is_the_split_of_the_input_channels():
    """layer(x) == rest(x[:, n:], addend = head(x[:, :n]) repeated over D) when x[:, :n] is constant over D
    (the reference half of the sampled cost volume, block_cost.py:51)."""
    from temporalstereo_amd.layers import Conv3d
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    torch.manual_seed(5)
    n, cin, cout, D = 16, 40, 8, 5
    m = Conv3d(cin, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 1, 1), bias=False, norm=('BN3d', cout),
               activation='SiLU').to(dev).eval()
    m.norm.running_mean.normal_(0, 0.2); m.norm.running_var.uniform_(0.5, 1.5)
    left = _rand(7, 2, n, 19, 45, dev=dev)
    rest_in = _rand(8, 2, cin - n, D, 19, 45, dev=dev)
    x = torch.cat([left.unsqueeze(2).expand(-1, -1, D, -1, -1), rest_in], 1)
    with torch.no_grad():
        ref = m(x)
    head, rest = native.fold_split_input(native.fold_wrapper(m, "hw"), n)
    got = native.conv_hw(rest_in, rest, addend=native.conv_hw(left.unsqueeze(2), head))
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_block_cost_warped_is_the_volume_without_its_reference_half():
    import temporalstereo_amd.functional as TF
    dev = _dev()
    l, r = _rand(11, 2, 32, 24, 40, dev=dev), _rand(12, 2, 32, 24, 40, dev=dev)
    d = t(synth.uniform(13, "d", (2, 5, 24, 40), -2.0, 30.0), dev)
    full = TF.block_cost(l, r, d, 3)
    part = TF.block_cost_warped(l, r, d, 3)
    assert part.shape[1] == full.shape[1] - 32
    assert torch.equal(part, full[:, 32:])


def test_copy_rows_and_stream_fork():
    from temporalstereo_amd import _lib
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    big = torch.zeros(3, 10, 4, 5, device=dev)
    src = _rand(21, 3, 4, 4, 5, dev=dev)
    native.copy_rows(src, big[:, 6:])
    assert torch.equal(big[:, 6:], src) and float(big[:, :6].abs().sum()) == 0.0
    # both forms of the kernel: rows of whole aligned quads (16 bytes per lane) and everything else (ragged rows, pitches that are not
    # multiples of 4, a destination 4 bytes off a 16-byte boundary), each into a frame that must stay untouched
    for rows, elems, spitch, dpitch, doff in ((1, 1 << 21, 1 << 21, 1 << 21, 0), (3, 4096, 4096, 8192, 4), (5, 1000, 1004, 1012, 8),
                                              (2, 999, 1000, 1001, 0), (4, 64, 64, 68, 1), (7, 8, 8, 12, 3), (1, 3, 3, 3, 0)):
        srcb = _rand(100 + rows, rows * spitch, dev=dev)
        dstb = torch.full((doff + rows * dpitch + 5,), -7.0, device=dev)
        L = _lib.lib()
        _lib.check(L.ts_copy_rows_fwd(_lib.ptr(srcb), _lib.ptr(dstb[doff:]), rows, elems, spitch, dpitch, native._stream()), "ts_copy_rows_fwd")
        torch.cuda.synchronize()
        want = torch.full_like(dstb, -7.0)
        for r in range(rows):
            want[doff + r * dpitch: doff + r * dpitch + elems] = srcb[r * spitch: r * spitch + elems]
        assert torch.equal(dstb, want), (rows, elems, spitch, dpitch, doff)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    out = torch.empty(1 << 22, device=dev)
    with torch.cuda.stream(a):
        out.fill_(3.0)
    native._edge(a, b)                                   # b waits for a
    with torch.cuda.stream(b):
        out.mul_(2.0)
    torch.cuda.synchronize()
    assert float(out.min()) == 6.0 and float(out.max()) == 6.0


def test_plan_replay_tracks_new_inputs_and_refuses_unknown_calls():
    """The recorded plan re-issues the pass on whatever the static input buffers hold."""
    from temporalstereo_amd import _lib
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    g = load("agg_tiny_single"); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    plan = InferenceEngine(net, backend="native", replay="plan")
    eager = InferenceEngine(net, backend="native", replay="eager")
    plan(lf, rf, il, ir, dict(prev))                                   # records
    lf2 = [x.flip(-1).contiguous() for x in lf]; rf2 = [x.flip(-1).contiguous() for x in rf]
    want = eager(lf2, rf2, il.flip(-1).contiguous(), ir.flip(-1).contiguous(), dict(prev))[0][0].clone()
    got = plan(lf2, rf2, il.flip(-1).contiguous(), ir.flip(-1).contiguous(), dict(prev))[0][0]
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-4)
    cap = next(iter(plan._graphs.values()))
    assert len(cap.recorder) > 50
    # inputs='bind': no copies -- the plan reads the caller's tensors in place, new storage = new plan
    bound = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
    first = bound(lf, rf, il, ir, dict(prev))[0][0].clone()
    for dst, src in zip(lf + rf, lf2 + rf2):
        dst.copy_(src)                                                 # "the backbone" writes the next frame in place
    il.copy_(il.flip(-1).contiguous()); ir.copy_(ir.flip(-1).contiguous())
    got2 = bound(lf, rf, il, ir, dict(prev))[0][0]
    assert len(bound._graphs) == 1
    np.testing.assert_allclose(got2.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-4)
    assert float((first - got2).abs().max()) > 1e-3
    L = _lib.lib()
    p = L.ts_plan_create()
    words = (_lib.ctypes.c_ulonglong * 1)(0)
    assert L.ts_plan_add_call(p, b"ts_version", words, 0) == -3          # not a launching entry point
    assert L.ts_plan_add_call(p, b"ts_stream_fork", words, 1) == -2      # wrong arity
    assert L.ts_plan_length(p) == 0 and L.ts_plan_run(p) == 0
    L.ts_plan_destroy(p)


def test_convex_upsample_with_candidates_equals_the_two_separate_kernels():
    from temporalstereo_amd.aggregation import blocks, native
    dev = _dev()
    torch.manual_seed(3)
    m = blocks.ConvexUpsample(16, upscale_factor=2, window_size=3).to(dev).eval()
    up = native.ConvexUp(m)
    feat = _rand(41, 2, 16, 17, 30, dev=dev)
    disp = t(synth.uniform(42, "d", (2, 1, 17, 30), 0.0, 40.0), dev)
    with torch.no_grad():
        ref = m(feat, disp)
    out = up(feat, disp)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    low, high, cand = native.range_candidates(out, 4.0, extra_front=2)
    out2, low2, high2, cand2 = up.with_candidates(feat, disp, None, 4.0, extra_front=2)
    assert torch.equal(out2, out) and torch.equal(low2, low) and torch.equal(high2, high)
    assert torch.equal(cand2[:, 2:], cand[:, 2:])
    want = torch.cat([(high - low).abs() * s + torch.min(low, high) for s in (0.0, 0.375, 0.5, 0.625, 1.0)], 1)   # fine.py:82-87
    np.testing.assert_allclose(cand[:, 2:].cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("name", ["agg_tiny_single", "agg_tiny_temporal"])
@pytest.mark.parametrize("replay", ["eager", "graph", "plan"])
def test_native_engine_matches_reference(name, replay):
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    g = load(name); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    eng = InferenceEngine(net, backend="native", replay=replay)
    outs = eng(lf, rf, il, ir, prev)
    _check_against_golden(g, outs)
    outs2 = eng(lf, rf, il, ir, prev)            # replay / second call gives the same answer
    assert epe(outs2[0][0].cpu(), t(g["disp_full"])) < 1e-3


def test_native_vs_module_config1():
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    g = load("agg_config1_256x512"); dev = _dev()
    dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    with torch.no_grad():
        ref = net(lf, rf, il, ir, dict(prev))
    got = InferenceEngine(net, backend="native", graph=False)(lf, rf, il, ir, dict(prev))
    for a, b in zip(got[0], ref[0]):
        assert epe(a.cpu(), b.cpu()) * (512 / a.shape[-1]) < 1e-3
    assert epe(got[0][0][:, :, ::4, ::4].cpu(), t(g["disp_full_sub4"])) < 1e-3


def test_helper_streams_are_qualified_and_shared():
    """The overlap's helper streams are picked by their cross-stream edge cost (some high-priority streams of
    this runtime signal 4-5x slower than others, which triples the pass time) and shared per device."""
    import bench
    from temporalstereo_amd.aggregation import native
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    net = bench.build_model(dev, synth.SEED0, 4).eval()
    e1, e2 = InferenceEngine(net), InferenceEngine(net)
    assert e1.net.fast is e2.net.fast and e1.net.aux is e2.net.aux and e1.net.fast is not e1.net.aux
    e3 = InferenceEngine(net, private_streams=True)
    assert e3.net.fast is not e1.net.fast and e3.net.aux is not e1.net.aux
    main = torch.cuda.current_stream()
    floor = native._round_trip_us(main, torch.cuda.Stream(device=dev))
    for s in (e1.net.fast, e1.net.aux, e3.net.fast, e3.net.aux):
        assert s.priority == -1
        assert native._round_trip_us(main, s) < 2.5 * floor


@pytest.mark.parametrize("aligned", [True, False])
def test_unet_upsample_kernel_vs_module_formula(aligned):
    """ts_unet_upsample_fwd (vector path: 4 pixels per thread; scalar path when the output is not 16-byte aligned)
    against the torch formulation of UNet.upsample (module.py:468-482)."""
    from temporalstereo_amd import _lib
    dev = _dev()
    B, h, w = 2, 9, 13
    Ho, Wo = 4 * h, 4 * w
    mask = _rand(61, B, 9, Ho, Wo, scale=2.0, dev=dev)
    disp = t(synth.uniform(62, "d", (B, 1, h, w), 0.0, 40.0), dev)
    nb = F.unfold(disp, kernel_size=(3, 3), padding=(1, 1)).reshape(B, 9, h, w)
    ref = torch.sum(F.interpolate(nb * Wo / w, size=(Ho, Wo), mode='bilinear', align_corners=True) * F.softmax(mask, dim=1),
                    dim=1, keepdim=True)
    buf = torch.zeros(B * Ho * Wo + 4, device=dev)
    out = buf[0 if aligned else 1:][:B * Ho * Wo].view(B, 1, Ho, Wo)
    rc = _lib.lib().ts_unet_upsample_fwd(_lib.ptr(mask), _lib.ptr(disp), _lib.ptr(out), B, h, w, Ho, Wo,
                                         _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _lib.check(rc, "ts_unet_upsample_fwd")
    assert float((out - ref).abs().max()) < 2e-4


def test_resize_bilinear_pair_matches_two_single_calls():
    from temporalstereo_amd.aggregation import native
    dev = _dev()
    a, b = _rand(63, 2, 2, 17, 30, dev=dev), _rand(64, 2, 2, 17, 30, dev=dev)
    pa, pb = native.resize_bilinear_pair(a, b, (8, 15), 0.5, 1.0)
    assert torch.equal(pa, native.resize_bilinear(a, (8, 15), 0.5)) and torch.equal(pb, native.resize_bilinear(b, (8, 15), 1.0))
    ref = F.interpolate(a * 0.5, size=(8, 15), mode='bilinear', align_corners=True)
    assert float((pa - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("depth", [2, 3])
def test_engine_passes_in_flight_match_plain_engine(depth):
    """pipeline=N: the pass is recorded on N sets of buffers used in turn, as a three-stage pipeline over the engine's
    streams that does not wait for the previous call's tail.  Outputs are bit-identical to the plain engine, follow the
    CONTENT of the bound inputs, and a call's outputs survive exactly N-1 further calls."""
    import bench
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = _dev()
    seed = synth.SEED0 + 7
    dims = dict(coarse=dict(in_planes=32, C=8, num_sample=4), fine=dict(in_planes=16, C=8), precise=dict(in_planes=8, C=8))
    import temporalstereo_amd as ts
    net = ts.TEMPORALSTEREO(coarse=ts.CoarseAggregation(32, 8, 4), fine=ts.FineAggregation(16, 8, 5), precise=ts.PreciseAggregation(8, 8, 5))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_values(shapes, seed).items()}, strict=True)
    net = net.to(dev)
    B, H, W = 2, 96, 160

    def frame(s):
        lf, rf = synth.feature_pyramid(s, B, H, W, chans=(8, 16, 32))
        il, ir = synth.images(s, B, H, W)
        return ([torch.from_numpy(x).to(dev) for x in lf], [torch.from_numpy(x).to(dev) for x in rf],
                torch.from_numpy(il).to(dev), torch.from_numpy(ir).to(dev))
    bound, other = frame(seed), frame(seed + 1)
    bench.calibrate_batchnorm(net, bound)
    plain = InferenceEngine(net, backend="native", replay="plan")
    want_a = [t.clone() for t in plain(*bound, {})[0]]
    want_b = [t.clone() for t in plain(*other, {})[0]]
    with pytest.raises(ValueError):
        InferenceEngine(net, backend="native", replay="plan", inputs="copy", pipeline=2)
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth)
    first = [eng(*bound, {}) for _ in range(depth)]
    o1, o2 = first[0], first[-1]
    assert len({o[0][0].data_ptr() for o in first}) == depth                 # N sets of buffers
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for o in first for a, b in zip(o[0], want_a))
    # new content in the bound tensors (the producer's job; complete on the device before the call)
    for dst, src in zip(bound[0] + bound[1] + [bound[2], bound[3]], other[0] + other[1] + [other[2], other[3]]):
        dst.copy_(src)
    torch.cuda.synchronize()
    o3 = eng(*bound, {})                                                     # re-uses the buffers of o1
    torch.cuda.synchronize()
    assert o3[0][0].data_ptr() == o1[0][0].data_ptr()
    assert all(torch.equal(a, b) for a, b in zip(o3[0], want_b))
    assert all(torch.equal(a, b) for a, b in zip(o2[0], want_a))            # the call before is still intact
    for _ in range(20):                                                      # steady state, no synchronisation in between
        last = eng(*bound, {})
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(last[0], want_b))


def test_engine_refolds_after_load_state_dict_and_inplace_updates():
    """The engine works on folded copies of the weights (and plans bake pointers to them): loading a checkpoint or
    stepping an optimizer AFTER construction (demo.py:250 of the reference loads after building) must be seen."""
    import synth
    from helpers import load, dims_from_golden, synth_state, aggregator_inputs
    import temporalstereo_amd as ts
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    g = load("agg_tiny_single")
    dims = dims_from_golden(g)
    mk = lambda: ts.TEMPORALSTEREO(
        coarse=ts.CoarseAggregation(dims['coarse']['in_planes'], dims['coarse']['C'], dims['coarse']['num_sample']),
        fine=ts.FineAggregation(dims['fine']['in_planes'], dims['fine']['C'], 5),
        precise=ts.PreciseAggregation(dims['precise']['in_planes'], dims['precise']['C'], 5)).to(dev).eval()
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    net = mk()                                                  # random initial weights
    eng = InferenceEngine(net, backend="native", replay="plan")
    before = eng(lf, rf, il, ir, dict(prev))[0][0].clone()
    net.load_state_dict(synth_state(dims, int(g["seed"]), golden=g), strict=True)     # the checkpoint arrives afterwards
    after = eng(lf, rf, il, ir, dict(prev))[0][0].clone()
    ref = torch.from_numpy(g["disp_full"]).to(dev)
    assert float((after - ref).abs().mean()) < 1e-3, "engine kept the weights it was built with"
    assert float((before - ref).abs().mean()) > 1e-2
    with torch.no_grad():                                       # an optimizer-style in-place update of every tensor
        for p in net.parameters():
            p.mul_(1.01)
    moved = eng(lf, rf, il, ir, dict(prev))[0][0].clone()
    fresh = InferenceEngine(net, backend="native", replay="plan")(lf, rf, il, ir, dict(prev))[0][0]
    assert float((moved - fresh).abs().max()) == 0.0
    assert float((moved - after).abs().mean()) > 0.0


def test_ops_refuse_cpu_and_mixed_inputs():
    import temporalstereo_amd as ts
    x = torch.zeros(1, 8, 4, 8)
    with pytest.raises(RuntimeError):
        ts.block_cost(x, x, 3, 3)


def test_two_phase_sequence_is_bit_identical_to_single_calls():
    """engine.begin / finish (frame t+1's state-independent half issued before frame t's state update; two alternating buffer
    sets) against plain calls of the same engine kind, over a four-frame temporal sequence with growing local maps."""
    import synth
    import bench
    from temporalstereo_amd import temporal
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 77
    B, H, W = 2, 192, 320
    net = bench.build_model(dev, seed, 6)
    frames = [bench.make_inputs(dev, seed + 10 * t, B, (H, W)) for t in range(4)]
    bench.calibrate_batchnorm(net, frames[0])
    K = torch.from_numpy(synth.sceneflow_intrinsics(B, H, W)).to(dev)
    T = torch.from_numpy(synth.small_motion(seed, B)).to(dev)
    eye = torch.eye(4, device=dev).expand(B, 4, 4).contiguous()

    def update(info):
        return temporal.update_map(dict(info), K, T, eye, 0.54, H, W, use_past_cost=True, local_map_size=3)

    def clone(out):
        return [d.clone() for d in out[0]], {k: (v.clone() if torch.is_tensor(v) else ({a: b.clone() for a, b in v.items()} if isinstance(v, dict) else v))
                                            for k, v in out[5].items()}
    plain = InferenceEngine(net, backend="native", replay="plan")
    want, states, info = [], [], {}
    for t in range(4):
        # the splat inside update_map accumulates with fp32 atomics (run-to-run rounding differs, as in the reference): both
        # engines are fed the SAME updated states
        state = {} if t == 0 else update(info)
        states.append({k: (v.clone() if torch.is_tensor(v) else ({a: b.clone() for a, b in v.items()} if isinstance(v, dict) else v)) for k, v in state.items()})
        d, info = clone(plain(*frames[t], state))
        want.append(d)
    # the frames live in ONE set of bound tensors that the producer refills (begin(t+1) is issued after finish(t) was issued;
    # the copy into the bound tensors is ordered on the caller's stream and begin's streams do not wait for it -> synchronise)
    bound = [[x.clone() for x in frames[0][0]], [x.clone() for x in frames[0][1]], frames[0][2].clone(), frames[0][3].clone()]

    def load(t):
        torch.cuda.synchronize()
        for dst, src in zip(bound[0] + bound[1] + bound[2:], frames[t][0] + frames[t][1] + list(frames[t][2:])):
            dst.copy_(src)
        torch.cuda.synchronize()
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
    for rep in range(2):                                   # second repetition: every plan is a replay
        load(0)
        out = eng.finish(eng.begin(*bound), {})
        d, info = clone(out)
        assert all(float((a - b).abs().max()) == 0.0 for a, b in zip(d, want[0])), "frame 0"
        for t in range(1, 4):
            load(t)
            h = eng.begin(*bound)
            update(info)                                     # the state update the early half overlaps (its result is replaced by
            out = eng.finish(h, dict(states[t]))             # the recorded one, see above)
            d, info = clone(out)
            assert all(float((a - b).abs().max()) == 0.0 for a, b in zip(d, want[t])), "frame %d (repetition %d)" % (t, rep)


def _folded_arrays(obj, prefix="", seen=None, out=None):
    """Every kernel-layout array reachable from a NativeAggregator: {path: tensor}."""
    from temporalstereo_amd.aggregation import native as N
    seen = set() if seen is None else seen
    out = {} if out is None else out
    if id(obj) in seen:
        return out
    seen.add(id(obj))
    if isinstance(obj, N.Folded):
        for k in ("w", "scale", "shift", "w6", "w6s"):
            t = getattr(obj, k, None)
            if torch.is_tensor(t):
                out[prefix + "." + k] = t
        return out
    if isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            _folded_arrays(v, "%s[%d]" % (prefix, i), seen, out)
        return out
    if hasattr(obj, "__dict__") and type(obj).__module__.startswith("temporalstereo_amd.aggregation.native"):
        for k, v in vars(obj).items():
            if k in ("mod", "tape", "fast", "aux") or k.startswith("_tape"):
                continue
            if torch.is_tensor(v):
                out[prefix + "." + k] = v
            else:
                _folded_arrays(v, prefix + "." + k, seen, out)
    return out


def test_in_place_refold_equals_a_rebuild():
    """native.Tape: after EVERY parameter and running statistic changed, refresh_weights() -- one ts_conv_weight_layout_many2 launch, one
    ts_bn_fold_many launch, one split launch per bf16-split copy -- leaves every folded array (weights in kernel layout incl. the merged
    heads, the block-diagonal pair, the Q weights of the warp-commuted first layers; scale / shift; x6 / x6s splits) equal to those of
    an aggregator BUILT from the new values, in the same storage; and an engine keeps its recorded plan across refresh().
    Reference behaviour this serves: weights that change under a live model -- demo.py:250 loads the checkpoint after building it, and
    a training step's previous frames (projects/TemporalStereo/TemporalStereo.py:268-274) see the optimizer's latest update."""
    import bench
    from temporalstereo_amd.aggregation import native as N
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 2
    net = bench.build_model(dev, seed).eval()
    inputs = bench.make_inputs(dev, seed, 1, (256, 512))
    bench.calibrate_batchnorm(net, inputs)
    eng = InferenceEngine(net, backend="native", replay="plan")
    with torch.no_grad():
        eng(*inputs, {})                                         # records the plan; creates the lazily split copies
    agg = eng.net
    assert not agg.tape.unsupported
    before = {k: (v.data_ptr(), v.clone()) for k, v in _folded_arrays(agg).items()}
    plans = dict(eng._graphs)
    g = torch.Generator(device="cpu").manual_seed(7)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.0 + 0.2 * torch.randn(p.shape, generator=g).to(dev)).add_(0.01 * torch.randn(p.shape, generator=g).to(dev))
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.add_(0.1 * torch.randn(b.shape, generator=g).to(dev))
            elif name.endswith("running_var"):
                b.mul_(1.0 + 0.3 * torch.rand(b.shape, generator=g).to(dev))
    with torch.no_grad():
        out_a = eng(*inputs, {})                                 # the version stamp moved: the engine re-folds in place, same plan
    assert eng._graphs == plans and all(eng._graphs[k] is plans[k] for k in plans), "the recorded plan was dropped"
    after = _folded_arrays(agg)
    fresh_eng = InferenceEngine(net, backend="native", replay="plan")
    with torch.no_grad():
        out_b = fresh_eng(*inputs, {})
    fresh = _folded_arrays(fresh_eng.net)
    assert sorted(after) == sorted(fresh) == sorted(before)
    changed = 0
    for k, t in after.items():
        assert t.data_ptr() == before[k][0], k                  # in place
        f = fresh[k]
        if t.dtype == torch.uint8:                               # bf16-split copies: made by the same kernel from the re-laid weights
            assert torch.equal(t, f), k
        else:
            assert torch.allclose(t, f, rtol=2e-6, atol=1e-7), (k, float((t - f).abs().max()))
        changed += int(not torch.equal(t, before[k][1]))
    assert changed > 0.8 * len(after)                            # (the perturbation really reached the arrays)
    d = float((out_a[0][0] - out_b[0][0]).abs().mean())           # (scale / shift may differ in the last bit: a contraction in the fold kernel)
    assert d < 1e-3, d
