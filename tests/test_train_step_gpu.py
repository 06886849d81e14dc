"""The data-parallel training step (temporalstereo_amd.train.TrainStep) on one GPU: the eager step learns, and the hipGraph
replay of previous frame + update + forward + losses + backward follows the same trajectory.
Reference: projects/TemporalStereo/TemporalStereo.py:130-168 (training_step), :250-280 (multi_frame_forward)."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu

H, W, NS = 128, 256, 8


def _setup(dev, seed=5):
    """The committed trained checkpoint on a planted-disparity two-frame scene with its own poses.  (Rounds 1-2 ran this file on random
    weights and independent noise features: such a network amplifies the last-bit differences that the fp32 atomics of the splat and of a
    few backward kernels leave between ANY two runs -- first-step gradients of two runs differed by 1 % of their norm once in eight
    runs, second-step losses by 3 % -- which made a comparison of two runs a coin toss in the full suite.)"""
    import bench
    net = bench.load_trained(bench.build_model(dev, seed, NS)).train()
    sc = synth.stereo_sequence(synth.SEED0 + seed, 1, H, W, frames=2, max_disp=16 * NS)
    to = lambda a: torch.from_numpy(a).to(dev)
    frames = []
    for t, (lf, rf, il, ir) in enumerate(sc["frames"]):
        lf, rf = [to(x) for x in lf], [to(x) for x in rf]
        if t == 1:
            lf, rf = [x.requires_grad_(True) for x in lf], [x.requires_grad_(True) for x in rf]
        frames.append((lf, rf, to(il), to(ir)))
    eye = torch.eye(4, device=dev).expand(1, 4, 4).contiguous()
    return net, frames, to(sc["gt"][1]), to(sc["K"]), [(eye, eye), (to(sc["T"][1]), eye)]


def _run(graph, steps=4):
    from temporalstereo_amd.train import TrainStep
    dev = torch.device("cuda:0")
    net, frames, gt, K, poses = _setup(dev)
    step = TrainStep(net, max_disp=16 * NS, local_map_size=1, graph=graph, lr=1e-4)
    losses = []
    for _ in range(steps):
        losses.append(float(step(frames, gt, K, poses)))
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    return losses, grads, net


def test_eager_step_learns():
    losses, grads, net = _run(False)
    assert all(np.isfinite(losses)), losses
    assert min(losses[1:]) < losses[0], losses        # RMSprop on a random-weight network: not monotone, but downhill
    assert len(grads) > 200 and all(bool(torch.isfinite(g).all()) for g in grads.values())
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())


def test_graph_replay_follows_the_eager_trajectory():
    """Same initial weights, inputs and optimizer: the replayed step computes the eager step's loss and gradients and stays on its loss
    curve (the warm-up passes of the capture leave no trace in BatchNorm's running statistics).  Measured spread between any two runs on
    this checkpoint and scene (fp32 atomics in the splat and a few backward kernels): first-step gradients 2-5e-6 of their norm, losses
    1e-7 for two steps and 1.3e-5 at the third; the bounds below are 20x that."""
    e1, ge1, _ = _run(False, steps=1)
    g1, gg1, _ = _run(True, steps=1)
    np.testing.assert_allclose(g1[0], e1[0], rtol=1e-5)
    assert set(gg1) == set(ge1) and len(gg1) > 200
    num = sum(float(((gg1[n].double() - ge1[n].double()) ** 2).sum()) for n in ge1)
    den = sum(float((ge1[n].double() ** 2).sum()) for n in ge1)
    assert (num / den) ** 0.5 < 1e-4, (num, den)
    eager, ge, _ = _run(False, steps=3)
    graph, gg, net = _run(True, steps=3)
    assert all(np.isfinite(graph)), graph
    for i, rtol in enumerate((1e-5, 1e-4, 1e-3)):
        np.testing.assert_allclose(graph[i], eager[i], rtol=rtol, err_msg="step %d: %s vs %s" % (i, graph, eager))
    assert all(bool(torch.isfinite(g).all()) for g in gg.values())
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())


def test_graph_mode_refuses_an_unsafe_runtime(monkeypatch):
    from temporalstereo_amd.train import TrainStep
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        TrainStep(torch.nn.Linear(2, 2), graph=True)


def test_inference_engine_sees_the_fused_optimizer_step():
    """An engine built BEFORE training steps must not keep running on its stale folded weights: ClipRMSprop and the BatchNorm
    statistics kernels update through raw pointers, so TrainStep bumps the tensors' versions, which is what the engine stamps
    (validation during training, projects/TemporalStereo/TemporalStereo.py:170-200)."""
    from temporalstereo_amd.train import TrainStep
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    net, frames, gt, K, poses = _setup(dev)
    lf, rf, il, ir = frames[0]
    net.eval()
    early = InferenceEngine(net, backend="native", replay="plan")
    before = early(lf, rf, il, ir, {})[0][0].clone()
    step = TrainStep(net.train(), max_disp=16 * NS, local_map_size=1, lr=1e-3)
    versions = [p._version for p in step.params]
    for _ in range(2):
        step(frames, gt, K, poses)
    assert all(p._version > v for p, v in zip(step.params, versions) if p.grad is not None)
    net.eval()
    after = early(lf, rf, il, ir, {})[0][0].clone()
    fresh = InferenceEngine(net, backend="native", replay="plan")(lf, rf, il, ir, {})[0][0]
    torch.cuda.synchronize()
    assert float((after - before).abs().max()) > 1e-4, "two optimizer steps left the output unchanged: the engine did not re-fold"
    assert float((after - fresh).abs().max()) < 1e-5, "an engine built before the steps differs from one built after them"


def test_fused_optimizer_state_round_trip():
    from temporalstereo_amd.train import ClipRMSprop
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(5, 7, generator=g).to(dev)), torch.nn.Parameter(torch.randn(11, generator=g).to(dev))]
    ref = [p.detach().clone() for p in ps]
    a = ClipRMSprop(ps, lr=1e-2, max_norm=0.5)
    grads = [torch.randn(p.shape, generator=g).to(dev) for p in ps]
    for p, gr in zip(ps, grads):
        p.grad = gr.clone()
    a.step()
    state = a.state_dict()
    mid = [p.detach().clone() for p in ps]
    a.step()
    want = [p.detach().clone() for p in ps]
    # resume from the state on fresh parameters holding the mid-point values
    qs = [torch.nn.Parameter(m.clone()) for m in mid]
    b = ClipRMSprop(qs, lr=1.0, max_norm=0.5)
    b.load_state_dict(state)
    for q, gr in zip(qs, grads):
        q.grad = gr.clone()
    b.step()
    torch.cuda.synchronize()
    assert b.steps == 2 and all(float((q - w).abs().max()) == 0.0 for q, w in zip(qs, want))
    assert all(float((m - r).abs().max()) > 0 for m, r in zip(mid, ref))


@pytest.mark.parametrize("switch", ["TS_TRAIN_WGRAD_DEFER", "TS_TRAIN_NATIVE_PREV"])
def test_step_shortcuts_do_not_change_the_gradients(monkeypatch, switch):
    """Round 5's two structural shortcuts of the step against the step without them, first-step gradients of every parameter:
    TS_TRAIN_WGRAD_DEFER -- one wgrad_finish launch per step instead of one per layer (the same partial sums added in the same order:
    equal to the step's run-to-run spread); TS_TRAIN_NATIVE_PREV -- the previous (eval / no_grad) frame through the inference form
    of the network (bf16-split convolutions, contracted first layers: its state differs from the module path's by ~1e-5 px, which the
    current frame's gradients follow).  Reference: projects/TemporalStereo/TemporalStereo.py:250-280."""
    res = {}
    for v in ("1", "0"):
        monkeypatch.setenv(switch, v)
        _, g, _ = _run(False, steps=1)
        res[v] = g
    assert sorted(res["1"]) == sorted(res["0"])
    top = max(float(t.norm()) for t in res["0"].values())
    worst, where = 0.0, None
    for k, a in res["0"].items():
        # relative to the tensor's own norm, but not below 1 % of the largest gradient's: coarse.past_conv.weight (norm 8e-4 of the
        # largest) differs by 3-4e-4 of ITS norm between two runs of the very same step (tools/exp/step_spread.py: fp32 atomics of the
        # splat feed the memory it convolves)
        n = max(float(a.norm()), 1e-2 * top)
        e = float((res["1"][k] - a).norm()) / n
        if e > worst:
            worst, where = e, k
    print("%s on vs off: worst relative L2 over the parameters %.2e (%s)" % (switch, worst, where))
    # measured on / off: 0.4e-4 ... 1.1e-4 (the past_conv weights, whose gradients two runs of ONE setting move by as much)
    assert worst < (5e-4 if switch == "TS_TRAIN_WGRAD_DEFER" else 2e-3), (switch, where, worst)
