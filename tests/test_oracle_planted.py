"""CPU: the oracle at FULL size against the imported reference, with the trained checkpoint on planted-disparity scenes
(tests/golden/planted_*.npz from tools/gen_golden.py planted_cases; checkpoint tests/golden/ckpt_planted.npz from
tools/train_checkpoint.py).  Pins (i) the scene generator's bytes on this host, (ii) the oracle aggregation + the oracle's
temporal update at BASELINE sizes, carrying state over a T=2 sequence."""
import os

import numpy as np
import pytest
import torch

import parity_tools as PT
import synth
from helpers import load


def test_planted_scene_is_stereo_and_rigid():
    """left(x) = right(x - d(x)) at every level, and frame t's disparity is frame t-1's moved by the pose (update_map)."""
    from oracle import temporal as otemp
    H, W = 128, 256
    s = synth.stereo_sequence(5, 2, H, W, frames=2, max_disp=64, fx=300.0, noise=0.0, bumps=False)
    lf, rf, il, ir = s["frames"][0]
    d = s["gt"][0]
    x = 200
    for lvl, sc in enumerate((4, 8, 16)):
        L, R = lf[lvl], rf[lvl]
        xs = x // sc
        dd = d[0, 0, :: sc, :: sc][: L.shape[2], xs] / sc           # approximate (nearest full-resolution sample)
        src = xs - dd
        x0 = np.floor(src).astype(int)
        fr = src - x0
        rows = np.arange(L.shape[2])
        want = (1 - fr) * R[0, 0, rows, x0] + fr * R[0, 0, rows, x0 + 1]
        assert np.abs(L[0, 0, :, xs] - want).max() < 0.15           # smooth unit-variance maps: a few % of a pixel of slack
    info = {"prev_disp": torch.from_numpy(s["gt"][0])}
    eye = torch.eye(4).expand(2, 4, 4).contiguous()
    out = otemp.update_map(info, torch.from_numpy(s["K"]), torch.from_numpy(s["T"][1]), eye, 1.0, H, W, use_past_cost=False, local_map_size=1)
    lm = out["local_map"]
    up = torch.nn.functional.interpolate(lm * W / lm.shape[-1], size=(H, W), mode="bilinear", align_corners=True)
    err = (up - torch.from_numpy(s["gt"][1])).abs()[:, :, 16:-16, 40:-40]
    moved = (torch.from_numpy(s["gt"][1]) - torch.from_numpy(s["gt"][0])).abs().mean()
    assert float(err.mean()) < 0.1 and float(err.mean()) < 0.25 * float(moved)


@pytest.mark.parametrize("fixture", ["planted_c1_s1", "planted_c4_s0"])
def test_oracle_matches_reference_at_full_size(fixture):
    from oracle import aggregation as oagg
    from oracle import temporal as otemp
    g = load(fixture)
    c = PT.CONFIGS[str(g["config"])]
    B, H, W, frames = (int(g[k]) for k in ("B", "H", "W", "frames"))
    max_disp, sub = int(g["max_disp"]), int(g["sub"])
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sc = synth.stereo_sequence(int(g["seed"]), B, H, W, frames=frames, max_disp=max_disp, fx=c["fx"], baseline=c["baseline"])
    sd = PT.load_checkpoint()
    T = torch.from_numpy
    eye = torch.eye(4).expand(B, 4, 4).contiguous()
    info = {}
    for t in range(frames):
        lf, rf, il, ir = sc["frames"][t]
        cs = synth.checksum([lf, rf, il, ir])
        assert abs(cs - float(g["input_checksum_%d" % t])) <= 1e-9 * abs(cs), "scene generator does not reproduce the fixture's inputs"
        if t > 0:
            info = otemp.update_map(dict(info), T(sc["K"]), T(sc["T"][t]), eye, c["baseline"], H, W, use_past_cost=True, local_map_size=c["n_local"])
        with torch.no_grad():
            out = oagg.aggregate(sd, [T(x) for x in lf], [T(x) for x in rf], T(il), T(ir), dict(info), cfg=dict(coarse=dict(num_sample=c["num_sample"])))
        info = out[5]
        e = PT.epe(out[0][0], T(sc["gt"][t]), max_disp)
        d = (out[0][0][:, :, ::sub, ::sub].double() - T(g["disp_full_sub_%d" % t]).double()).abs()
        assert abs(e - float(g["epe_%d" % t])) < 1e-5, (t, e, float(g["epe_%d" % t]))
        assert float(d.mean()) < 1e-4 and float(d.max()) < 2e-2, (t, float(d.mean()), float(d.max()))
    assert 0.02 < float(g["epe_0"]) < 0.5           # the checkpoint is a trained one: sub-pixel on its own training distribution


def test_oracle_backward_matches_reference_autograd_in_float64():
    """The oracle in train mode, differentiated by the framework in float64, against the reference's own float64 backward
    (tests/golden/planted_train_grads.npz, keys 'f64::*'): loss terms, feature gradients, every parameter's gradient norm and
    seeded projection.  This is what makes the oracle an arbiter for the product's backward (tests/test_backward_stagewise_gpu.py)."""
    from oracle import aggregation as oagg
    from oracle import losses as olo
    g = load("planted_train_grads")
    B, H, W, ns, max_disp, seed = (int(g[k]) for k in ("B", "H", "W", "num_sample", "max_disp", "seed"))
    sc = synth.stereo_sequence(seed, B, H, W, frames=1, max_disp=max_disp, fx=float(g["fx"]))
    assert abs(synth.checksum(list(sc["frames"][0])) - float(g["input_checksum"])) <= 1e-9 * abs(float(g["input_checksum"]))
    T64 = lambda a: torch.from_numpy(a).double()
    lf, rf, il, ir = sc["frames"][0]
    sd = {k: (v.double().requires_grad_(not k.endswith(("running_mean", "running_var"))) if v.is_floating_point() else v)
          for k, v in PT.load_checkpoint().items()}
    lf64, rf64 = [T64(x).requires_grad_(True) for x in lf], [T64(x).requires_grad_(True) for x in rf]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    out = oagg.aggregate(sd, lf64, rf64, T64(il), T64(ir), {}, cfg=dict(coarse=dict(num_sample=ns)), training=True)
    gt = T64(sc["gt"][0])
    l1 = [w * olo.smooth_l1_loss_per_level(olo.rescale_to_full(d, (H, W)), gt, max_disp) for w, d in zip((2.0, 1.0, 0.7, 0.5), out[0])]
    wd = [2.0 * w * olo.wasserstein_loss_per_level(c, o, s, gt, max_disp) for w, c, o, s in zip((1.0, 0.7, 0.5), out[1], out[3], out[2])]
    for i, v in enumerate(l1):
        assert abs(float(v) - float(g["f64::loss::l1_loss_lvl%d" % i])) < 1e-9 * abs(float(v)) + 1e-12
    for i, v in enumerate(wd):
        assert abs(float(v) - float(g["f64::loss::wars_loss_lvl%d" % i])) < 1e-9 * abs(float(v)) + 1e-12
    (sum(l1) + sum(wd)).backward()
    for i in range(3):
        for side, ts in (("left", lf64), ("right", rf64)):
            ref = torch.from_numpy(g["f64::g_%s_%d" % (side, i)])
            assert float((ts[i].grad[:, ::8] - ref).norm() / ref.norm()) < 1e-8, (side, i)
    keys = [str(k) for k in g["all_keys"]]
    scale = float(np.max(g["f64::all_norm"]))
    for k, n_ref, p_ref in zip(keys, g["f64::all_norm"], g["f64::all_proj"]):
        gr = sd[k].grad
        assert gr is not None, k
        p = float((gr.flatten() * torch.from_numpy(synth.normal(seed, "proj" + k, (gr.numel(),))).double()).sum())
        assert abs(float(gr.norm()) - n_ref) <= 1e-7 * n_ref + 1e-12 * scale, (k, float(gr.norm()), n_ref)
        assert abs(p - p_ref) <= 1e-7 * n_ref + 1e-12 * scale, (k, p, p_ref)
