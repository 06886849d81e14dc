"""Workload of the bench line: BASELINE.json configs[1] -- FlyingThings3D 540x960 run at 544x960 (as the reference does,
projects/TemporalStereo/configs/sceneflow.yaml:84-85), D=192 (COARSE.NUM_SAMPLE = 12), sceneflow.yaml channel widths."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import synth  # noqa: E402  (deterministic synthetic inputs, shared with the tests)

RUN_H, RUN_W = 544, 960            # 540x960 resized to a multiple of 16 (datasets/base.py:176-185)
MAX_DISP = 192
HBM_PEAK = 8.0e12                  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
DIMS = dict(coarse=dict(in_planes=256, C=32, num_sample=MAX_DISP // 16), fine=dict(in_planes=128, C=16),
            precise=dict(in_planes=64, C=8))
CKPT = os.path.join(ROOT, "tests", "golden", "ckpt_planted.npz")


def k1_algorithmic_bytes(B, C, H, W, D, sampled):
    """inputs once + output once, fp32.  sampled == "warped": the inference form that leaves out
    the D-fold repeat of the left features (ts_block_cost_sampled_warped_fwd); "corr": the correlation blocks alone."""
    if sampled == "warped":
        return 4 * B * H * W * (2 * C + D + (C + 3 * C // 8) * D)
    if sampled == "corr":
        return 4 * B * H * W * (2 * C + D + (3 * C // 8) * D)
    if sampled:
        return 4 * B * H * W * (2 * C + D + (2 * C + 3 * C // 8) * D)
    return 4 * B * H * W * (2 * C + (C + 3 * C // 8) * D)


def build_model(dev, seed, num_sample=None):
    import temporalstereo_amd as ts
    net = ts.TEMPORALSTEREO(
        coarse=ts.CoarseAggregation(DIMS['coarse']['in_planes'], DIMS['coarse']['C'],
                                    num_sample or DIMS['coarse']['num_sample']),
        fine=ts.FineAggregation(DIMS['fine']['in_planes'], DIMS['fine']['C'], 5),
        precise=ts.PreciseAggregation(DIMS['precise']['in_planes'], DIMS['precise']['C'], 5))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.state_values(shapes, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()}, strict=True)
    return net.to(dev)


def make_inputs(dev, seed, B, hw=None):
    H, W = hw or (RUN_H, RUN_W)
    chans = (DIMS['precise']['in_planes'], DIMS['fine']['in_planes'], DIMS['coarse']['in_planes'])
    lf, rf = synth.feature_pyramid(seed, B, H, W, chans=chans)
    il, ir = synth.images(seed, B, H, W)
    to = lambda a: torch.from_numpy(a).to(dev)
    return [to(x) for x in lf], [to(x) for x in rf], to(il), to(ir)


def load_trained(net):
    """The committed checkpoint (tools/train_checkpoint.py: this repository's TrainStep on planted-disparity scenes)."""
    with np.load(CKPT) as z:
        net.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    return net


def make_planted_inputs(dev, seed, B, hw=None):
    """One frame of a planted-disparity scene (tests/synth.stereo_sequence)
    -> ((left_feats, right_feats, left_image, right_image), gt)."""
    H, W = hw or (RUN_H, RUN_W)
    sc = synth.stereo_sequence(seed, B, H, W, frames=1, max_disp=MAX_DISP)
    lf, rf, il, ir = sc["frames"][0]
    to = lambda a: torch.from_numpy(a).to(dev)
    return ([to(x) for x in lf], [to(x) for x in rf], to(il), to(ir)), torch.from_numpy(sc["gt"][0])


def calibrate_batchnorm(net, inputs, prev_info=None):
    """One train-mode pass with momentum 1: running statistics := this input's batch statistics, so the random-weight
    network is conditioned like a trained one (same protocol as tools/gen_golden.py).  prev_info: temporal state of the
    frame."""
    bns = [m for m in net.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))]
    for m in bns:
        m.momentum = 1.0
    net.train(True)
    with torch.no_grad():
        net(*inputs, dict(prev_info or {}))
    for m in bns:
        m.momentum = 0.1
    net.train(False)


def timed_us(fn, n=100, warm=10):
    """Mean duration (us) of `fn`, n calls back to back between ONE pair of HIP events on the current stream (the queue
    stays full: the events see kernel time, not the host gap in front of every launch)."""
    with torch.no_grad():
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
