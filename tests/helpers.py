"""Shared helpers for tests (not collected)."""
import json
import os

import numpy as np
import torch

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def t(x, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(x)).to(device)


def state_shapes(tag):
    with open(os.path.join(GOLDEN, "state_shapes_%s.json" % tag)) as fh:
        return {k: tuple(v) for k, v in json.load(fh).items()}


def dims_from_golden(g):
    d = [int(v) for v in g["dims"]]
    return dict(coarse=dict(in_planes=d[0], C=d[1], num_sample=d[2]), fine=dict(in_planes=d[3], C=d[4]),
                precise=dict(in_planes=d[5], C=d[6]))


def shapes_tag(dims):
    return "%dx%dx%d" % (dims['coarse']['C'], dims['fine']['C'], dims['precise']['C'])


def synth_state(dims, seed, device="cpu", golden=None):
    """Synthetic weights; BatchNorm running statistics come from the fixture when it carries the
    calibrated ones (keys 'bn::<name>', see tools/gen_golden.py)."""
    vals = synth.state_values(state_shapes(shapes_tag(dims)), seed)
    if golden is not None:
        for k, v in golden.items():
            if k.startswith("bn::"):
                vals[k[4:]] = v
    return {k: t(v, device) for k, v in vals.items()}


def aggregator_inputs(g, dims, device="cpu"):
    seed, B, H, W = int(g["seed"]), int(g["B"]), int(g["H"]), int(g["W"])
    chans = (dims['precise']['in_planes'], dims['fine']['in_planes'], dims['coarse']['in_planes'])
    lf, rf = synth.feature_pyramid(seed, B, H, W, chans=chans)
    il, ir = synth.images(seed, B, H, W)
    prev = {}
    if int(g["temporal"]):
        prev = {'cost_memory': {'disp_sample': t(g["mem_disp_sample"], device),
                                'cost_volume': t(g["mem_cost_volume"], device)},
                'use_past_cost': True, 'local_map': t(g["local_map"], device),
                'local_map_size': int(g["local_map"].shape[1])}
    return [t(x, device) for x in lf], [t(x, device) for x in rf], t(il, device), t(ir, device), prev


def epe(a, b):
    return float((a.double() - b.double()).abs().mean())
