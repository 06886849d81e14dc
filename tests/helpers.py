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


def temporal_update_from_golden(g, mod, device="cpu"):
    """Runs `mod.update_map` (oracle.temporal or temporalstereo_amd.temporal) on the inputs of a
    tests/golden/temporal_update_*.npz fixture (made by the reference's own update_map, tools/gen_golden.py)."""
    to = lambda a: t(a, device)
    info = {"prev_disp": to(g["prev_disp"]),
            "cost_memory": {"disp_sample": to(g["mem_disp_sample"]), "cost_volume": to(g["mem_cost_volume"])}}
    if int(g["has_local_in"]):
        info["local_map"] = to(g["local_map_in"])
    if int(g["composed"]):
        info["T_past_to_now"] = torch.bmm(to(g["T_now"]), to(g["inv_T_past"]))
    H, W = (int(v) for v in g["full_hw"])
    return mod.update_map(info, to(g["K"]), to(g["T_now"]), to(g["inv_T_past"]), to(g["baseline"]), H, W,
                          use_past_cost=bool(int(g["use_past_cost"])), local_map_size=int(g["local_map_size"]))


def check_temporal_update(g, info, tol_mean, tol_far):
    def close(a, b, what):
        a = a.detach().cpu().double()
        b = torch.from_numpy(b).double()
        assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
        d = (a - b).abs()
        # targets that receive almost no splat weight amplify rounding (x / (den + 1e-22)): bulk + rare outliers
        assert float(d.mean()) < tol_mean and float((d > 1e-2 * (1 + b.abs())).double().mean()) < tol_far, \
            (what, float(d.mean()), float(d.max()))
    if int(g["has_memory_out"]):
        close(info["cost_memory"]["disp_sample"], g["out_disp_sample"], "disp_sample")
        close(info["cost_memory"]["cost_volume"], g["out_cost_volume"], "cost_volume")
    else:
        assert info.get("cost_memory") is None
    if int(g["has_local_out"]):
        close(info["local_map"], g["out_local_map"], "local_map")
        assert info["local_map_size"] == int(g["local_map_size"])
    assert info["use_past_cost"] == bool(int(g["use_past_cost"]))


def multi_rank_env(world, **extra):
    """Environment of a `world`-rank GPU sub-process launch, chosen by what the box has:

    * >= `world` devices: ONE RANK PER DEVICE on RCCL (backend "nccl", device = LOCAL_RANK) -- so that a `-m gpu` run on a multi-GPU
      node exercises RCCL over xGMI and the peer mailboxes across a device boundary by itself (the reference's arrangement:
      pl.Trainer(strategy='ddp'), projects/TemporalStereo/dist_train.py:82-96);
    * fewer: every rank on device 0 with gloo carrying the collectives (RCCL refuses two ranks on one device): TS_BENCH_BACKEND /
      TS_BENCH_DEVICE, the hooks bench.py and the workers read.

    Returns (env, arrangement) with arrangement 'one rank per device, nccl' | 'all ranks on device 0, gloo'."""
    env = dict(os.environ, MIOPEN_FIND_MODE="2", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TS_BENCH_BACKEND", "TS_BENCH_DEVICE"):
        env.pop(k, None)
    forced = os.environ.get("TS_TEST_MULTI_RANK", "")            # "one_device" forces the gloo arrangement on a multi-GPU box
    if torch.cuda.device_count() >= world and forced != "one_device":
        arrangement = "one rank per device, nccl"
    else:
        env.update(TS_BENCH_BACKEND="gloo", TS_BENCH_DEVICE="0")
        arrangement = "all ranks on device 0, gloo"
    env.update({k: str(v) for k, v in extra.items()})
    return env, arrangement
