"""Temporal-sequence throughput at BASELINE.json configs [2]-[4]: per sequence of T frames,
frame 0 single-frame pass -> update_map (ts_reproject_memory_fwd) -> frame t temporal pass -> ...
State handed over exactly as the reference's wrapper does (projects/TemporalStereo/TemporalStereo.py:282-324):
prev_disp and the top-2 cost memory come out of the aggregation, update_map moves them into the next frame.

  python tools/sequence_bench.py [--configs 2 3 4] [--frames 2] [--iters 20]
Prints one JSON line per config (pairs/s = B*T / sequence time, features resident in HBM).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench          # noqa: E402
import synth          # noqa: E402

CONFIGS = {
    2: dict(name="FlyingThings3D 544x960 D=192 temporal, batch 4", B=4, H=544, W=960, num_sample=12, local=1),
    3: dict(name="TartanAir 480x640 D=128 temporal, batch 8", B=8, H=480, W=640, num_sample=8, local=3),
    4: dict(name="KITTI 384x1248 D=192 temporal, batch 2", B=2, H=384, W=1248, num_sample=12, local=3),
}


def run(idx, frames, iters, two_phase=True):
    from temporalstereo_amd import temporal
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    c = CONFIGS[idx]
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + idx
    B, H, W = c["B"], c["H"], c["W"]
    net = bench.build_model(dev, seed, c["num_sample"])
    inputs = bench.make_inputs(dev, seed, B, (H, W))
    bench.calibrate_batchnorm(net, inputs)
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind" if two_phase else "copy")
    K = torch.from_numpy(synth.sceneflow_intrinsics(B, H, W)).to(dev)
    T = torch.from_numpy(synth.small_motion(seed, B)).to(dev)
    Ti = torch.inverse(T)

    def update(info):
        return temporal.update_map(info, K, T, Ti, 0.54, H, W, use_past_cost=True, local_map_size=c["local"])

    def sequence():
        if not two_phase:
            out = eng(*inputs, {})
            for _ in range(frames - 1):
                out = eng(*inputs, update(out[5]))
            return out
        # two-phase: the state-independent half of frame t+1 is issued before frame t's state update, so that on the device
        # it overlaps frame t's 1/4-level tail and update_map (engine.begin / finish)
        out = eng.finish(eng.begin(*inputs), {})
        for _ in range(frames - 1):
            h = eng.begin(*inputs)
            out = eng.finish(h, update(out[5]))
        return out

    with torch.no_grad():
        for _ in range(3):
            sequence()

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        t_seq = timed(sequence, iters)
        first = eng(*inputs, {})
        state = update(dict(first[5]))
        t_single = timed(lambda: eng(*inputs, {}), iters)
        t_update = timed(lambda: update(dict(first[5])), iters)
        t_temporal = timed(lambda: eng(*inputs, state), iters)
    return dict(config=idx, workload=c["name"], frames=frames, batch=B, ms_per_sequence=t_seq * 1e3,
                pairs_per_s=B * frames / t_seq, ms_single_pass=t_single * 1e3, ms_update_map=t_update * 1e3,
                ms_temporal_pass=t_temporal * 1e3, local_map_size=c["local"],
                schedule="two-phase (begin/finish: frame t+1's state-independent half overlaps frame t's tail and update_map)" if two_phase
                else "one pass at a time")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="+", default=[2, 3, 4])
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--one-at-a-time", action="store_true")
    a = ap.parse_args()
    for i in a.configs:
        print(json.dumps(run(i, a.frames, a.iters, not a.one_at_a_time)), flush=True)
