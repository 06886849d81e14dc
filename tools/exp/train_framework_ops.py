#!/usr/bin/env python
"""Which lines of the package still launch FRAMEWORK kernels in a training step (GPU box only).

One eager TrainStep (T=2, batch 1, config-2 sizes) under torch.profiler with Python stacks; every aten operator that launched a
device kernel is listed with the innermost frame of this repository on its stack (forward operators) or with the autograd node that
ran it (backward operators).  usage: python tools/exp/train_framework_ops.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd.train import TrainStep  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 2
    net = bench.build_model(dev, seed)
    frames = []
    for t in range(2):
        lf, rf, il, ir = bench.make_inputs(dev, seed + 1000 * t, 1)
        if t == 1:
            lf, rf = [x.requires_grad_(True) for x in lf], [x.requires_grad_(True) for x in rf]
        frames.append((lf, rf, il, ir))
    bench.calibrate_batchnorm(net, frames[0])
    gt = torch.from_numpy(synth.smooth(synth.normal(seed, "gt", (1, 1, bench.RUN_H, bench.RUN_W))) * 20.0 + 70.0).to(dev)
    K = torch.from_numpy(synth.sceneflow_intrinsics(1, bench.RUN_H, bench.RUN_W)).to(dev)
    T = torch.from_numpy(synth.small_motion(seed, 1)).to(dev)
    eye = torch.eye(4, device=dev).expand(1, 4, 4).contiguous()
    poses = [(eye, eye), (T, eye)]
    step = TrainStep(net, max_disp=bench.MAX_DISP, local_map_size=1, graph=False, sync_bn=False)
    for _ in range(3):
        step(frames, gt, K, poses)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step(frames, gt, K, poses)
        torch.cuda.synchronize()
    rows = collections.Counter()
    shapes = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::") or not ev.kernels:
            continue
        if any(c.kernels for c in ev.cpu_children if c.name.startswith("aten::")):
            continue                                  # a wrapper operator: its child is listed
        site = None
        for fr in ev.stack or []:
            if "/repo/" in fr or "temporalstereo_amd" in fr or "bench.py" in fr:
                site = fr.split("/")[-1] if "/" in fr else fr
                break
        if site is None:
            p = ev.cpu_parent
            while p is not None and site is None:
                if "Backward" in p.name or "autograd" in p.name or "AccumulateGrad" in p.name:
                    site = "<backward> " + p.name
                p = p.cpu_parent
        key = (ev.name, site or "?", str(ev.input_shapes)[:90])
        rows[key] += len(ev.kernels)
    total = sum(rows.values())
    print("framework kernel launches in one eager step: %d" % total)
    for (name, site, shp), n in sorted(rows.items(), key=lambda kv: (-kv[1], kv[0])):
        print("%3d  %-28s %-70s %s" % (n, name, site, shp))


if __name__ == "__main__":
    main()
