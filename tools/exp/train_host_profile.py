"""Where the host time of the eager training step goes (cProfile over a few steps; the device runs behind the host)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd.train import TrainStep  # noqa: E402

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
gt = torch.from_numpy(synth.smooth(synth.normal(seed, "gt", (1, 1, 544, 960))) * 20.0 + 70.0).to(dev)
K = torch.from_numpy(synth.sceneflow_intrinsics(1, 544, 960)).to(dev)
T = torch.from_numpy(synth.small_motion(seed, 1)).to(dev)
eye = torch.eye(4, device=dev).expand(1, 4, 4).contiguous()
poses = [(eye, eye), (T, eye)]
step = TrainStep(net, graph=False)
for _ in range(5):
    step(frames, gt, K, poses)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(frames, gt, K, poses)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
