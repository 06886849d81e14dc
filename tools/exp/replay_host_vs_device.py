"""Replayed training step: how long the HOST spends inside graph.replay() against the step's wall time (GPU box only).
If the two are equal the replay is bound by the host issuing the graph's nodes, not by the kernels."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.train import TrainStep
dev = torch.device("cuda:0"); seed = synth.SEED0 + 2
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
step = TrainStep(net, max_disp=bench.MAX_DISP, local_map_size=1, graph=True, sync_bn=False)
step(frames, gt, K, poses)
frames, gt, K, poses = step.bound_inputs()
for _ in range(5): step(frames, gt, K, poses)
torch.cuda.synchronize()
g = step._g
# (a) replay alone: host time of the call, and wall time to completion
for name, n in (("replay only", 20),):
    host = wall = 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.replay(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host += t1 - t0; wall += t2 - t0
    print("%s: host inside replay() %.2f ms, until the device is done %.2f ms" % (name, host / n * 1e3, wall / n * 1e3))
# (b) back-to-back replays (the queue never drains)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("20 replays back to back: host %.2f ms per replay, wall %.2f ms per replay" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
print("graph nodes:", "n/a")
