"""Host time of the eager training step split by where it is spent: inside the fused wrappers' forward / backward bodies (functional._ConvBNAct),
inside the other autograd Functions of functional.py, and the rest (module walk, autograd engine, torch ops, optimizer)."""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd import functional as TF  # noqa: E402
from temporalstereo_amd.train import TrainStep  # noqa: E402

acc = collections.defaultdict(lambda: [0.0, 0])


def wrap(cls, name):
    fn = getattr(cls, name)
    raw = fn.__func__ if hasattr(fn, "__func__") else fn

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            e = acc[(cls.__name__, name)]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(cls, name, staticmethod(timed))


for v in list(vars(TF).values()):
    if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
        wrap(v, "forward"); wrap(v, "backward")

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
for _ in range(8):
    step(frames, gt, K, poses)
torch.cuda.synchronize()
acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N):
    step(frames, gt, K, poses)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print("step %.2f ms (with the timing wrappers)" % (wall * 1e3))
tot = 0.0
for (c, n), (t, k) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %-9s %6.0f calls/step  %7.3f ms/step  %6.1f us/call" % (c, n, k / N, t / N * 1e3, t / k * 1e6))
    tot += t / N
print("inside Function bodies: %.2f ms/step; everything else (engine, module walk, torch ops, optimizer, waiting): %.2f ms" % (tot * 1e3, (wall - tot) * 1e3))
