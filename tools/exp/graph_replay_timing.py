"""Experiment: where does a replayed training step spend its time -- host time inside CUDAGraph.replay() vs. device time between
events around it, with and without the weight-layout arena (TS_TRAIN_LAYOUT_ARENA)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.exp.dbg_train_graph import setup, MAX_DISP
from temporalstereo_amd.train import TrainStep

net, frames, gt, K, poses = setup()
step = TrainStep(net, max_disp=MAX_DISP, local_map_size=1, graph=True)
for _ in range(5):
    step(frames, gt, K, poses)
torch.cuda.synchronize()
g = step._g
host, dev = [], []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    host.append((t1 - t0) * 1e3)
    dev.append(e0.elapsed_time(e1))
print("arena", os.environ.get("TS_TRAIN_LAYOUT_ARENA", "1"), "replay() host ms %.2f" % (sum(host) / len(host)), "device ms %.2f" % (sum(dev) / len(dev)),
      "nodes?", len(step.layouts.entries))
