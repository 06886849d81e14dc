"""Experiment: host time of one native plan replay (ts_plan_run) against the GPU time of the pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import build_model, make_inputs, calibrate_batchnorm, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine

dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = build_model(dev, seed)
inputs = make_inputs(dev, seed, 1)
calibrate_batchnorm(net, inputs)
for depth in (1, 3):
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth)
    with torch.no_grad():
        for _ in range(2 * depth + 2):
            eng(*inputs, {})
        torch.cuda.synchronize()
        # host-only: issue 20 passes back to back, time the issue loop; the device queue is deep enough not to push back
        t0 = time.perf_counter()
        for _ in range(20):
            eng(*inputs, {})
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    caps = list(eng._graphs.values())
    print("pipeline %d: host issue %.3f ms per pass, wall %.3f ms per pass, plan length %s" % (depth, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, [len(c.plan) if hasattr(c, "plan") else "?" for c in caps][:3]))
