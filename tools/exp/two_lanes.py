"""Experiment: N independent three-stage pipelines (engines with private helper streams) fed alternately, against one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from bench import build_model, make_inputs, calibrate_batchnorm, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine

dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
B = int(os.environ.get("BATCH", "1"))
net = build_model(dev, seed)
inputs = make_inputs(dev, seed, B)
calibrate_batchnorm(net, inputs)
for lanes in (1, 2, 3):
    for depth in (2, 3):
        engs = [InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth, private_streams=(lanes > 1)) for _ in range(lanes)]
        # each lane on its own caller stream
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(lanes - 1)]
        def step(i):
            k = i % lanes
            with torch.cuda.stream(streams[k]), torch.no_grad():
                return engs[k](*inputs, {})
        for i in range(lanes * depth * 2):
            step(i)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("lanes %d depth %d batch %d: %.1f pairs/s" % (lanes, depth, B, n * B / el))
        del engs
