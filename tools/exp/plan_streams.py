"""Which launches of a recorded pipelined pass go to which stream (the Recorder's log), in issue order per stream."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth  # noqa: E402
from temporalstereo_amd.aggregation.engine import InferenceEngine  # noqa: E402

dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.load_trained(bench.build_model(dev, seed)).eval()
inputs, _ = bench.make_planted_inputs(dev, seed, 1)
eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=3)
with torch.no_grad():
    eng(*inputs, {})
torch.cuda.synchronize()
cap = next(iter(eng._graphs.values()))
log = cap.rec.log if hasattr(cap, "rec") else cap.recorder.log
main = torch.cuda.current_stream().cuda_stream
names = {main: "caller", eng.net.fast.cuda_stream: "fast", eng.net.aux.cuda_stream: "aux"}
per = collections.OrderedDict()
for name, st in log:
    per.setdefault(names.get(st, hex(st)), []).append(name)
for k, v in per.items():
    c = collections.Counter(v)
    print(k, len(v), dict(c))
