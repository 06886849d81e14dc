"""Host enqueue time vs device time of one native aggregation pass, per replay mode."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, 1); bench.calibrate_batchnorm(net, inputs)
for replay, ov in (("plan", True), ("plan", False), ("eager", True), ("eager", False)):
    eng = InferenceEngine(net, backend="native", replay=replay)
    eng.net.overlap = ov
    for _ in range(5): eng(*inputs, {})
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): eng(*inputs, {})
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("replay=%s overlap=%s host enqueue %.3f ms/pass, total %.3f ms/pass" % (replay, ov, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
