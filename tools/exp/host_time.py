"""Host enqueue time vs device time of one native aggregation pass."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation import native
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, 1); bench.calibrate_batchnorm(net, inputs)
agg = native.NativeAggregator(net)
for ov in (True, False):
    agg.overlap = ov
    for _ in range(5): agg(*inputs, {})
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): agg(*inputs, {})
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("overlap=%s host enqueue %.3f ms/pass, total %.3f ms/pass" % (ov, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
