import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
B, H, W, ns = 1, 544, 960, 12
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed, ns)
inputs = bench.make_inputs(dev, seed, B, (H, W))
bench.calibrate_batchnorm(net, inputs)
def timeit(eng, n=20):
    with torch.no_grad():
        for _ in range(4): eng(*inputs, {})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): eng(*inputs, {})
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
engs = []
for i in range(10):
    if len(sys.argv) > 1 and i in (3, 6):
        dummy = torch.cuda.Stream(device=dev, priority=-1)      # shifts the pool index by one
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
    t = timeit(eng)
    na = eng.net
    print("engine %d: %.3f ms  fast=%#x aux=%#x ids %s %s" % (i, t, na.fast.cuda_stream, na.aux.cuda_stream, na.fast.stream_id, na.aux.stream_id), flush=True)
    engs.append(eng)
print("re-time all engines:", " ".join("%.3f" % timeit(e) for e in engs))
