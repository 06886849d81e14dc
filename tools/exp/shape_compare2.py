import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation import native
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
cases = [(1, 544, 960, 12), (2, 544, 960, 12), (4, 544, 960, 12), (8, 480, 640, 8), (2, 384, 1248, 12)]
orig_cap = native._chunk_cap
def timeit(net, inputs, overlap):
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
    eng.net.overlap = overlap
    with torch.no_grad():
        for _ in range(4): eng(*inputs, {})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): eng(*inputs, {})
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3
for B, H, W, ns in cases:
    seed = synth.SEED0 + 2
    net = bench.build_model(dev, seed, ns)
    inputs = bench.make_inputs(dev, seed, B, (H, W))
    bench.calibrate_batchnorm(net, inputs)
    out = []
    out.append(("overlap cap8", timeit(net, inputs, True)))
    for cap in (16, 32):
        native._chunk_cap = lambda c, cap=cap: orig_cap(cap if c == 8 else c)
        out.append(("overlap cap%d" % cap, timeit(net, inputs, True)))
    native._chunk_cap = orig_cap
    kinds = native._PAR["kinds"]
    native._PAR["kinds"] = frozenset()
    out.append(("overlap no-branches cap8", timeit(net, inputs, True)))
    native._PAR["kinds"] = kinds
    for cap in (8, 32):
        orig_cap(cap)
        native._chunk_cap = lambda c: None
        out.append(("serial cap%d" % cap, timeit(net, inputs, False)))
    native._chunk_cap = orig_cap; orig_cap(32)
    print("B=%d %dx%d: " % (B, H, W) + "  ".join("%s %.3f" % kv for kv in out), flush=True)
