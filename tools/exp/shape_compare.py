import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
cases = [(2, 544, 960, 12), (2, 384, 1248, 12), (2, 384, 1280, 12), (2, 512, 1024, 12), (8, 480, 640, 8), (4, 480, 640, 8), (1, 384, 1248, 12)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, H, W, ns in cases:
    seed = synth.SEED0 + 2
    net = bench.build_model(dev, seed, ns)
    inputs = bench.make_inputs(dev, seed, B, (H, W))
    bench.calibrate_batchnorm(net, inputs)
    res = []
    for overlap in (True, False):
        eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
        eng.net.overlap = overlap
        with torch.no_grad():
            for _ in range(4): eng(*inputs, {})
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): eng(*inputs, {})
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        res.append(dt * 1e3)
    print("B=%d %dx%d ns=%d: overlap %.3f ms (%.0f pairs/s, %.2f ns/px)  serial %.3f ms" % (B, H, W, ns, res[0], B / res[0] * 1e3, res[0] * 1e6 / (B * H * W), res[1]), flush=True)
    del net, inputs, eng
    torch.cuda.empty_cache()
