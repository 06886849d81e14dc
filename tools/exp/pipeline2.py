import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, B); bench.calibrate_batchnorm(net, inputs)
ref_eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
ref = [t.clone() for t in ref_eng(*inputs, {})[0]]
for depth in (1, 2, 3):
    eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth)
    for _ in range(6): out = eng(*inputs, {})
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out[0], ref))
    t0 = time.perf_counter()
    for _ in range(100): out = eng(*inputs, {})
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    bad = 0
    for i in range(50):
        out = eng(*inputs, {})
        if i % 7 == 0:
            torch.cuda.synchronize()
            bad += 0 if all(torch.equal(a, b) for a, b in zip(out[0], ref)) else 1
    torch.cuda.synchronize()
    print("pipeline=%d batch %d: %.3f ms per pass = %.0f pairs/s; outputs identical to the plain engine: %s (%d mismatching checks)"
          % (depth, B, dt * 1e3, B / dt, same, bad), flush=True)
