"""Run the native pass repeatedly on the same inputs: every output must be bit-identical each time."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, 1); bench.calibrate_batchnorm(net, inputs)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for replay in ("plan", "eager"):
    eng = InferenceEngine(net, backend="native", replay=replay, inputs="bind")
    out = eng(*inputs, {})
    ref = [t.clone() for t in out[0]] + [t.clone() for t in out[1]]
    bad = 0
    for i in range(N):
        out = eng(*inputs, {})
        cur = list(out[0]) + list(out[1])
        if i % 8 == 0 or True:
            diffs = [float((a - b).abs().max()) for a, b in zip(cur, ref)]
            if any(d != 0.0 for d in diffs):
                bad += 1
                if bad <= 5: print(replay, "iteration", i, "max abs diffs", diffs, flush=True)
    print("replay=%s: %d / %d iterations differed" % (replay, bad, N), flush=True)

# three passes in flight: snapshots are taken on the caller's stream (ordered after each pass's tail) without ever
# synchronising the host, so consecutive passes really overlap while they are checked
eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=3)
ref_full = InferenceEngine(net, backend="native", replay="plan", inputs="bind")(*inputs, {})
ref_t = [t.clone() for t in ref_full[0]] + [t.clone() for t in ref_full[1]]
for _ in range(4): eng(*inputs, {})
snaps = []
for i in range(N):
    out = eng(*inputs, {})
    snaps.append([t.clone() for t in list(out[0]) + list(out[1])])
    if len(snaps) == 50:
        torch.cuda.synchronize()
        bad = sum(1 for sn in snaps if any(not torch.equal(a, b) for a, b in zip(sn, ref_t)))
        if bad: print("pipeline=3: %d of 50 snapshots differ (iteration %d)" % (bad, i), flush=True)
        snaps = []
torch.cuda.synchronize()
print("pipeline=3: %d overlapped passes checked against the plain engine" % N, flush=True)
