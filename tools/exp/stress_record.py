"""Repeat the record -> replay-on-new-inputs sequence of tests/test_native_gpu.py on the tiny golden case."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load, dims_from_golden, aggregator_inputs
from test_aggregator_gpu import _build
from temporalstereo_amd.aggregation.engine import InferenceEngine
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for name in ("agg_tiny_single", "agg_tiny_temporal"):
    g = load(name); dims = dims_from_golden(g)
    net = _build(dims, int(g["seed"]), dev, golden=g)
    lf, rf, il, ir, prev = aggregator_inputs(g, dims, dev)
    lf2 = [x.flip(-1).contiguous() for x in lf]; rf2 = [x.flip(-1).contiguous() for x in rf]
    il2, ir2 = il.flip(-1).contiguous(), ir.flip(-1).contiguous()
    for it in range(N):
        plan = InferenceEngine(net, backend="native", replay="plan")
        eager = InferenceEngine(net, backend="native", replay="eager")
        plan(lf, rf, il, ir, dict(prev))
        want = [t.clone() for t in eager(lf2, rf2, il2, ir2, dict(prev))[0]]
        got = plan(lf2, rf2, il2, ir2, dict(prev))[0]
        d = [float((a - b).abs().max()) for a, b in zip(got, want)]
        if any(x > 1e-4 for x in d):
            bad += 1; print(name, it, d, flush=True)
print("mismatches:", bad)
