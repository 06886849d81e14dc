"""Measurements next to the headline (none of them changes `value`): what one caller sees, the engine without the
bf16-split convolutions, independent lanes, temporal sequences."""
import os
import sys
import time

import torch

from .common import ROOT


def one_pass_at_a_time(net, inputs, steps, batch):
    """The same engine with one pass in flight: every pass waits for the previous one (what a latency-bound caller
    sees)."""
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    single = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
    with torch.no_grad():
        for _ in range(3):
            single(*inputs, {})
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            single(*inputs, {})
        torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t1) / steps
    return dict(value=batch / dt1, unit="pairs/s per GPU", ms_per_step=dt1 * 1e3,
                note="same engine with frames_in_flight=1 on rank 0: every pass waits for the previous one")


def f32_mfma_only(net, inputs, steps, batch, depth):
    """The pipelined engine with every convolution on the f32-input MFMA kernel (TS_CONV_X6=0 semantics), measured in
    this run: the headline uses ts_conv3d_hw_x6_fwd / _x6s_fwd where a layer allows it (fp32 products from six bf16 MFMA
    products)."""
    from temporalstereo_amd.aggregation import native as _N
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    if not _N.X6:
        return None
    _N.X6 = False
    try:
        eng32 = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth)
        with torch.no_grad():
            for _ in range(depth + 5):
                eng32(*inputs, {})
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                eng32(*inputs, {})
            torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t1) / steps
        return dict(value=batch / dt32, unit="pairs/s per GPU", ms_per_step=dt32 * 1e3)
    finally:
        _N.X6 = True


def concurrent_lanes(net, dev, dist, world, more_inputs, seed, rank, inflight, steps, batch):
    """Serving-style concurrency: N independent batch-1 passes in flight on one GPU (each its own plan, buffers and
    streams)."""
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    lanes = []
    for i in range(inflight):
        st = torch.cuda.Stream(device=dev)
        ins = more_inputs(seed + 100 * (i + 1) + rank)
        with torch.cuda.stream(st):
            eng = InferenceEngine(net, backend="native", replay="plan", inputs="bind", private_streams=True)
            with torch.no_grad():
                for _ in range(3):
                    eng(*ins, {})
        lanes.append((st, eng, ins))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for k in range(steps):
            st, eng, ins = lanes[k % inflight]
            with torch.cuda.stream(st):
                eng(*ins, {})
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return dict(inflight=inflight, value=world * batch * steps / el, unit="pairs/s", ms_per_step=el / steps * 1e3,
                note="%d independent batch-1 passes in flight per GPU (own launch plan, buffers and streams each); not "
                     "the headline value" % inflight)


def sequence_leg(iters=10):
    """Temporal-sequence throughput at BASELINE configs[2]-[4] (their stated batches, T=2; tools/sequence_bench.py)."""
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import sequence_bench
    return [sequence_bench.run(i, 2, iters) for i in (2, 3, 4)]
