#!/usr/bin/env python
"""Headline benchmark: stereo pairs/s of the cost-volume aggregation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode native|native-eager|native-graph|module|module-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (coarse -> fine -> precise aggregation: cost-volume build,
3-D aggregation pyramid, top-k soft-argmax regression, upsamplers) over one batch of synthetic
FlyingThings3D-shaped inputs already resident in HBM: BASELINE.json configs[1] = 540x960 run at
544x960 (as the reference does, sceneflow.yaml:84-85), D=192 (COARSE.NUM_SAMPLE=12), single frame,
batch 1 per GPU, fp32, eval mode.  Ranks are independent replicas (stereo pairs shard with no
data-path collective): value = pairs all ranks processed / max-over-ranks time ("weak" scaling).

Prints ONE JSON line (rank 0) with the extra objects
  roofline     cost-volume build at the 1/4 level (the dominant K1 launch): algorithmic bytes
                / its mean duration measured with HIP events on the launch
               stream inside the timed region, against the 8.0 TB/s HBM peak
  cpu_baseline the CPU oracle (oracle/, a port of the reference's torch CPU path) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

# the one framework pass this script makes (BatchNorm calibration) should not trigger MIOpen's
# exhaustive solver search (seconds of naive-kernel benchmarking that would drown a profile)
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# hipGraph replays of the training step (train-graph leg): explicit opt-in, before anything touches the GPU (temporalstereo_amd/train.py)
from temporalstereo_amd import train as _ts_train  # noqa: E402
try:
    _ts_train.enable_graph_replay()
except RuntimeError:          # imported as a module by a process that already used the GPU: only the train-graph leg needs it
    pass

import synth  # noqa: E402  (deterministic synthetic inputs, shared with the tests)

RUN_H, RUN_W = 544, 960            # 540x960 resized to a multiple of 16 (datasets/base.py:176-185)
MAX_DISP = 192
HBM_PEAK = 8.0e12                  # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
DIMS = dict(coarse=dict(in_planes=256, C=32, num_sample=MAX_DISP // 16), fine=dict(in_planes=128, C=16),
            precise=dict(in_planes=64, C=8))


def k1_algorithmic_bytes(B, C, H, W, D, sampled):
    """inputs once + output once, fp32.  sampled == "warped": the inference
    form that leaves out the D-fold repeat of the left features (ts_block_cost_sampled_warped_fwd)."""
    if sampled == "warped":
        return 4 * B * H * W * (2 * C + D + (C + 3 * C // 8) * D)
    if sampled == "corr":           # the correlation blocks alone (ts_block_cost_sampled_corr_fwd)
        return 4 * B * H * W * (2 * C + D + (3 * C // 8) * D)
    if sampled:
        return 4 * B * H * W * (2 * C + D + (2 * C + 3 * C // 8) * D)
    return 4 * B * H * W * (2 * C + (C + 3 * C // 8) * D)


def build_model(dev, seed, num_sample=None):
    import temporalstereo_amd as ts
    net = ts.TEMPORALSTEREO(
        coarse=ts.CoarseAggregation(DIMS['coarse']['in_planes'], DIMS['coarse']['C'], num_sample or DIMS['coarse']['num_sample']),
        fine=ts.FineAggregation(DIMS['fine']['in_planes'], DIMS['fine']['C'], 5),
        precise=ts.PreciseAggregation(DIMS['precise']['in_planes'], DIMS['precise']['C'], 5))
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    vals = synth.state_values(shapes, seed)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()}, strict=True)
    return net.to(dev)


def make_inputs(dev, seed, B, hw=None):
    H, W = hw or (RUN_H, RUN_W)
    chans = (DIMS['precise']['in_planes'], DIMS['fine']['in_planes'], DIMS['coarse']['in_planes'])
    lf, rf = synth.feature_pyramid(seed, B, H, W, chans=chans)
    il, ir = synth.images(seed, B, H, W)
    to = lambda a: torch.from_numpy(a).to(dev)
    return [to(x) for x in lf], [to(x) for x in rf], to(il), to(ir)


CKPT = os.path.join(ROOT, "tests", "golden", "ckpt_planted.npz")


def load_trained(net):
    """The committed checkpoint (tools/train_checkpoint.py: this repository's TrainStep on planted-disparity scenes)."""
    with np.load(CKPT) as z:
        net.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
    return net


def make_planted_inputs(dev, seed, B, hw=None):
    """One frame of a planted-disparity scene (tests/synth.stereo_sequence) -> ((left_feats, right_feats, left_image, right_image), gt)."""
    H, W = hw or (RUN_H, RUN_W)
    sc = synth.stereo_sequence(seed, B, H, W, frames=1, max_disp=MAX_DISP)
    lf, rf, il, ir = sc["frames"][0]
    to = lambda a: torch.from_numpy(a).to(dev)
    return ([to(x) for x in lf], [to(x) for x in rf], to(il), to(ir)), torch.from_numpy(sc["gt"][0])


def calibrate_batchnorm(net, inputs, prev_info=None):
    """One train-mode pass with momentum 1: running statistics := this input's batch statistics, so
    the random-weight network is conditioned like a trained one (same protocol as tools/gen_golden.py).
    prev_info: temporal state of the frame (a temporal model's statistics come from temporal frames)."""
    bns = [m for m in net.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))]
    for m in bns:
        m.momentum = 1.0
    net.train(True)
    with torch.no_grad():
        net(*inputs, dict(prev_info or {}))
    for m in bns:
        m.momentum = 0.1
    net.train(False)


class K1Probe:
    """HIP events (torch.cuda.Event on the launch stream) bracketing exactly the K1 C-ABI call.

    Capture phase: during the timed steps it only remembers the arguments of each distinct K1 call.
    Measure phase: after the timed steps the same calls are replayed back-to-back (the queue stays
    full, so the events see kernel time, not host launch gaps) with one event pair per call."""

    def __init__(self):
        from temporalstereo_amd import functional as TF
        self.TF = TF
        self.calls = {}            # key -> (left, right, disp_or_int, scales)
        self.records = []
        self.timing = False

    def __enter__(self):
        self.TF._k1_probe = self._probe
        self._orig = {"block_cost": self.TF.block_cost, "block_cost_warped": self.TF.block_cost_warped,
                      "block_cost_corr": self.TF.block_cost_corr}

        def remember(name):
            orig = self._orig[name]

            def f(reference_fm, target_fm, disp_sample, block_cost_scale=3):
                B, C, H, W = reference_fm.shape
                sampled = not isinstance(disp_sample, int)
                kind = "warped" if name == "block_cost_warped" else ("corr" if name == "block_cost_corr" else sampled)
                key = (B, C, H, W, disp_sample.shape[1] if sampled else disp_sample, kind)
                if key not in self.calls:
                    self.calls[key] = (orig, reference_fm.detach(), target_fm.detach(),
                                       disp_sample.detach() if sampled else disp_sample, block_cost_scale)
                return orig(reference_fm, target_fm, disp_sample, block_cost_scale)
            return f
        for name in self._orig:          # aggregation.levels / aggregation.native reach the ops as TF.<name>
            setattr(self.TF, name, remember(name))
        return self

    def __exit__(self, *exc):
        self.TF._k1_probe = None
        for name, fn in self._orig.items():
            setattr(self.TF, name, fn)

    def _probe(self, key, launch):
        if not self.timing:
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = launch()
        e.record()
        self.records.append((key, s, e))
        return rc

    def measure(self, iters):
        """Mean duration of each remembered K1 launch: the C-ABI entry point on the pipeline's own input tensors with
        preallocated output / workspace, `iters` launches back to back between ONE pair of HIP events on the launch stream (an
        event pair per launch would time the host gap in front of every launch; the queue stays full this way).  The rocprofv3
        kernel trace of this command gives the same figure as main kernel + expansion kernel (profiles/)."""
        from temporalstereo_amd import _lib
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        calls = dict(self.calls)
        for key, (fn, l, r, d, sc) in self.calls.items():
            if key[5] in ("warped", "corr"):     # also time the complete op on the same tensors: SURVEY.md section 8(d)'s
                calls.setdefault(key[:5] + (True,), (None, l, r, d, sc))            # unfused-boundary figure
            if key[5] == "corr":                 # ... and the variant of rounds 1-3 (volume without its reference half)
                calls.setdefault(key[:5] + ("warped",), (None, l, r, d, sc))
        out_t = {}
        for key, (_, l, r, d, sc) in calls.items():
            B, C, H, W, D, kind = key
            l, r = l.contiguous(), r.contiguous()
            ctot = {True: 2 * C, "warped": C, False: C, "corr": 0}[kind] + sc * (C // 8)
            out = torch.empty((B, ctot, D, H, W), device=l.device, dtype=torch.float32)
            ws = torch.empty(max(int(L.ts_block_cost_workspace_bytes(B, C, H, W, D, sc)), 256), device=l.device, dtype=torch.uint8)
            if kind is False:
                launch = lambda: L.ts_block_cost_int_fwd(l.data_ptr(), r.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, H, W, D, sc, st)
            elif kind is True:
                dd = d.contiguous()
                launch = lambda: L.ts_block_cost_sampled_fwd(l.data_ptr(), r.data_ptr(), dd.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, H, W, D, sc, st)
            elif kind == "corr":
                dd = d.contiguous()
                launch = lambda: L.ts_block_cost_sampled_corr_fwd(l.data_ptr(), r.data_ptr(), dd.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, H, W, D, sc, st)
            else:
                dd = d.contiguous()
                launch = lambda: L.ts_block_cost_sampled_warped_fwd(l.data_ptr(), r.data_ptr(), dd.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, H, W, D, sc, st)
            for _ in range(20):
                _lib.check(launch(), "K1")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                launch()
            e1.record()
            torch.cuda.synchronize()
            out_t[key] = e0.elapsed_time(e1) / iters * 1e-3
            del out, ws
        return out_t


def cpu_baseline(sd, inputs_cpu, budget_s=20.0, all_cores=False):
    """The oracle aggregation (torch CPU ops) with the bench's own weights on the bench's own (buffer set 0) inputs."""
    from oracle import aggregation as oagg
    # more threads than ~16 only adds oversubscription on these small tensors (256-thread runs of this
    # workload measured 76 s/pass on the GPU box's host); the count used is reported as `cores`
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    lf, rf, il, ir = inputs_cpu
    cfg = dict(coarse=dict(num_sample=DIMS['coarse']['num_sample']))
    with torch.no_grad():
        t0 = time.perf_counter()
        out = oagg.aggregate(sd, lf, rf, il, ir, {}, cfg=cfg)        # warm-up (also the parity reference)
        first = time.perf_counter() - t0
        n, t_acc = 0, 0.0
        while t_acc < budget_s and n < 10:
            t0 = time.perf_counter()
            oagg.aggregate(sd, lf, rf, il, ir, {}, cfg=cfg)
            t_acc += time.perf_counter() - t0
            n += 1
    extra = {}
    if all_cores and (os.cpu_count() or 1) > cores:
        # once, for the record (SURVEY.md section 8(d) asks for all host cores): on these small tensors every thread beyond
        # ~16 only adds synchronisation, which is why the reported baseline uses 16
        torch.set_num_threads(os.cpu_count())
        with torch.no_grad():
            t0 = time.perf_counter()
            oagg.aggregate(sd, lf, rf, il, ir, {}, cfg=cfg)
            ta = time.perf_counter() - t0
        torch.set_num_threads(cores)
        extra = dict(all_host_threads=dict(cores=os.cpu_count(), value=1.0 / ta, unit="pairs/s", sample="1 pass"))
    return dict(value=n / t_acc, unit="pairs/s", cores=cores, kind="port", **extra,
                sample="%d forward passes of the config-2 aggregation (544x960, D=192, B=%d) through oracle/ "
                       "(torch %s CPU kernels, %d threads; first pass %.2fs excluded)" % (n, lf[0].shape[0], torch.__version__, cores, first)), out


def training_leg(dev, rank, world, steps, warmup, batch, seed, graph=False, sync_bn=True):
    """Data-parallel TRAINING steps of the same path (temporalstereo_amd.train.TrainStep: T=2 frame loop with the previous frame
    in eval()/no_grad, update_map, train-mode forward, fused smooth-L1 + Wasserstein losses, backward through the HIP kernels,
    bucketed gradient all-reduce over RCCL + SyncBatchNorm, clip 0.1, RMSprop), FlyingThings3D 544x960 D=192, `batch` pairs per GPU.
    Returns the `training` object of the JSON line (whole-job pairs/s over the max-over-ranks time)."""
    import torch.distributed as dist
    from temporalstereo_amd.train import TrainStep
    net = build_model(dev, seed)
    frames = []
    for t in range(2):
        lf, rf, il, ir = make_inputs(dev, seed + rank + 1000 * t, batch)
        if t == 1:
            lf, rf = [x.requires_grad_(True) for x in lf], [x.requires_grad_(True) for x in rf]
        frames.append((lf, rf, il, ir))
    calibrate_batchnorm(net, frames[0])
    gt = torch.from_numpy(synth.smooth(synth.normal(seed + rank, "gt", (batch, 1, RUN_H, RUN_W))) * 20.0 + 70.0).to(dev)
    K = torch.from_numpy(synth.sceneflow_intrinsics(batch, RUN_H, RUN_W)).to(dev)
    T = torch.from_numpy(synth.small_motion(seed + rank, batch)).to(dev)
    eye = torch.eye(4, device=dev).expand(batch, 4, 4).contiguous()
    poses = [(eye, eye), (T, eye)]
    from temporalstereo_amd import functional as TF
    step = TrainStep(net, max_disp=MAX_DISP, local_map_size=1, graph=graph, sync_bn=sync_bn)
    ex0 = TF._EXCHANGES[0]
    loss = step(frames, gt, K, poses)             # (graph: capture -- its two warm-up passes and the capture itself issue the exchanges three times)
    per_step = (TF._EXCHANGES[0] - ex0) // (3 if graph else 1)
    for _ in range(max(warmup - 1, 0)):
        loss = step(frames, gt, K, poses)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    exch = 0.0
    for _ in range(steps):
        loss = step(frames, gt, K, poses)
        exch += step.timings["exchange_ms"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    if step.peer is not None:
        step.peer.check()                         # raises if an exchange timed out waiting for a peer (the numbers would mean nothing)
    nparam = sum(p.numel() for p in step.params)
    collectives = dict(syncbn_exchanges_per_step=per_step if step.sync_bn else 0,
                       syncbn_transport=("peer mailboxes over hipIpc/xGMI: one kernel per exchange, no communicator launch (csrc/peer.hip)"
                                         if step.peer is not None else ("torch.distributed all_gather / all_reduce per layer" if step.sync_bn else "none")),
                       communicator_launches_per_step=(0 if (step.peer is not None or not step.sync_bn) else per_step) +
                       (0 if world == 1 else (1 if step.buckets is None else len(step.buckets.buckets))))
    return dict(collectives=collectives, value=world * batch * steps / el, unit="pairs/s", ms_per_step=el / steps * 1e3, steps=steps, batch_per_gpu=batch,
                frames=2, mode="hipGraph replay of previous frame + update + forward + losses + backward" if graph else "eager autograd",
                sync_bn=step.sync_bn,
                gradient_exchange_ms=exch / steps, gradient_bytes=4 * nparam, final_loss=float(loss),
                buckets_launched_in_backward=(step.buckets.launched_in_backward if step.buckets is not None else None),
                note="training step of the aggregation path (features given, requires_grad): previous frame eval/no_grad + update_map + "
                     "train-mode forward + fused losses + backward + bucketed all-reduce (RCCL) + clip 0.1 + RMSprop")


def sequence_leg(iters=10):
    """Temporal-sequence throughput at BASELINE configs[2]-[4] (their stated batches, T=2; tools/sequence_bench.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sequence_bench
    return [sequence_bench.run(i, 2, iters) for i in (2, 3, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="stereo pairs per GPU per step (config 2: 1)")
    ap.add_argument("--mode", default="native",
                    choices=["native", "native-eager", "native-graph", "module", "module-graph", "module-hip", "train", "train-graph"],
                    help="native: all-HIP inference path (aggregation.native) replayed from a recorded native "
                         "launch plan; native-eager: the same, issued op by op from Python; module: nn.Module "
                         "forward with the framework's own (MIOpen) convolutions; module-hip: nn.Module forward "
                         "with the HIP convolution Functions (unfused BatchNorm / activation); -graph: replayed "
                         "as one hipGraph")
    ap.add_argument("--condition-s", type=float, default=1.5,
                    help="upper bound (seconds) of the untimed device-conditioning phase in front of the warm-up steps: the same pass "
                         "repeated in batches of 10 until two consecutive batches agree to 1 %% (at least 0.3 s), so that the timed region "
                         "does not start on a GPU that is still ramping its clocks; reported as `conditioning`; 0 switches it off")
    ap.add_argument("--random-weights", action="store_true",
                    help="rounds 1-2 protocol: random weights with calibrated BatchNorm statistics on independent smooth-noise features "
                         "instead of the committed trained checkpoint on a planted-disparity scene (same shapes, same kernels, same speed; "
                         "an untrained pyramid amplifies rounding differences, so its parity block has a tail)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true", help="cpu_baseline: also one pass on ALL host threads (slow: oversubscribed)")
    ap.add_argument("--no-extras", action="store_true", help="skip the `training` and `sequence` objects of the default line")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch the known-size read / fill / copy streams of csrc/calib.hip (1 GiB each) once, so that a PMC "
                         "pass of this command carries its own FETCH_SIZE / WRITE_SIZE calibration (tools/k1_traffic.py)")
    ap.add_argument("--frames-in-flight", type=int, default=3, choices=(1, 2, 3),
                    help="native mode: N > 1 = the engine keeps N independent passes in flight on N sets of launch-plan "
                         "buffers, each pass a three-stage pipeline over the engine's streams; 1 = one pass at a time")
    ap.add_argument("--inflight", type=int, default=0,
                    help="extra measurement (does not change `value`): pairs/s with this many independent pairs in "
                         "flight per GPU, each a batch-1 pass on its own streams; 0/1 skips it")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under torch.distributed.run (RCCL
        # rendezvous on 127.0.0.1), same arguments; rank 0 prints the one JSON line, which is passed through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # test hooks for boxes with fewer GPUs than ranks: TS_BENCH_DEVICE pins every rank to one device and
    # TS_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU); the driver sets neither
    local = int(os.environ.get("TS_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    seed = synth.SEED0 + 2                      # config index 2
    if a.mode in ("train", "train-graph"):
        tr = training_leg(dev, rank, world, a.steps, a.warmup, a.batch, seed, graph=a.mode == "train-graph")

        def line():
            return json.dumps(dict(metric="stereo pairs/sec, TRAINING step, FlyingThings3D 540x960 D=192 T=2 (aggregation hot path)",
                                   value=tr["value"], unit="pairs/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                                   ms_per_step=tr["ms_per_step"], higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32",
                                   data="synthetic",
                                   config=dict(workload="FlyingThings3D 540x960 (run 544x960) D=192 temporal T=2 training step, batch %d/GPU" % a.batch,
                                               run_hw=[RUN_H, RUN_W], max_disp=MAX_DISP, batch_per_gpu=a.batch,
                                               parallelism="dp%d" % world, exec_mode=a.mode),
                                   training=tr))
        if world > 1 and os.environ.get("TS_BENCH_PEER", "1") != "0":
            if rank == 0:
                print(line(), flush=True)      # the collectives leg is on record whatever the legs below do (a reader takes the LAST line)
            # the same step with SyncBatchNorm's exchanges as kernels over the peer mailboxes (eager, then replayed from a hipGraph:
            # legal for world > 1 only in this form).  After the collectives leg, and guarded: it has only ever run with the ranks on
            # ONE device (tests/test_ddp_gpu.py) -- across devices a peer that does not answer times out (no hang) and the leg reports it.
            for key, g in (("peer", False), ("peer_hipgraph", True)):
                try:
                    from temporalstereo_amd.train import graph_replay_safe
                    if g and not graph_replay_safe():
                        tr[key] = dict(skipped="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in the environment when the HIP runtime started")
                        continue
                    # (few steps: should a replayed exchange cost a scheduler quantum, as it does with two ranks on ONE device, the leg
                    # still ends well inside the parent's timeout)
                    r = training_leg(dev, rank, world, min(a.steps, 6), min(a.warmup, 2), a.batch, seed, graph=g, sync_bn="peer")
                    tr[key] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "final_loss", "collectives")}
                except Exception as e:
                    tr[key] = dict(error="%s: %s" % (type(e).__name__, e))
                    break
        if rank == 0:
            print(line(), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    net = build_model(dev, seed)
    planted = os.path.exists(CKPT) and not a.random_weights
    gt0 = None
    if planted:
        load_trained(net).eval()
        inputs, gt0 = make_planted_inputs(dev, seed + rank, a.batch)
    else:
        inputs = make_inputs(dev, seed + rank, a.batch)
        calibrate_batchnorm(net, inputs)
    more_inputs = (lambda sd_: make_planted_inputs(dev, sd_, a.batch)[0]) if planted else (lambda sd_: make_inputs(dev, sd_, a.batch))

    from temporalstereo_amd.aggregation.engine import InferenceEngine
    mode = a.mode
    replay = {"native": "plan", "native-eager": "eager", "native-graph": "graph", "module": "eager", "module-graph": "graph",
              "module-hip": "eager"}[mode]
    if mode in ("module", "module-graph"):
        from temporalstereo_amd import layers
        layers.set_conv_backend("torch")
    # inputs='bind': the features stay where the (out-of-scope) backbone would write them, resident in HBM
    depth = a.frames_in_flight if mode == "native" else 1
    runner = InferenceEngine(net, backend=mode.split("-")[0], replay=replay, inputs="bind", pipeline=depth)

    # N-buffered producer: with N passes in flight the (out-of-scope) backbone writes frame k's features into buffer set k mod N
    # while the passes on the other sets are still running -- every pipeline slot is bound to its OWN input tensors (set 0 =
    # `inputs`, the one the parity check and the K1 probe look at; the others hold different frames)
    input_sets = [inputs] + [more_inputs(seed + rank + 7919 * i) for i in range(1, depth)]
    calls = [0]

    def step():
        ins = input_sets[calls[0] % depth]
        calls[0] += 1
        with torch.no_grad():
            if runner is not None:
                return runner(*ins, {})
            return net(*ins, {})

    with K1Probe() as k1:
        for _ in range(depth):  # set-up, not benchmark steps: record the launch plan of every buffer set / capture the graph
            out = step()
        torch.cuda.synchronize()
        # Device conditioning (protocol): a fresh process reaches its first timed step ~10 passes (~10 ms of GPU
        # work) after start-up, on a device that is still leaving its idle power state -- the round-2 driver line was 27 % below
        # the steady rate for that reason alone.  Bounded, untimed, reported; the warm-up and the timed steps follow unchanged.
        conditioning = None
        if a.condition_s > 0:
            tc0 = time.perf_counter()
            trace, n_cond = [], 0
            while True:
                tb = time.perf_counter()
                for _ in range(10):
                    out = step()
                torch.cuda.synchronize()
                trace.append((time.perf_counter() - tb) * 100.0)          # ms per pass of this batch
                n_cond += 10
                el = time.perf_counter() - tc0
                settled = len(trace) >= 2 and abs(trace[-1] - trace[-2]) <= 0.01 * trace[-1]
                if el >= a.condition_s or (el >= 0.3 and settled):
                    break
            conditioning = dict(passes=n_cond, seconds=time.perf_counter() - tc0, bound_s=a.condition_s,
                                first_batch_ms_per_pass=trace[0], last_batch_ms_per_pass=trace[-1],
                                note="untimed; batches of 10 passes until two consecutive batches agree to 1 % (>= 0.3 s) or the bound")
        for _ in range(a.warmup):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        while calls[0] % depth != 0:     # (not timed) `out` below = the pass on buffer set 0, the frame the oracle is run on
            step()
        out = step()
        torch.cuda.synchronize()
        k1_launched = list(k1.calls)                 # what the pipeline itself launched (before the extra measurements below add keys)
        k1_times = k1.measure(max(a.steps, 200)) if rank == 0 else {}
        k1_b4 = None
        if rank == 0 and mode.startswith("native"):
            # the same launch on four pairs: 930 MB per launch, beyond the 256 MiB Infinity Cache (SURVEY.md section 8(d) hygiene)
            from temporalstereo_amd import functional as TF
            key1 = (a.batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, "warped")
            if key1 not in k1.calls:
                key1 = key1[:5] + ("corr",)
            if key1 in k1.calls:
                _, l1, r1, d1, sc1 = k1.calls[key1]
                rep = max(1, 4 // a.batch)
                l4, r4, d4 = (x.repeat(rep, 1, 1, 1).contiguous() for x in (l1, r1, d1))
                with torch.no_grad():
                    for _ in range(3):
                        TF.block_cost(l4, r4, d4, sc1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        TF.block_cost(l4, r4, d4, sc1)
                    e1.record()
                    torch.cuda.synchronize()
                t4 = e0.elapsed_time(e1) / 20 * 1e-3
                nb4 = k1_algorithmic_bytes(l4.shape[0], l4.shape[1], l4.shape[2], l4.shape[3], 5, True)
                k1_b4 = dict(batch=int(l4.shape[0]), algorithmic_bytes=nb4, mean_us=t4 * 1e6, achieved=nb4 / t4 / 1e9, frac=nb4 / t4 / HBM_PEAK,
                             note="same launch on 4 pairs (930 MB per launch: beyond the 256 MiB Infinity Cache); this box's plain fill of that "
                                  "size runs at 4.6-5.4 TB/s, i.e. 0.58-0.68 of the spec figure is the write ceiling")
                del l4, r4, d4
        # cost volume + first (1,3,3) layer of the 1/4 level on the pipeline's own tensors, two ways --
        # materialised (rounds 1-3: volume without its reference half, convolved) and contracted (correlation blocks + the warped half
        # contracted over channels before the warp: the volume's 2C main channels are never written).  `fused` = the UNFUSED op's
        # algorithmic bytes over the time of what replaces it.
        fused = None
        if rank == 0 and mode.startswith("native"):
            ckey = (a.batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, "corr")
            if ckey not in k1.calls:
                ckey = ckey[:5] + ("warped",)
            agg_n = getattr(runner, "net", None)
            if ckey in k1.calls and hasattr(agg_n, "precise"):
                from temporalstereo_amd import functional as TF
                from temporalstereo_amd.aggregation import native as _N
                _, lq, rq, dq, scq = k1.calls[ckey]
                pr = agg_n.precise
                lq, rq, dq = lq.contiguous(), rq.contiguous(), dq.contiguous()

                def timed_us(fn, n=100):
                    with torch.no_grad():
                        for _ in range(10):
                            fn()
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(n):
                            fn()
                        e1.record()
                        torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / n * 1e3
                with torch.no_grad():
                    lt, rt = pr.left_term(lq), pr.right_term(rq)
                t_mat = timed_us(lambda: _N.conv_hw(TF.block_cost_warped(lq, rq, dq, scq), pr.init0.f0, 1, pr.init0.dil, addend=lt))
                t_con = timed_us(lambda: _N.conv_hw_warp(TF.block_cost_corr(lq, rq, dq, scq), pr.init0_corr, rt, dq, lt.squeeze(2), pr.init0.dil))
                t_q = timed_us(lambda: pr.right_term(rq))
                nbf = k1_algorithmic_bytes(*ckey[:5], True)
                fused = dict(kernels="ts_block_cost_sampled_corr_fwd + ts_conv3d_hw_warp_fwd (gather + (1,3,3) convolution over the correlation "
                                     "blocks) + the 1x1 pre-contraction right -> Q (ts_conv3d_d_fwd, k = 1; a function of the features only, "
                                     "issued at the start of a pass)",
                             replaces="ts_block_cost_sampled_warped_fwd + ts_conv3d_hw_fwd over [warped | corr] (rounds 1-3)",
                             mean_us=t_con + t_q, on_the_level_chain_us=t_con, precontraction_us=t_q, replaced_mean_us=t_mat,
                             unfused_algorithmic_bytes=nbf, equivalent=nbf / ((t_con + t_q) * 1e-6) / 1e9, unit="GB/s",
                             frac=nbf / ((t_con + t_q) * 1e-6) / HBM_PEAK, in_use=bool(_N.FUSED_K1),
                             note="algorithmic bytes of the UNFUSED cost-volume op (SURVEY 8(d)) over the time of cost volume + first layer "
                                  "in the contracted form; both forms include the first layer's convolution, so compare mean_us with "
                                  "replaced_mean_us, and `equivalent` with the unfused op's `achieved` only as a bytes-never-moved figure")
        if a.calibrate and rank == 0:
            from temporalstereo_amd import _lib
            nbytes = 1 << 30
            ca, cb = torch.empty(nbytes // 4, device=dev), torch.ones(nbytes // 4, device=dev)
            stc = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for kind in (0, 1, 2):
                _lib.check(_lib.lib().ts_calib_stream(kind, _lib.ptr(ca), _lib.ptr(cb), nbytes, stc), "ts_calib_stream")
            torch.cuda.synchronize()
            del ca, cb

    # one pass at a time (what a latency-bound caller sees), next to the several-in-flight headline
    one_at_a_time = None
    if depth > 1 and rank == 0:
        single = InferenceEngine(net, backend="native", replay="plan", inputs="bind")
        with torch.no_grad():
            for _ in range(3):
                single(*inputs, {})
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                single(*inputs, {})
            torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / a.steps
        one_at_a_time = dict(value=a.batch / dt1, unit="pairs/s per GPU", ms_per_step=dt1 * 1e3,
                             note="same engine with frames_in_flight=1 on rank 0: every pass waits for the previous one")

    # The same pipelined engine with every convolution on the f32-input MFMA kernel (TS_CONV_X6=0 semantics): the headline uses
    # ts_conv3d_hw_x6_fwd where a layer allows it -- fp32 products assembled from six bf16 MFMA products, closer to the exact sums than the
    # f32 MFMA chain (tests/test_conv_x6_gpu.py) -- and this is the number without it, measured in this run.
    f32_only = None
    if mode == "native" and rank == 0 and not a.no_extras:
        from temporalstereo_amd.aggregation import native as _N
        if _N.X6:
            _N.X6 = False
            try:
                eng32 = InferenceEngine(net, backend="native", replay="plan", inputs="bind", pipeline=depth)
                with torch.no_grad():
                    for _ in range(depth + 5):
                        eng32(*inputs, {})
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(a.steps):
                        eng32(*inputs, {})
                    torch.cuda.synchronize()
                dt32 = (time.perf_counter() - t1) / a.steps
                f32_only = dict(value=a.batch / dt32, unit="pairs/s per GPU", ms_per_step=dt32 * 1e3)
                del eng32
            finally:
                _N.X6 = True

    # Serving-style concurrency: N independent batch-1 passes in flight on one GPU (each its own plan, buffers
    # and streams).  The chain of small launches of one pass leaves most CUs idle; another pass fills them.
    concurrent = None
    if a.inflight > 1 and mode == "native":
        lanes = []
        for i in range(a.inflight):
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
            for k in range(a.steps):
                st, eng, ins = lanes[k % a.inflight]
                with torch.cuda.stream(st):
                    eng(*ins, {})
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        conc_elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([conc_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            conc_elapsed = float(tt.item())
        concurrent = dict(inflight=a.inflight, value=world * a.batch * a.steps / conc_elapsed, unit="pairs/s",
                          ms_per_step=conc_elapsed / a.steps * 1e3,
                          note="%d independent batch-1 passes in flight per GPU (own launch plan, buffers and streams "
                               "each); not the headline value" % a.inflight)

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    sequence = None
    if world == 1 and mode == "native" and not a.no_extras:
        try:        # before the training legs: their graph capture leaves a private memory pool and extra streams behind
            sequence = sequence_leg(20)
        except Exception as e:
            sequence = dict(error="%s: %s" % (type(e).__name__, e))

    training = None
    if mode == "native" and not a.no_extras and world == 1:
        # (one GPU only: with several ranks the training step's collectives -- SyncBatchNorm, bucketed all-reduce -- would put
        # the headline line at the mercy of a rank that fails inside them; `--mode train --gpus N` is the multi-GPU training run)
        try:
            training = training_leg(dev, rank, world, 12, 6, 1, seed)
            g = training_leg(dev, rank, world, 16, 3, 1, seed, graph=True)
            training["hipgraph"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "final_loss")}
        except Exception as e:      # the headline number must survive a failure of the extra leg
            training = dict(error="%s: %s" % (type(e).__name__, e)) if training is None else dict(training, hipgraph_error="%s: %s" % (type(e).__name__, e))

    result = None
    if rank == 0:
        pairs = world * a.batch * a.steps
        # dominant cost-volume launch: the 1/4-resolution (precise) sampled build
        # The judged figure is the complete op at its unfused boundary; the native
        # pipeline itself runs the variant without the repeated left half, reported next to it.
        pkey = (a.batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, True)
        wkey = pkey[:5] + ("warped",)
        roofline = None
        if pkey in k1_times:
            nbytes = k1_algorithmic_bytes(*pkey)
            ach = nbytes / k1_times[pkey]
            # all three levels, each as the COMPLETE op at its unfused boundary (329.3 MB per pair at batch 1)
            used = [k for k in k1_times if k[5] is True or k[5] is False]
            all_b = sum(k1_algorithmic_bytes(*k) for k in used)
            all_t = sum(k1_times[k] for k in used)
            # ... and what the pipeline itself launches at each level (round 4: the correlation blocks alone at the sampled levels)
            launched = [k for k in k1_launched if k in k1_times]
            lau_b = sum(k1_algorithmic_bytes(*k) for k in launched)
            lau_t = sum(k1_times[k] for k in launched)
            traffic, traffic_file = None, None
            for cand in ("r04_k1_hbm_traffic_pmc.json", "r03_k1_hbm_traffic_pmc.json", "r02_k1_hbm_traffic_pmc.json"):
                try:    # HBM bytes per launch from the PMC passes committed under profiles/ (cannot be collected in-process)
                    with open(os.path.join(ROOT, "profiles", cand)) as fh:
                        pm = json.load(fh)
                    if list(pm["workload_key"]) == list(pkey):
                        traffic, traffic_file = pm["hbm_bytes_per_launch"], cand
                        break
                except (OSError, ValueError, KeyError):
                    pass
            ckey_p = pkey[:5] + ("corr",)
            vkey = ckey_p if ckey_p in k1.calls else wkey
            roofline = dict(bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK,
                            traffic=traffic,
                            measured="HIP events on the launch stream around %d back-to-back C-ABI launches on the pipeline's "
                                     "own input tensors, right after the timed steps" % max(a.steps, 200),
                            kernel="ts_block_cost_sampled_fwd (block_cost_fast + block_cost_upsample_rows) on "
                                   "[%d,%d,%d,%d] x %d candidates" % pkey[:5],
                            algorithmic_bytes=nbytes, mean_us=k1_times[pkey] * 1e6,
                            frac_of_measured_copy_ceiling=ach / 6.29e12,
                            pipeline_variant=(dict(kernel=("ts_block_cost_sampled_corr_fwd (the correlation blocks alone: the first layer takes the "
                                                           "warped half in pre-contracted form, see `fused`)" if vkey[5] == "corr" else
                                                           "ts_block_cost_sampled_warped_fwd (volume without the D-fold repeat of "
                                                           "the left features)") + "; what the native pipeline launches",
                                                   algorithmic_bytes=k1_algorithmic_bytes(*vkey), mean_us=k1_times[vkey] * 1e6,
                                                   achieved=k1_algorithmic_bytes(*vkey) / k1_times[vkey] / 1e9,
                                                   frac=k1_algorithmic_bytes(*vkey) / k1_times[vkey] / HBM_PEAK)
                                              if vkey in k1_times else None),
                            warped_variant=(dict(kernel="ts_block_cost_sampled_warped_fwd (rounds 1-3: volume without the D-fold repeat of the left features)",
                                                 algorithmic_bytes=k1_algorithmic_bytes(*wkey), mean_us=k1_times[wkey] * 1e6,
                                                 frac=k1_algorithmic_bytes(*wkey) / k1_times[wkey] / HBM_PEAK) if wkey in k1_times else None),
                            all_levels=dict(algorithmic_bytes=all_b, mean_us=all_t * 1e6,
                                            achieved=all_b / all_t / 1e9, frac=all_b / all_t / HBM_PEAK,
                                            note="coarse (int) + fine + precise (sampled), each the complete op at its unfused boundary"),
                            all_levels_as_launched=dict(algorithmic_bytes=lau_b, mean_us=lau_t * 1e6, achieved=lau_b / lau_t / 1e9,
                                                        frac=lau_b / lau_t / HBM_PEAK,
                                                        note="the variants the pipeline launches, with THEIR algorithmic bytes (rounds 1-3 reported this as all_levels)"),
                            fused=fused,
                            beyond_infinity_cache=k1_b4,
                            traffic_source="profiles/%s: FETCH_SIZE / WRITE_SIZE passes of this command under rocprofv3 (tools/k1_traffic.py, "
                                           "tools/prof_r04.sh); a PMC pass cannot run inside the timed process" % traffic_file)
        result = dict(metric="stereo pairs/sec, FlyingThings3D 540x960 D=192 (aggregation hot path)",
                      value=pairs / elapsed, unit="pairs/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                      ms_per_step=elapsed / a.steps * 1e3, higher_is_better=True, scaling="weak",
                      vs_baseline=None, dtype="f32", data="synthetic",
                      config=dict(workload="BASELINE configs[1]: FlyingThings3D 540x960 (run 544x960) D=192 "
                                           "single-frame aggregation, batch %d/GPU, eval" % a.batch,
                                  run_hw=[RUN_H, RUN_W], max_disp=MAX_DISP, batch_per_gpu=a.batch,
                                  parallelism="replicas x%d" % world, exec_mode=mode, frames_in_flight=depth, input_buffer_sets=depth,
                                  weights_and_inputs=("trained checkpoint tests/golden/ckpt_planted.npz on planted-disparity scenes (tests/synth.stereo_sequence)"
                                                      if planted else "random weights, calibrated BatchNorm, smooth-noise features (--random-weights)"),
                                  conv_arithmetic="fp32 everywhere; stride-1 (1,3,3) layers with Cin >= 16, Cout > 8 -- and, on grids of 256+ workgroups, the stride-2 (1,3,3) layers and the 4x4 deconvolutions (x6s) -- form each fp32 product from "
                                                  "six bf16 MFMA products with fp32 accumulation (x6: dropped terms <= 2^-24 of a product, chunks summed apart; measured max "
                                                  "error vs fp64 0.15e-6-0.26e-6 of the output magnitude, the f32-input MFMA kernel 0.3e-6-0.7e-6); "
                                                  "f32_mfma_only = this engine with that switched off"),
                      roofline=roofline)
        if conditioning is not None:
            result["conditioning"] = conditioning
        if one_at_a_time is not None:
            result["one_pass_at_a_time"] = one_at_a_time
        if f32_only is not None:
            result["f32_mfma_only"] = f32_only
        if concurrent is not None:
            result["concurrent_pairs"] = concurrent
        if training is not None:
            result["training"] = training
        if sequence is not None:
            result["sequence"] = sequence
        if world == 1 and not a.no_cpu_baseline:
            # the oracle with the very same weights (incl. BatchNorm statistics) on the very same inputs (buffer set 0): timed as the
            # CPU baseline, and its first pass is the parity reference for `out`
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            cpu_in = tuple([x.cpu() for x in t] if isinstance(t, list) else t.cpu() for t in inputs)
            base, ref = cpu_baseline(sd, cpu_in, all_cores=a.cpu_all_cores)
            full, rfull = out[0][0].detach().cpu().double(), ref[0][0].double()
            if gt0 is not None:
                # planted scene: a real ground truth; EPE = mean |d - gt| over 0 < gt < max_disp (data/evaluation/pixel_error.py:33-63)
                gt = gt0.double()
            else:
                # no ground truth exists for noise inputs: gt* = reference output + N(0,1) clipped to (0, MAX_DISP)
                gen = torch.Generator().manual_seed(seed)
                gt = (rfull + torch.randn(rfull.shape, generator=gen, dtype=torch.float64)).clamp(0, MAX_DISP)
            valid = (gt > 0) & (gt < MAX_DISP)
            e_ours, e_ref = float((full - gt).abs()[valid].mean()), float((rfull - gt).abs()[valid].mean())
            result["cpu_baseline"] = base
            result["parity"] = dict(delta_epe_px=abs(e_ours - e_ref), epe_px=e_ours, epe_reference_px=e_ref,
                                    mean_abs_diff_px=float((full - rfull).abs().mean()),
                                    max_abs_diff_px=float((full - rfull).abs().max()),
                                    frac_pixels_off_by_0p01=float(((full - rfull).abs() > 0.01).double().mean()),
                                    tolerance_px=1e-3, ground_truth="planted disparity of the synthetic scene" if gt0 is not None else "reference output + N(0,1)",
                                    reference="oracle/ (CPU port pinned to the reference's golden vectors, incl. full-size runs of the reference with this checkpoint)")
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and mode == "native" and not a.no_extras:
            # The data-parallel TRAINING step on the same N GPUs (SyncBatchNorm + bucketed all-reduce over RCCL), as its own job with a
            # timeout: a rank that hangs inside a collective must not take the headline line with it.  The inference ranks are done
            # (process group destroyed above); rank 0 launches `bench.py --mode train --gpus N` and embeds its `training` object.
            import subprocess
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
                                "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE") and not k.startswith("TORCHELASTIC_")}
            try:
                import signal
                child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--mode", "train", "--gpus", str(world), "--steps", "10",
                                          "--warmup", "4", "--batch", str(a.batch)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                         start_new_session=True)                    # its own process group: the launcher AND its ranks
                try:
                    so, se = child.communicate(timeout=float(os.environ.get("TS_BENCH_TRAIN_TIMEOUT", "300")))
                    lines = [ln for ln in so.splitlines() if ln.startswith("{")]
                    result["training"] = json.loads(lines[-1])["training"] if (child.returncode == 0 and lines) else \
                        dict(error="exit code %d: %s" % (child.returncode, (se or so)[-400:]))
                except subprocess.TimeoutExpired:
                    os.killpg(child.pid, signal.SIGKILL)                              # exactly the group started above
                    so, _ = child.communicate()
                    lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
                    if lines:       # the collectives leg had finished (its line goes out before the peer legs start)
                        result["training"] = dict(json.loads(lines[-1])["training"], later_legs="timed out")
                    else:
                        result["training"] = dict(error="timed out (a rank stuck in a collective?)")
            except Exception as e:          # the headline must survive anything the extra leg does
                result["training"] = dict(error="%s: %s" % (type(e).__name__, e))
        print(json.dumps(result), flush=True)
    return result


if __name__ == "__main__":
    main()
