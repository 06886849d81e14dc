#!/usr/bin/env python
"""Headline benchmark: stereo pairs/s of the cost-volume aggregation hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode native|native-eager|...|train|train-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (coarse -> fine -> precise aggregation: cost-volume build, 3-D aggregation
pyramid, top-k soft-argmax regression, upsamplers) over one batch of synthetic FlyingThings3D-shaped inputs already
resident in HBM: BASELINE.json configs[1] = 540x960 run at 544x960 (as the reference does, sceneflow.yaml:84-85),
D=192 (COARSE.NUM_SAMPLE=12), single frame, batch 1 per GPU, fp32, eval mode.  Ranks are independent replicas (stereo
pairs shard with no data-path collective): value = pairs all ranks processed / max-over-ranks time ("weak" scaling).

This file: arguments, rank set-up, THE TIMED REGION (`timed_steps`), the CPU baseline / parity leg (the only code
outside tests/ that runs oracle/) and the one JSON line.  Every other measurement lives in benchlegs/ and runs AFTER
the timed region:
  roofline            benchlegs/k1.py        cost-volume build at the 1/4 level: algorithmic bytes over
                                             its mean duration by HIP events, against the 8.0 TB/s HBM peak AND against
                                             this board's fill ceiling measured in the same run
  one_pass_at_a_time, f32_mfma_only,
  concurrent_pairs, sequence   benchlegs/extras.py
  training            benchlegs/training.py  the data-parallel training step (also `--mode train | train-graph`)
  cpu_baseline, parity                       below
"""
import argparse
import json
import os
import sys
import time

# the one framework pass this script makes (BatchNorm calibration) should not trigger MIOpen's exhaustive solver search
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# hipGraph replays of the training step (train-graph leg): explicit opt-in, before anything touches the GPU
# (temporalstereo_amd/train.py)
from temporalstereo_amd import train as _ts_train  # noqa: E402
try:
    _ts_train.enable_graph_replay()
# imported as a module by a process that already used the GPU: only the train-graph leg needs it
except RuntimeError:
    pass

from benchlegs import extras, k1 as k1legs, training as trainlegs  # noqa: E402
# noqa: E402,F401
from benchlegs.common import (CKPT, DIMS, HBM_PEAK, MAX_DISP, RUN_H, RUN_W, build_model, calibrate_batchnorm,
                              k1_algorithmic_bytes, load_trained, make_inputs, make_planted_inputs, synth)
from benchlegs.training import training_leg  # noqa: E402,F401  (tests and tools reach these through `bench`)
from benchlegs.extras import sequence_leg  # noqa: E402,F401


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1, help="stereo pairs per GPU per step (config 2: 1)")
    ap.add_argument("--mode", default="native",
                    choices=["native", "native-eager", "native-graph", "module", "module-graph", "module-hip", "train",
                             "train-graph"],
                    help="native: all-HIP inference path (aggregation.native) replayed from a recorded native launch "
                         "plan; native-eager: the same, issued op by op from Python; module: nn.Module forward with "
                         "the framework's own (MIOpen) convolutions; module-hip: nn.Module forward with the HIP "
                         "convolution Functions (unfused BatchNorm / activation); -graph: replayed as one hipGraph; "
                         "train / train-graph: the training step (benchlegs/training.py)")
    ap.add_argument("--condition-s", type=float, default=1.5,
                    help="upper bound (seconds) of the untimed device-conditioning phase in front of the warm-up "
                         "steps: the same pass repeated in batches of 10 until two consecutive batches agree to 1 %% "
                         "(at least 0.3 s), so that the timed region does not start on a GPU that is still ramping its "
                         "clocks; reported as `conditioning`; 0 switches it off")
    ap.add_argument("--random-weights", action="store_true",
                    help="rounds 1-2 protocol: random weights with calibrated BatchNorm statistics on independent "
                         "smooth-noise features instead of the committed trained checkpoint on a planted-disparity "
                         "scene (same shapes, kernels and speed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-all-cores", action="store_true",
                    help="cpu_baseline: also one pass on ALL host threads (slow: oversubscribed)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `training` and `sequence` objects of the default line")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch the known-size read / fill / copy streams of csrc/calib.hip (1 GiB each) once, "
                         "so that a PMC pass of this command carries its own FETCH_SIZE / WRITE_SIZE calibration "
                         "(tools/k1_traffic.py)")
    ap.add_argument("--frames-in-flight", type=int, default=3, choices=(1, 2, 3, 4),
                    help="native mode: N > 1 = the engine keeps N independent passes in flight on N sets of "
                         "launch-plan buffers, each pass a pipeline of stages over the engine's streams (six at batch "
                         "1); 1 = one "
                         "pass at a time")
    ap.add_argument("--inflight", type=int, default=0,
                    help="extra measurement (does not change `value`): pairs/s with this many independent pairs in "
                         "flight per GPU, each a batch-1 pass on its own streams; 0/1 skips it")
    return ap.parse_args()


def init_ranks(a):
    """-> (rank, world, device, dist | None).  Plain `python bench.py --gpus N` becomes the launcher: one rank per GPU
    under torch.distributed.run (RCCL rendezvous on 127.0.0.1), same arguments; rank 0's JSON line is passed through."""
    env = os.environ
    world, rank, local = int(env.get("WORLD_SIZE", "1")), int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "RANK" not in os.environ:
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
    return rank, world, dev, dist


def condition_device(step, bound_s):
    """Untimed, bounded, reported: a fresh process reaches its first timed step ~10 passes after
    start-up, on a device that is still leaving its idle power state.  Batches of 10 passes until two consecutive
    batches agree to 1 % (>= 0.3 s)."""
    tc0 = time.perf_counter()
    trace, n = [], 0
    while True:
        tb = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        trace.append((time.perf_counter() - tb) * 100.0)          # ms per pass of this batch
        n += 10
        el = time.perf_counter() - tc0
        settled = len(trace) >= 2 and abs(trace[-1] - trace[-2]) <= 0.01 * trace[-1]
        if el >= bound_s or (el >= 0.3 and settled):
            break
    return dict(passes=n, seconds=time.perf_counter() - tc0, bound_s=bound_s, first_batch_ms_per_pass=trace[0],
                last_batch_ms_per_pass=trace[-1],
                note="untimed; batches of 10 passes until two consecutive batches agree to 1 % (>= 0.3 s) or the bound")


def timed_steps(step, warmup, steps, dist, dev):
    """THE TIMED REGION: W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides.
    -> seconds (MAX over ranks)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def cpu_baseline(sd, inputs_cpu, budget_s=20.0, all_cores=False):
    """The oracle aggregation (torch CPU ops, oracle/: test infrastructure, here as the reported CPU baseline and the
    parity reference) with the bench's own weights on the bench's own (buffer set 0) inputs.  -> (cpu_baseline object,
    oracle outputs)."""
    from oracle import aggregation as oagg
    # more threads than ~16 only adds oversubscription on these small tensors (256-thread runs of this workload
    # measured 76 s/pass on the GPU box's host); the count used is reported as `cores`
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
        # once, for the record: every thread beyond ~16 only adds synchronisation here
        torch.set_num_threads(os.cpu_count())
        with torch.no_grad():
            t0 = time.perf_counter()
            oagg.aggregate(sd, lf, rf, il, ir, {}, cfg=cfg)
            ta = time.perf_counter() - t0
        torch.set_num_threads(cores)
        extra = dict(all_host_threads=dict(cores=os.cpu_count(), value=1.0 / ta, unit="pairs/s", sample="1 pass"))
    sample = ("%d forward passes of the config-2 aggregation (544x960, D=192, B=%d) through oracle/ (torch %s CPU "
              "kernels, %d threads; first pass %.2fs excluded)" % (n, lf[0].shape[0], torch.__version__, cores, first))
    return dict(value=n / t_acc, unit="pairs/s", cores=cores, kind="port", sample=sample, **extra), out


def parity_block(out, ref, gt0, seed):
    """EPE of the product path and of the oracle against the ground truth (data/evaluation/pixel_error.py:33-63),
    |dEPE| < 1e-3 px."""
    full, rfull = out[0][0].detach().cpu().double(), ref[0][0].double()
    if gt0 is not None:
        gt, gt_name = gt0.double(), "planted disparity of the synthetic scene"
    else:
        # noise inputs have no ground truth: gt* = reference output + N(0,1) clipped to (0, MAX_DISP)
        gen = torch.Generator().manual_seed(seed)
        gt = (rfull + torch.randn(rfull.shape, generator=gen, dtype=torch.float64)).clamp(0, MAX_DISP)
        gt_name = "reference output + N(0,1)"
    valid = (gt > 0) & (gt < MAX_DISP)
    e_ours, e_ref = float((full - gt).abs()[valid].mean()), float((rfull - gt).abs()[valid].mean())
    d = (full - rfull).abs()
    return dict(delta_epe_px=abs(e_ours - e_ref), epe_px=e_ours, epe_reference_px=e_ref,
                mean_abs_diff_px=float(d.mean()), max_abs_diff_px=float(d.max()),
                frac_pixels_off_by_0p01=float((d > 0.01).double().mean()), tolerance_px=1e-3, ground_truth=gt_name,
                reference="oracle/ (CPU port pinned to the reference's golden vectors, incl. full-size runs of the "
                          "reference with this checkpoint)")


_CONV_ARITHMETIC = (
    "fp32 everywhere; stride-1 (1,3,3) layers with Cin >= 16, Cout > 8 -- and, on grids of 256+ workgroups, the "
    "stride-2 (1,3,3) layers and the 4x4 deconvolutions (x6s) -- form each fp32 product from six bf16 MFMA products "
    "with fp32 accumulation (x6: dropped terms <= 2^-24 of a product, chunks summed apart; measured max error vs fp64 "
    "0.15e-6-0.26e-6 of the output magnitude, the f32-input MFMA kernel 0.3e-6-0.7e-6); f32_mfma_only = this engine "
    "with that switched off")


def main():
    a = parse_args()
    rank, world, dev, dist = init_ranks(a)
    seed = synth.SEED0 + 2                      # config index 2
    if a.mode in ("train", "train-graph"):
        return trainlegs.train_mode_main(a, dev, rank, world, dist, seed)

    # ---- workload: weights, inputs, engine
    # ---------------------------------------------------------------------------------------
    net = build_model(dev, seed)
    planted = os.path.exists(CKPT) and not a.random_weights
    gt0 = None
    if planted:
        load_trained(net).eval()
        inputs, gt0 = make_planted_inputs(dev, seed + rank, a.batch)
    else:
        inputs = make_inputs(dev, seed + rank, a.batch)
        calibrate_batchnorm(net, inputs)
    if planted:
        more_inputs = lambda s: make_planted_inputs(dev, s, a.batch)[0]          # noqa: E731
    else:
        more_inputs = lambda s: make_inputs(dev, s, a.batch)                     # noqa: E731

    from temporalstereo_amd.aggregation.engine import InferenceEngine
    mode = a.mode
    replay = {"native": "plan", "native-eager": "eager", "native-graph": "graph", "module": "eager",
              "module-graph": "graph",
              "module-hip": "eager"}[mode]
    if mode in ("module", "module-graph"):
        from temporalstereo_amd import layers
        layers.set_conv_backend("torch")
    depth = a.frames_in_flight if mode == "native" else 1
    # inputs='bind': the features stay where the (out-of-scope) backbone would write them, resident in HBM
    runner = InferenceEngine(net, backend=mode.split("-")[0], replay=replay, inputs="bind", pipeline=depth)
    # N-buffered producer: with N passes in flight the backbone writes frame k's features into buffer set k mod N
    # while the passes on the other sets are still running -- every pipeline slot is bound to its OWN input tensors
    # (set 0 = `inputs`: parity, K1 probe)
    input_sets = [inputs] + [more_inputs(seed + rank + 7919 * i) for i in range(1, depth)]
    calls = [0]

    def step():
        ins = input_sets[calls[0] % depth]
        calls[0] += 1
        with torch.no_grad():
            return runner(*ins, {})

    # ---- set-up, conditioning, THE TIMED REGION
    # ----------------------------------------------------------------------------------
    with k1legs.K1Probe() as k1:
        # set-up, not benchmark steps: record the launch plan of every buffer set / capture the graph
        for _ in range(depth):
            step()
        torch.cuda.synchronize()
        conditioning = condition_device(step, a.condition_s) if a.condition_s > 0 else None
        elapsed = timed_steps(step, a.warmup, a.steps, dist, dev)
        # (not timed) `out` below = the pass on buffer set 0, the frame the oracle is run on
        while calls[0] % depth != 0:
            step()
        out = step()
        torch.cuda.synchronize()
        launched = list(k1.calls)        # what the pipeline itself launched (before the measurements below add keys)

        # ---- everything below is outside the timed region
        # ------------------------------------------------------------------------
        roofline = k1legs.run(k1, launched, runner, dev, a.batch, max(a.steps, 200), mode.startswith("native"),
                              ceilings=not a.calibrate) if rank == 0 else None
        if a.calibrate and rank == 0:
            from temporalstereo_amd import _lib
            nbytes = 1 << 30
            ca, cb = torch.empty(nbytes // 4, device=dev), torch.ones(nbytes // 4, device=dev)
            stc = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for kind in (0, 1, 2):
                _lib.check(_lib.lib().ts_calib_stream(kind, _lib.ptr(ca), _lib.ptr(cb), nbytes, stc), "ts_calib_stream")
            torch.cuda.synchronize()
            del ca, cb

    native_extras = mode == "native" and not a.no_extras
    one = extras.one_pass_at_a_time(net, inputs, a.steps, a.batch) if (depth > 1 and rank == 0) else None
    f32_only = extras.f32_mfma_only(net, inputs, a.steps, a.batch, depth) if (native_extras and rank == 0) else None
    concurrent = None
    if a.inflight > 1 and mode == "native":
        concurrent = extras.concurrent_lanes(net, dev, dist, world, more_inputs, seed, rank, a.inflight, a.steps,
                                             a.batch)
    sequence = None
    if world == 1 and native_extras:
        # before the training legs: their graph capture leaves a private memory pool and extra streams behind
        try:
            sequence = extras.sequence_leg(20)
        except Exception as e:
            sequence = dict(error="%s: %s" % (type(e).__name__, e))
    # (one GPU only: with several ranks the training step's collectives would put the headline line at the mercy of a
    # rank that fails inside them; rank 0 runs `bench.py --mode train --gpus N` as its own job below)
    training = trainlegs.single_gpu_training(dev, seed) if (native_extras and world == 1) else None

    # ---- the line ------------------------------------------------------------------------------------------------
    result = None
    if rank == 0:
        weights = ("trained checkpoint tests/golden/ckpt_planted.npz on planted-disparity scenes "
                   "(tests/synth.stereo_sequence)" if planted
                   else "random weights, calibrated BatchNorm, smooth-noise features (--random-weights)")
        result = dict(metric="stereo pairs/sec, FlyingThings3D 540x960 D=192 (aggregation hot path)",
                      value=world * a.batch * a.steps / elapsed, unit="pairs/s", n_gpus=world, steps=a.steps,
                      warmup=a.warmup, ms_per_step=elapsed / a.steps * 1e3, higher_is_better=True, scaling="weak",
                      vs_baseline=None, dtype="f32", data="synthetic",
                      config=dict(workload="BASELINE configs[1]: FlyingThings3D 540x960 (run 544x960) D=192 "
                                           "single-frame aggregation, batch %d/GPU, eval" % a.batch,
                                  run_hw=[RUN_H, RUN_W], max_disp=MAX_DISP, batch_per_gpu=a.batch,
                                  parallelism="replicas x%d" % world, exec_mode=mode, frames_in_flight=depth,
                                  input_buffer_sets=depth, weights_and_inputs=weights,
                                  conv_arithmetic=_CONV_ARITHMETIC),
                      roofline=roofline)
        if mode == "native" and depth > 1:
            # disclosed with the passes in flight: the streams a pass is cut over (NativeAggregator._stages_for)
            streams = {id(v) for v in runner.net._stages_for(a.batch).values() if v is not None}
            result["config"]["pipeline_streams"] = 1 + len(streams)
        if training is not None:
            # flat copies: a record that keeps only the top level of nested objects still shows them
            result["config"]["training_ms_per_step"] = training.get("ms_per_step")
            result["config"]["training_hipgraph_ms_per_step"] = (training.get("hipgraph") or {}).get("ms_per_step")
        if one is not None:
            result["config"]["one_pass_at_a_time_pairs_per_s"] = one.get("value")
        for key, val in (("conditioning", conditioning), ("one_pass_at_a_time", one), ("f32_mfma_only", f32_only),
                         ("concurrent_pairs", concurrent), ("training", training), ("sequence", sequence)):
            if val is not None:
                result[key] = val
        if world == 1 and not a.no_cpu_baseline:
            # the oracle with the very same weights (incl. BatchNorm statistics) on the very same inputs (buffer set
            # 0): timed as the CPU baseline, and its first pass is the parity reference for `out`
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
            cpu_in = tuple([x.cpu() for x in t] if isinstance(t, list) else t.cpu() for t in inputs)
            result["cpu_baseline"], ref = cpu_baseline(sd, cpu_in, all_cores=a.cpu_all_cores)
            result["parity"] = parity_block(out, ref, gt0, seed)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and native_extras:     # the inference ranks are done (process group destroyed above)
            result["training"] = trainlegs.multi_gpu_training_child(world, a.batch)
        print(json.dumps(result), flush=True)
    return result


if __name__ == "__main__":
    main()
