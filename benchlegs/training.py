"""The data-parallel TRAINING step of the same path (temporalstereo_amd.train.TrainStep), BASELINE configs[1] geometry,
T = 2: `python bench.py --mode train | train-graph [--gpus N]`, and the `training` object of the default line.
Reference: TemporalStereo.training_step + multi_frame_forward (projects/TemporalStereo/TemporalStereo.py:130-168,
:250-280) under pl.Trainer(strategy='ddp', sync_batchnorm=True, gradient_clip_val=0.1) (dist_train.py:82-96)."""
import json
import os
import signal
import subprocess
import sys
import time

import torch

from .common import MAX_DISP, ROOT, RUN_H, RUN_W, build_model, calibrate_batchnorm, make_inputs, synth

_NOTE = ("training step of the aggregation path (features given, requires_grad): previous frame eval/no_grad + "
         "update_map + train-mode forward + fused losses + backward + bucketed all-reduce (RCCL) + clip 0.1 + RMSprop")


def training_leg(dev, rank, world, steps, warmup, batch, seed, graph=False, sync_bn=True):
    """`steps` optimisation steps, `batch` pairs per GPU (previous frame in eval()/no_grad, update_map, train-mode
    forward, fused smooth-L1 + Wasserstein losses, backward through the HIP kernels, SyncBatchNorm + bucketed gradient
    all-reduce, clip 0.1, RMSprop) between barriers; the whole-job pairs/s over the max-over-ranks time."""
    import torch.distributed as dist
    from temporalstereo_amd import functional as TF
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
    step = TrainStep(net, max_disp=MAX_DISP, local_map_size=1, graph=graph, sync_bn=sync_bn)
    try:
        ex0 = TF._EXCHANGES[0]
        # (graph: the capture -- two warm-up passes and the capture issue the exchanges 3x)
        loss = step(frames, gt, K, poses)
        per_step = (TF._EXCHANGES[0] - ex0) // (3 if graph else 1)
        if graph:       # inputs resident where the captured step reads them (as the headline's inputs='bind')
            frames, gt, K, poses = step.bound_inputs()
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
            # raises if an exchange timed out waiting for a peer (the numbers would mean nothing)
            step.peer.check()
        nparam = sum(p.numel() for p in step.params)
        peer_txt = "peer mailboxes over hipIpc/xGMI: one kernel per exchange, no communicator launch (csrc/peer.hip)"
        coll_txt = "torch.distributed all_gather / all_reduce per layer" if step.sync_bn else "none"
        launches = (0 if (step.peer is not None or not step.sync_bn) else per_step) + \
                   (0 if world == 1 else (1 if step.buckets is None else len(step.buckets.buckets)))
        collectives = dict(syncbn_exchanges_per_step=per_step if step.sync_bn else 0,
                           syncbn_transport=peer_txt if step.peer is not None else coll_txt,
                           communicator_launches_per_step=launches)
        return dict(collectives=collectives, value=world * batch * steps / el, unit="pairs/s",
                    ms_per_step=el / steps * 1e3, steps=steps, batch_per_gpu=batch, frames=2,
                    mode=("hipGraph replay of previous frame + update + forward + losses + backward" if graph
                          else "eager autograd"),
                    sync_bn=step.sync_bn, gradient_exchange_ms=exch / steps, gradient_bytes=4 * nparam,
                    final_loss=float(loss),
                    buckets_launched_in_backward=(step.buckets.launched_in_backward if step.buckets is not None
                                                  else None),
                    note=_NOTE)
    finally:
        # peer mailboxes unmapped behind a barrier; nothing stays installed for the next leg
        step.close()


def single_gpu_training(dev, seed):
    """`training` of the default one-GPU line: the eager step, and `hipgraph` = the same step replayed from a captured
    graph."""
    training = None
    try:
        training = training_leg(dev, 0, 1, 12, 6, 1, seed)
        g = training_leg(dev, 0, 1, 16, 3, 1, seed, graph=True)
        training["hipgraph"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "final_loss")}
    except Exception as e:      # the headline number must survive a failure of the extra leg
        err = "%s: %s" % (type(e).__name__, e)
        training = dict(error=err) if training is None else dict(training, hipgraph_error=err)
    return training


def train_mode_main(a, dev, rank, world, dist, seed):
    """`--mode train | train-graph`: the JSON line of the training run (rank 0 prints it; with several ranks also the
    peer-mailbox legs, each guarded so that the collectives leg's line is on record whatever they do)."""
    tr = training_leg(dev, rank, world, a.steps, a.warmup, a.batch, seed, graph=a.mode == "train-graph")

    def line():
        cfg = dict(workload="FlyingThings3D 540x960 (run 544x960) D=192 temporal T=2 training step, "
                            "batch %d/GPU" % a.batch,
                   run_hw=[RUN_H, RUN_W], max_disp=MAX_DISP, batch_per_gpu=a.batch, parallelism="dp%d" % world,
                   exec_mode=a.mode)
        return json.dumps(dict(metric="stereo pairs/sec, TRAINING step, FlyingThings3D 540x960 D=192 T=2 "
                                      "(aggregation hot path)",
                               value=tr["value"], unit="pairs/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                               ms_per_step=tr["ms_per_step"], higher_is_better=True, scaling="weak",
                               vs_baseline=None, dtype="f32", data="synthetic", config=cfg, training=tr))
    if world > 1 and os.environ.get("TS_BENCH_PEER", "1") != "0":
        if rank == 0:
            # the collectives leg is on record whatever the legs below do (a reader takes the LAST line)
            print(line(), flush=True)
        # the same step with SyncBatchNorm's exchanges as kernels over the peer mailboxes (eager, then replayed from a
        # hipGraph: legal for world > 1 only in this form).  A mailbox that cannot be mapped, or a peer that does not
        # answer, ends the leg with the reason in its object -- the run degrades to the collectives leg above.
        from temporalstereo_amd.train import graph_replay_safe
        for key, g in (("peer", False), ("peer_hipgraph", True)):
            try:
                if g and not graph_replay_safe():
                    tr[key] = dict(skipped="DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in the environment when the HIP "
                                           "runtime started")
                    continue
                r = training_leg(dev, rank, world, min(a.steps, 6), min(a.warmup, 2), a.batch, seed, graph=g,
                                 sync_bn="peer")
                tr[key] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "mode", "final_loss",
                                             "collectives")}
            except Exception as e:
                tr[key] = dict(error="%s: %s" % (type(e).__name__, e),
                               degraded_to="collectives leg (the line's own ms_per_step)")
                break
    if rank == 0:
        print(line(), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def multi_gpu_training_child(world, batch):
    """The data-parallel training step on the same N GPUs as its own job with a timeout (`bench.py --mode train --gpus
    N`): a rank that hangs inside a collective must not take the headline line with it.  Returns the child's `training`
    object."""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME",
            "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")
    env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
    try:
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--gpus",
                                  str(world), "--steps", "10", "--warmup", "4", "--batch", str(batch)],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 start_new_session=True)      # its own process group: the launcher AND its ranks
        try:
            so, se = child.communicate(timeout=float(os.environ.get("TS_BENCH_TRAIN_TIMEOUT", "300")))
            lines = [ln for ln in so.splitlines() if ln.startswith("{")]
            if child.returncode == 0 and lines:
                return json.loads(lines[-1])["training"]
            return dict(error="exit code %d: %s" % (child.returncode, (se or so)[-400:]))
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)                              # exactly the group started above
            so, _ = child.communicate()
            lines = [ln for ln in (so or "").splitlines() if ln.startswith("{")]
            if lines:       # the collectives leg had finished (its line goes out before the peer legs start)
                return dict(json.loads(lines[-1])["training"], later_legs="timed out")
            return dict(error="timed out (a rank stuck in a collective?)")
    except Exception as e:          # the headline must survive anything the extra leg does
        return dict(error="%s: %s" % (type(e).__name__, e))
