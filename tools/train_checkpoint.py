#!/usr/bin/env python
"""Train the aggregation path on planted-disparity scenes (tests/synth.stereo_sequence) with the product's own training step
(temporalstereo_amd.train.TrainStep: HIP forward + backward, the reference's loss weights, clip 0.1, RMSprop 1e-3) and write the
state dict as an .npz -- the "contractive checkpoint" of VERDICT round 2, item 1.  Runs on the GPU box:

    python tools/train_checkpoint.py --steps 1500 --out gpurun_out/ckpt.npz

The result is DATA produced by this repository's code; it is committed as tests/golden/ckpt_planted.npz and tools/gen_golden.py
then runs the imported REFERENCE with it on CPU to make the full-size fixtures the end-to-end tests are held to.
Scenes come from a pool of worker processes (numpy only); geometry cycles over the BASELINE configurations so that one set of
weights serves all of them (the modules are convolutional over D: `num_sample` is not a weight shape).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (H, W, max_disp, fx, baseline, frames, local_map_size, batch): BASELINE configs[1..4] geometries (tests/parity_tools.CONFIGS)
GEOMETRIES = [
    (544, 960, 192, 1050.0 * 544 / 540, 1.0, 2, 1, 2),
    (480, 640, 128, 320.0, 0.25, 4, 3, 2),
    (544, 960, 192, 1050.0 * 544 / 540, 1.0, 1, 0, 2),
    (384, 1248, 192, 721.5377, 0.54, 2, 3, 2),
    (544, 960, 192, 1050.0 * 544 / 540, 1.0, 3, 1, 1),
]
SEED_BASE = 77_000_000          # training scenes; the tests use seeds around synth.SEED0 = 20260928


def make_scene(job):
    import synth
    step, geo = job
    H, W, max_disp, fx, baseline, frames, n_local, B = geo
    s = synth.stereo_sequence(SEED_BASE + step, B, H, W, frames=frames, max_disp=max_disp, fx=fx, baseline=baseline)
    return step, geo, s


def to_device(scene, dev, torch):
    T = lambda a: torch.from_numpy(a).to(dev, non_blocking=True)
    frames = [([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir)) for lf, rf, il, ir in scene["frames"]]
    return frames, [T(g) for g in scene["gt"]], T(scene["K"]), [T(t) for t in scene["T"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--workers", type=int, default=24)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ckpt_planted.npz"))
    ap.add_argument("--init", default=None, help="continue from this .npz")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--budget-s", type=float, default=1500.0, help="stop early (and save) after this much wall time")
    a = ap.parse_args()

    import torch
    import bench
    import synth
    from temporalstereo_amd.train import TrainStep
    dev = torch.device("cuda:0")
    torch.manual_seed(a.seed)
    net = bench.build_model(dev, synth.SEED0 + a.seed)
    net.weight_init()                                # the reference's initialiser (coarse.py:52-67): BatchNorm gamma 1, beta 0
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.reset_running_stats()
    if a.init:
        net.load_state_dict({k: torch.from_numpy(v) for k, v in np.load(a.init).items()}, strict=True)
    step = TrainStep(net, max_disp=192, local_map_size=1, lr=a.lr, graph=False)
    eye = {}
    log = []
    t0 = time.time()
    jobs = [(i, GEOMETRIES[i % len(GEOMETRIES)]) for i in range(a.steps)]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)

    def save(path):
        np.savez(path, **{k: v.detach().cpu().numpy() for k, v in net.state_dict().items()})

    def evaluate(tag):
        """Eval-mode EPE per frame on held-out scenes of every geometry (module path: HIP convolutions, running statistics)."""
        from temporalstereo_amd import temporal
        rows = []
        net.eval()
        for gi, geo in enumerate(GEOMETRIES[:4]):
            H, W, max_disp, fx, baseline, frames, n_local, B = geo
            sc = synth.stereo_sequence(SEED_BASE - 1 - gi, 1, H, W, frames=frames, max_disp=max_disp, fx=fx, baseline=baseline)
            fr, gt, K, T = to_device(sc, dev, torch)
            net.coarse.num_sample = max_disp // 16
            I = torch.eye(4, device=dev).expand(1, 4, 4).contiguous()
            info = {}
            with torch.no_grad():
                for t in range(frames):
                    if t > 0:
                        info = temporal.update_map(dict(info), K, T[t], I, baseline, H, W, use_past_cost=True, local_map_size=n_local)
                    state = {k: v for k, v in info.items() if k in ("cost_memory", "use_past_cost", "local_map", "local_map_size") and v is not None}
                    out = net(*fr[t], state)
                    info = out[5]
                    rows.append(dict(tag=tag, geometry=[H, W, max_disp], frame=t, epe=float((out[0][0] - gt[t]).abs().mean()),
                                     epe_quarter=float((torch.nn.functional.interpolate(out[0][1] * 4, size=(H, W), mode="bilinear", align_corners=True) - gt[t]).abs().mean())))
        return rows

    with mp.get_context("spawn").Pool(a.workers) as pool:
        done = 0
        for i, geo, scene in pool.imap(make_scene, jobs, chunksize=1):
            H, W, max_disp, fx, baseline, frames, n_local, B = geo
            fr, gt, K, T = to_device(scene, dev, torch)
            cur = fr[-1]
            fr[-1] = ([x.requires_grad_(True) for x in cur[0]], [x.requires_grad_(True) for x in cur[1]], cur[2], cur[3])
            if B not in eye:
                eye[B] = torch.eye(4, device=dev).expand(B, 4, 4).contiguous()
            poses = [(T[t], eye[B]) for t in range(frames)]
            net.coarse.num_sample = max_disp // 16
            step.l1.max_disp = step.wars.max_disp = max_disp
            step.local_map_size, step.baseline = n_local, baseline
            frac = i / max(a.steps - 1, 1)
            lr = a.lr * (1.0 if frac < 0.6 else (0.3 if frac < 0.85 else 0.1))
            for g in step.opt.param_groups:
                g["lr"] = lr
            loss = step(fr, gt[-1], K, poses)
            done += 1
            if i % 25 == 0 or i == a.steps - 1:
                row = dict(step=i, loss=float(loss), lr=lr, geometry=[H, W, max_disp, frames], elapsed_s=time.time() - t0)
                log.append(row)
                print(json.dumps(row), flush=True)
            if i and i % 250 == 0:
                rows = evaluate("step %d" % i)
                log.extend(rows)
                print(json.dumps(rows), flush=True)
                save(a.out)
            if time.time() - t0 > a.budget_s:
                print("budget reached at step %d" % i, flush=True)
                break
    rows = evaluate("final")
    log.extend(rows)
    print(json.dumps(rows), flush=True)
    save(a.out)
    with open(os.path.splitext(a.out)[0] + "_log.json", "w") as fh:
        json.dump(dict(args=vars(a), steps_done=done, seconds=time.time() - t0, log=log), fh, indent=1)
    print("wrote", a.out, os.path.getsize(a.out), "bytes", flush=True)


if __name__ == "__main__":
    main()
