#!/usr/bin/env python
"""Training-step timing of the aggregation (forward + backward, train-mode BatchNorm) at BASELINE
config-2 sizes with the HIP convolution Functions vs the framework's convolutions (GPU box only).

usage: python tools/train_step_bench.py [--iters N] [--batch B]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import bench, synth  # noqa: E402
from temporalstereo_amd import layers  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=10); ap.add_argument("--batch", type=int, default=1); ap.add_argument("--backends", default="hip,torch"); ap.add_argument("--graph", action="store_true", help="also time the step captured into a hipGraph (whole forward+backward)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    seed = synth.SEED0 + 2
    net = bench.build_model(dev, seed).train()
    inputs = bench.make_inputs(dev, seed, a.batch)
    for backend in a.backends.split(","):
        layers.set_conv_backend(backend)
        def step():
            net.zero_grad(set_to_none=True)
            out = net(*inputs, {})
            loss = sum(d.abs().mean() for d in out[0])
            loss.backward()
            return float(loss.detach())
        for _ in range(3): last = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
        g = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
        if a.graph:
            # whole-step capture (the documented whole-network pattern): the op-by-op step is host-bound
            # (GPU busy ~3 ms of the ~21 ms), a replay is one host call
            def gstep():
                net.zero_grad(set_to_none=True)
                out = net(*inputs, {})
                loss = sum(d.abs().mean() for d in out[0])
                loss.backward()
                return loss
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3): gstep()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            net.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph):
                gloss = gstep()
            for _ in range(3): graph.replay()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.iters): graph.replay()
            torch.cuda.synchronize(); gdt = (time.perf_counter() - t0) / a.iters
            gg = torch.cat([p.grad.flatten() for p in net.parameters() if p.grad is not None])
            print("conv backend %-5s  %8.2f ms/step captured as a hipGraph  loss %.6f  |grad| %.6e" % (backend, gdt * 1e3, float(gloss), float(gg.norm())), flush=True)
        print("conv backend %-5s  %8.2f ms/step (fwd+bwd, batch %d)  loss %.6f  |grad| %.6e" % (backend, dt * 1e3, a.batch, last, float(g.norm())), flush=True)
    layers.set_conv_backend("hip")


if __name__ == "__main__":
    main()
