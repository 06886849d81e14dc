"""What does a forked branch cost inside a replayed hipGraph on this runtime?  (round 5, training step: could the weight-gradient
launches run beside the input-gradient chain?)

A chain of N small-grid kernels on the capture stream plus M independent ones, captured four ways:
  serial      all N + M on the one stream
  fork1       the M on a side stream: one fork at the start, one join at the end
  forkK       the side stream waits on the main chain K times (every N/K kernels), one join at the end
  forkM       every side kernel waits on its own event of the main chain (the per-layer form round 3 tried)
Prints the replayed time of each.  Needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (train.py sets the same)."""
import os
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import sys
import torch

N = int(os.environ.get("N", 200)); M = int(os.environ.get("M", 100))
dev = torch.device("cuda:0")
a = torch.randn(256, 2048, device=dev); bm = torch.randn(2048, 256, device=dev)
outs_main = [torch.empty(256, 256, device=dev) for _ in range(N)]
outs_side = [torch.empty(256, 256, device=dev) for _ in range(M)]


def capture(mode, K=4):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            torch.mm(a, bm, out=outs_main[0])
        s.synchronize()
        with torch.cuda.graph(g, stream=s):
            cur = torch.cuda.current_stream()
            if mode == "serial":
                for i in range(N): torch.mm(a, bm, out=outs_main[i])
                for j in range(M): torch.mm(a, bm, out=outs_side[j])
            else:
                waits = {"fork1": 1, "forkK": K, "forkM": M}[mode]
                per = max(1, N // waits); done = 0
                for i in range(N):
                    if i % per == 0 and done < waits:
                        ev = torch.cuda.Event(); ev.record(cur); side.wait_event(ev)
                        with torch.cuda.stream(side):
                            for j in range(done * M // waits, (done + 1) * M // waits): torch.mm(a, bm, out=outs_side[j])
                        done += 1
                    torch.mm(a, bm, out=outs_main[i])
                ev = torch.cuda.Event(); ev.record(side); cur.wait_event(ev)
    return g, s


def timed(g, s, reps=30):
    with torch.cuda.stream(s):
        for _ in range(5): g.replay()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for mode in ("serial", "fork1", "forkK", "forkM"):
    g, s = capture(mode)
    print("%-7s N=%d M=%d  %.3f ms per replay" % (mode, N, M, timed(g, s)), flush=True)
# one kernel alone
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(100): torch.mm(a, bm, out=outs_main[0])
e1.record(); e1.synchronize()
print("one mm, issued back to back: %.2f us" % (e0.elapsed_time(e1) * 10))
