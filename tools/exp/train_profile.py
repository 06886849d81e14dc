import sys, os, cProfile, pstats, io, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed).train()
inputs = bench.make_inputs(dev, seed, 1)
def step():
    net.zero_grad(set_to_none=True)
    out = net(*inputs, {})
    loss = sum(d.abs().mean() for d in out[0])
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
