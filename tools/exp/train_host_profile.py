"""cProfile of the host side of one training step (GPU box): where the ~30 ms go."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, synth
from temporalstereo_amd.train import TrainStep
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed)
frames = []
for t in range(2):
    lf, rf, il, ir = bench.make_inputs(dev, seed + 1000 * t, 1)
    if t == 1:
        lf, rf = [x.requires_grad_(True) for x in lf], [x.requires_grad_(True) for x in rf]
    frames.append((lf, rf, il, ir))
bench.calibrate_batchnorm(net, frames[0])
gt = torch.from_numpy(synth.smooth(synth.normal(seed, "gt", (1, 1, 544, 960))) * 20.0 + 70.0).to(dev)
K = torch.from_numpy(synth.sceneflow_intrinsics(1, 544, 960)).to(dev)
T = torch.from_numpy(synth.small_motion(seed, 1)).to(dev)
eye = torch.eye(4, device=dev).expand(1, 4, 4).contiguous()
step = TrainStep(net, max_disp=192, local_map_size=1)
for _ in range(3):
    step(frames, gt, K, [(eye, eye), (T, eye)])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step(frames, gt, K, [(eye, eye), (T, eye)])
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])
