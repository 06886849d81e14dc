"""Experiment: poison every torch.empty / empty_like / new_empty with NaN and run the eager training step: a kernel that reads
memory it (or its launcher) never wrote shows up as a NaN loss / gradient."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.exp.dbg_train_graph import setup, MAX_DISP
from temporalstereo_amd.train import TrainStep

net, frames, gt, K, poses = setup()
step = TrainStep(net, max_disp=MAX_DISP, local_map_size=1, clip=0, lr=0.0)
loss = step(frames, gt, K, poses)
print("clean loss", float(loss))

_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
sites = {}


def poison(t):
    if t.is_floating_point() and t.is_cuda:
        t.fill_(float("nan"))
        f = traceback.extract_stack(limit=4)[0:2]
        key = " <- ".join("%s:%d" % (os.path.basename(x.filename), x.lineno) for x in reversed(f))
        sites[key] = sites.get(key, 0) + 1
    return t


torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: poison(_new_empty(self, *a, **k))
loss = step(frames, gt, K, poses)
torch.cuda.synchronize()
print("poisoned loss", float(loss))
bad = [n for n, p in net.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("non-finite gradients:", len(bad), bad[:40])
fg = [t for t in TrainStep._tensors((frames,)) if t.grad is not None]
print("feature grads finite:", [bool(torch.isfinite(t.grad).all()) for t in fg])
