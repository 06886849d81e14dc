"""Experiment: replay the captured training step with a frozen optimizer (lr 0) and list the gradients that change between replays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.exp.dbg_train_graph import setup, dev, MAX_DISP  # noqa
from temporalstereo_amd.train import TrainStep

lr = float(os.environ.get("LR", "0"))
net, frames, gt, K, poses = setup()
for m in net.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and os.environ.get("FREEZE_BN_STATS", "1") == "1":
        m.momentum = 0.0
step = TrainStep(net, max_disp=MAX_DISP, local_map_size=1, graph=os.environ.get("GRAPH", "1") == "1", clip=0, lr=lr)
step._debug_keep = True
ref = None
for it in range(4):
    loss = step(frames, gt, K, poses)
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    fg = [t.grad.detach().clone() for t in TrainStep._tensors(step._static if step.graph else (frames, gt, K, poses)) if t.grad is not None]
    print("step", it, "loss %.6f" % float(loss), "feature-grad norms", ["%.4g" % float(g.norm()) for g in fg])
    for k, v in step._dbg.items():
        vs = list(v.values()) if isinstance(v, dict) else v
        for i, t in enumerate(vs):
            if torch.is_tensor(t) and t.is_floating_point():
                g = t.grad if t.requires_grad else None
                print("     %-8s[%d] %-22s sum %.6e finite %s | grad %s" % (k, i, tuple(t.shape), float(t.double().sum()), bool(torch.isfinite(t).all()),
                      "-" if g is None else "sum %.6e finite %s" % (float(g.double().sum()), bool(torch.isfinite(g).all()))))
    if ref is None:
        ref = cur
        continue
    worst = []
    for n in cur:
        d = float((cur[n] - ref[n]).abs().max())
        s = float(ref[n].abs().max())
        if not (d <= 1e-3 * s + 1e-6):
            worst.append((n, d, s))
    print("   params whose gradient moved:", len(worst), "of", len(cur))
    for n, d, s in worst[:12]:
        print("      %-55s max|d| %.4g  (max|ref| %.4g)" % (n, d, s))
