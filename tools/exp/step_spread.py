"""Run-to-run spread of the first-step gradients of the training step (GPU box only): the same step twice per setting of
TS_TRAIN_WGRAD_DEFER, worst relative L2 over the parameters -- what a test of a structural change of the step has to allow."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_train_step_gpu as T
def run(v):
    os.environ["TS_TRAIN_WGRAD_DEFER"] = v
    _, g, _ = T._run(False, steps=1)
    return g
a, b, c, d = run("0"), run("0"), run("1"), run("1")
def worst(x, y):
    top = max(float(t.norm()) for t in x.values()); w = (0, None)
    for k in x:
        n = float(x[k].norm())
        if n < 1e-5 * top: continue
        e = float((x[k] - y[k]).norm()) / n
        if e > w[0]: w = (e, k)
    return w
print("0 vs 0", worst(a, b)); print("1 vs 1", worst(c, d)); print("0 vs 1", worst(a, c))
k = "coarse.past_conv.weight"
print(k, float(a[k].norm()), float((a[k]-b[k]).norm()), float((a[k]-c[k]).norm()), max(float(t.norm()) for t in a.values()))
