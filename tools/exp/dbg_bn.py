import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, synth, copy
from temporalstereo_amd import layers
dev = torch.device("cuda:0")
case = 5
rng = np.random.RandomState(40 + case)
B = int(rng.choice([1, 2, 3])); cin, cout = int(rng.choice([3, 8, 16])), int(rng.choice([4, 8, 32]))
D, H, W = int(rng.randint(2, 6)), int(rng.randint(5, 20)), int(rng.randint(6, 40))
print("B cin cout D H W", B, cin, cout, D, H, W)
for act in (None, "SiLU", "ReLU"):
  for bias in (False, True):
    m = layers.Conv3d(cin, cout, (3, 1, 1), (2, 1, 1), (1, 0, 0), (1, 1, 1), bias=bias, norm=("BN3d", cout), activation=act).to(dev).train()
    ref = copy.deepcopy(m)
    x = torch.from_numpy(synth.normal(65, "x", (B, cin, D, H, W))).to(dev)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = m(xa)
    layers.set_conv_backend("torch"); yb = ref(xb); layers.set_conv_backend("hip")
    g = torch.randn_like(yb)
    ya.backward(g); yb.backward(g)
    # fp64 truth
    r64 = copy.deepcopy(ref).double(); x64 = x.double().requires_grad_(True)
    layers.set_conv_backend("torch"); y64 = r64(x64); layers.set_conv_backend("hip")
    y64.backward(g.double())
    def rel(a, b): return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-12))
    print(act, bias, "y", rel(ya, y64), rel(yb, y64), "| gx", rel(xa.grad, x64.grad), rel(xb.grad, x64.grad),
          "| ggamma", rel(m.norm.weight.grad, r64.norm.weight.grad), rel(ref.norm.weight.grad, r64.norm.weight.grad),
          "| gw", rel(m.weight.grad, r64.weight.grad), rel(ref.weight.grad, r64.weight.grad))
