import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import bench, synth
from temporalstereo_amd import layers
from temporalstereo_amd.aggregation.engine import InferenceEngine
from oracle import aggregation as oagg
H, W, ns, nl = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda:0"); seed = synth.SEED0 + 3; B = 2
net = bench.build_model(dev, seed, ns); inputs = bench.make_inputs(dev, seed, B, (H, W)); bench.calibrate_batchnorm(net, inputs)
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
cpu_in = bench.make_inputs(torch.device("cpu"), seed, B, (H, W)); torch.set_num_threads(16)
with torch.no_grad(): o0 = oagg.aggregate(sd, *cpu_in, {}, cfg=dict(coarse=dict(num_sample=ns)))
mem = {k: v.clone() for k, v in o0[5]["cost_memory"].items()}
local = torch.nn.functional.interpolate(o0[0][0], size=(H // 8, W // 8), mode="bilinear", align_corners=True) / 8.0
local = torch.cat([local + 0.75 * k for k in range(nl)], 1).contiguous()
prev_c = {"cost_memory": mem, "use_past_cost": True, "local_map": local, "local_map_size": nl}
prev_g = {"cost_memory": {k: v.to(dev) for k, v in mem.items()}, "use_past_cost": True, "local_map": local.to(dev), "local_map_size": nl}
with torch.no_grad(): o1 = oagg.aggregate(sd, *cpu_in, dict(prev_c), cfg=dict(coarse=dict(num_sample=ns)))
outs = {}
for name in ("torch", "hip"):
    layers.set_conv_backend(name)
    with torch.no_grad(): outs["module[%s]" % name] = net(*inputs, dict(prev_g))
layers.set_conv_backend("hip")
outs["native"] = InferenceEngine(net, backend="native", replay="plan")(*inputs, dict(prev_g))
def stat(a, b):
    d = (a.double().cpu() - b.double().cpu()).abs()
    return "mean %.2e med %.2e far %.3f" % (float(d.mean()), float(d.median()), float((d > 0.1).double().mean()))
for k, v in outs.items():
    print("%-14s vs oracle: full  %s | fine-level disp %s | coarse %s" % (k, stat(v[0][0], o1[0][0]), stat(v[0][2], o1[0][2]), stat(v[0][3], o1[0][3])))
print("native vs module[hip]: full", stat(outs["native"][0][0], outs["module[hip]"][0][0]), "| fine", stat(outs["native"][0][2], outs["module[hip]"][0][2]))
print("native vs module[torch]: full", stat(outs["native"][0][0], outs["module[torch]"][0][0]))
