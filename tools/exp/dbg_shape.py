import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")
import bench, synth
from temporalstereo_amd import layers
from temporalstereo_amd.aggregation.engine import InferenceEngine
H, W, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0"); seed = synth.SEED0 + 3; B = 2
net = bench.build_model(dev, seed, ns); inputs = bench.make_inputs(dev, seed, B, (H, W)); bench.calibrate_batchnorm(net, inputs)
for backend in ("torch", "hip"):
    layers.set_conv_backend(backend)
    with torch.no_grad(): ref = net(*inputs, {})
    layers.set_conv_backend("hip")
    for replay in ("eager", "plan"):
        eng = InferenceEngine(net, backend="native", replay=replay)
        if replay == "eager": eng.net.overlap = False
        got = eng(*inputs, {})
        print("module[%s] vs native[%s]:" % (backend, replay),
              " disp", ["%.2e" % float((a - b).abs().mean()) for a, b in zip(got[0], ref[0])],
              " cost", ["%.2e" % float((a - b).abs().mean()) for a, b in zip(got[1], ref[1])],
              " samp", ["%.2e" % float((a - b).abs().mean()) for a, b in zip(got[2], ref[2])])
# CPU oracle as the arbiter
from oracle import aggregation as oagg
sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
cpu_in = bench.make_inputs(torch.device("cpu"), seed, B, (H, W))
torch.set_num_threads(16)
with torch.no_grad():
    orc = oagg.aggregate(sd, *cpu_in, {}, cfg=dict(coarse=dict(num_sample=ns)))
for name, run in (("module[torch]", "torch"), ("module[hip]", "hip"), ("native", None)):
    if run:
        layers.set_conv_backend(run)
        with torch.no_grad(): out = net(*inputs, {})
        layers.set_conv_backend("hip")
    else:
        out = InferenceEngine(net, backend="native", replay="plan")(*inputs, {})
    print("%-14s vs CPU oracle: disp" % name, ["%.2e" % float((a.cpu() - b).abs().mean()) for a, b in zip(out[0], orc[0])],
          " cost", ["%.2e" % float((a.cpu() - b).abs().mean()) for a, b in zip(out[1], orc[1])])
