"""Per-call timing of every native op in one config-2 aggregation pass (debug tool)."""
import sys, os, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd.aggregation import native
from temporalstereo_amd import functional as TF
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, int(sys.argv[1]) if len(sys.argv) > 1 else 1); bench.calibrate_batchnorm(net, inputs)
agg = native.NativeAggregator(net)
agg.overlap = False
native._chunk_cap(int(os.environ.get('CAP', '8')))
records = []
def wrap(mod, name, describe):
    orig = getattr(mod, name)
    def f(*a, **k):
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); out = orig(*a, **k); e.record(); torch.cuda.synchronize()
        records.append((name, describe(*a, **k), s.elapsed_time(e) * 1e3)); return out
    setattr(mod, name, f)
import math
def _gf(x, f, taps, stride, transposed):
    n = math.prod(x.shape[2:]) * x.shape[0]
    n = n * (stride * stride if transposed else 1) / (1 if transposed else stride * stride)
    return " GF=%.3f" % (2.0 * f.cin * f.cout * taps * n / 1e9)
wrap(native, "conv_hw", lambda x, f, stride=1, dilation=1, transposed=False, **k: "%d->%d %s s%d d%d %s" % (f.cin, f.cout, tuple(x.shape[2:]), stride, dilation, "T" if transposed else "") + _gf(x, f, 4 if transposed and stride == 2 else 9, stride, transposed))
wrap(native, "conv_d", lambda x, f, k, stride=1, dilation=1, padding=0, transposed=False, **kw: "%d->%d %s k%d s%d d%d %s" % (f.cin, f.cout, tuple(x.shape[2:]), k, stride, dilation, "T" if transposed else "") + _gf(x, f, k, 1, False))
wrap(native, "resize_add_act", lambda a, add, size, **k: "%s->%s" % (tuple(a.shape[1:]), tuple(size)))
wrap(native, "pool5", lambda x, a, m: str(tuple(x.shape[1:])))
wrap(native.TF, "block_cost", lambda l, r, d, s=3: "%s D=%s" % (tuple(l.shape[1:]), d if isinstance(d, int) else d.shape[1]))
wrap(native.TF, "topk_softargmax", lambda c, s, o, k=2: str(tuple(c.shape[1:])))
for _ in range(2):
    records.clear()
    t0 = time.perf_counter(); agg(*inputs, {}); torch.cuda.synchronize(); wall = time.perf_counter() - t0
tot = sum(r[2] for r in records)
print("ops %d, sum of op times %.2f ms (wall with syncs %.2f ms)" % (len(records), tot / 1e3, wall * 1e3))
agg_t = collections.OrderedDict()
for n, d, us in records:
    key = n + " " + d
    a = agg_t.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += us
for key, (n, us) in sorted(agg_t.items(), key=lambda kv: -kv[1][1])[:45]:
    gf = float(key.split("GF=")[1]) if "GF=" in key else 0.0
    print("%8.1f us  x%-2d  %s  %s" % (us, n, key, ("%.1f TF/s" % (gf * n / us / 1e3 * 1e3)) if gf else ""))
