"""How long do the coarse+fine levels (the latency chain) take WITHOUT the wide UNet half and the precise tail?"""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd import _lib
from temporalstereo_amd.aggregation import native
dev = torch.device("cuda:0")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, 1); bench.calibrate_batchnorm(net, inputs)
agg = native.NativeAggregator(net)
(l4, l8, l16), (r4, r8, r16), il, ir = inputs

def chain(with_wide):
    out = ([], [], [], [], [])
    main = torch.cuda.current_stream()
    mainp, fastp = _lib.ctypes.c_void_p(main.cuda_stream), _lib.ctypes.c_void_p(agg.fast.cuda_stream)
    _lib.check(_lib.lib().ts_stream_fork(mainp, fastp), "fork")
    aux = agg.aux
    native._edge(main, aux)
    native._PAR["on"], native._PAR["aux"] = True, aux
    native._chunk_cap(8)
    with torch.cuda.stream(aux):
        mc, mf = agg.coarse.up.mask(l16), agg.fine.up.mask(l8)
        ltf = agg.fine.left_term(l8)
    waited = []
    def joined(m):
        def get():
            if not waited:
                native._edge(aux, torch.cuda.current_stream()); waited.append(True)
            return m
        return get
    with torch.cuda.stream(agg.fast):
        ds = agg._coarse_level(l16, r16, {}, out, joined(mc))
    if with_wide:
        both, mask = agg.precise.unet_features(l4, r4, il, ir)
    with torch.cuda.stream(agg.fast):
        ds = agg._fine_level(l8, r8, ds, {}, out, joined(mf), joined(ltf))
    _lib.check(_lib.lib().ts_stream_fork(fastp, mainp), "fork")
    native._PAR["on"], native._PAR["aux"] = False, None
    native._chunk_cap(32)
    return ds

for with_wide in (False, True, False, True):
    with torch.no_grad():
        for _ in range(3): chain(with_wide)
        torch.cuda.synchronize()
        rec = _lib.Recorder()
        with rec:
            keep = chain(with_wide)
        torch.cuda.synchronize()
        for _ in range(5): rec.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): rec.run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("coarse+fine chain %s the wide UNet half: %.3f ms" % ("WITH" if with_wide else "without", dt * 1e3), flush=True)
