"""Wide UNet half on a CU-masked stream (chain gets CUs the wide kernels never occupy).  usage: cumask.py [ncu ...]"""
import sys, os, time, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, synth
from temporalstereo_amd import _lib
from temporalstereo_amd.aggregation import native
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
seed = synth.SEED0 + 2
net = bench.build_model(dev, seed); inputs = bench.make_inputs(dev, seed, 1); bench.calibrate_batchnorm(net, inputs)
agg = native.NativeAggregator(net)
(l4, l8, l16), (r4, r8, r16), il, ir = inputs
main = torch.cuda.current_stream()
floor = native._round_trip_us(main, torch.cuda.Stream(device=dev))

def masked_stream(ncu, invert=False):
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        on = i < ncu
        if invert: on = i >= 256 - ncu
        if on: words[i // 32] |= (1 << (i % 32))
    for attempt in range(8):
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
        assert rc == 0, rc
        s = torch.cuda.ExternalStream(st.value, device=dev)
        us = native._round_trip_us(main, s)
        if us <= 1.6 * floor:
            return s, us, attempt
    return s, us, attempt

def full_pass(wide):
    out = ([], [], [], [], [])
    disps, costs, offs, samples, ranges = out
    mainp, fastp = _lib.ctypes.c_void_p(main.cuda_stream), _lib.ctypes.c_void_p(agg.fast.cuda_stream)
    _lib.check(_lib.lib().ts_stream_fork(mainp, fastp), "fork")
    aux = agg.aux
    native._edge(main, aux)
    if wide is not None: native._edge(main, wide)
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
    if wide is None:
        both, mask = agg.precise.unet_features(l4, r4, il, ir)
    else:
        with torch.cuda.stream(wide):
            both, mask = agg.precise.unet_features(l4, r4, il, ir)
    with torch.cuda.stream(agg.fast):
        ds = agg._fine_level(l8, r8, ds, {}, out, joined(mf), joined(ltf))
    _lib.check(_lib.lib().ts_stream_fork(fastp, mainp), "fork")
    if wide is not None: native._edge(wide, main)
    full, d, c, o, s = agg.precise(both, mask, ds, {})
    native._PAR["on"], native._PAR["aux"] = False, None
    native._chunk_cap(32)
    return full

def timeit(wide):
    with torch.no_grad():
        for _ in range(3): full_pass(wide)
        torch.cuda.synchronize()
        rec = _lib.Recorder()
        with rec:
            keep = full_pass(wide)
        torch.cuda.synchronize()
        for _ in range(5): rec.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): rec.run()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 50 * 1e3, keep

base, ref = timeit(None)
print("baseline (wide half on the caller's stream): %.3f ms" % base, flush=True)
plain = torch.cuda.Stream(device=dev)
t, got = timeit(plain)
print("wide half on a plain normal-priority stream: %.3f ms  (max |diff| %.1e)" % (t, float((got - ref).abs().max())), flush=True)
for ncu in [int(a) for a in sys.argv[1:]] or [256, 224, 192, 160, 128]:
    for inv in (False, True):
        s, us, att = masked_stream(ncu, inv)
        t, got = timeit(s)
        print("wide half on %3d CUs (%s bits; stream round trip %.0f us after %d redraws): %.3f ms  (max |diff| %.1e)" %
              (ncu, "high" if inv else "low", us, att, t, float((got - ref).abs().max())), flush=True)
