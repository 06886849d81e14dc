"""Back-to-back latency of representative chain kernels in isolation (same stream, dependent launches)."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from temporalstereo_amd import _lib
from temporalstereo_amd.aggregation import native
import temporalstereo_amd.aggregation.blocks as blocks
dev = torch.device("cuda:0")
native._chunk_cap(int(os.environ.get("CAP", "8")))
def sep(cin, cout, stride=1, dil=1):
    m = blocks.DepthwiseConv3D(cin, cout, 3, stride, dil, dil).to(dev).eval()
    return native.SepConv(m)
cases = [("coarse 32->32 (12,34,60)", sep(32, 32), (1, 32, 12, 34, 60)),
         ("coarse s2 32->64 (12,34,60)", sep(32, 64, 2), (1, 32, 12, 34, 60)),
         ("1/32 64->64 (6,17,30)", sep(64, 64), (1, 64, 6, 17, 30)),
         ("1/64 64->64 (3,9,15)", sep(64, 64), (1, 64, 3, 9, 15)),
         ("fine 16->16 (5,68,120)", sep(16, 16), (1, 16, 5, 68, 120)),
         ("precise 8->8 (5,136,240)", sep(8, 8), (1, 8, 5, 136, 240))]
for name, sc, shp in cases:
    x = torch.randn(*shp, device=dev)
    with torch.no_grad():
        rec = _lib.Recorder()
        for _ in range(3): sc(x)
        torch.cuda.synchronize()
        with rec:
            y = sc(x)
        torch.cuda.synchronize()
        n = rec and _lib._real_lib().ts_plan_length(rec.plan)
        for _ in range(10): rec.run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): rec.run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%-32s %d launches, %.1f us per SepConv -> %.1f us per launch" % (name, n, dt * 1e6, dt * 1e6 / n), flush=True)
