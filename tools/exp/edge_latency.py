"""Cross-stream edge cost main -> s -> main for freshly created high-priority streams."""
import sys, os, time, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from temporalstereo_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
a = torch.zeros(1024, device=dev); b = torch.zeros(1024, device=dev)
main = torch.cuda.current_stream()
def P(s): return ctypes.c_void_p(s.cuda_stream)
def tiny(s):
    L.ts_copy_rows_fwd(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), 1, 1024, 1024, 1024, P(s))
def pingpong(s, n=200):
    for _ in range(10):
        tiny(main); L.ts_stream_fork(P(main), P(s)); tiny(s); L.ts_stream_fork(P(s), P(main))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tiny(main); L.ts_stream_fork(P(main), P(s)); tiny(s); L.ts_stream_fork(P(s), P(main))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def same(n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        tiny(main); tiny(main)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print("two tiny kernels on main: %.1f us" % same())
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
streams = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(12)]
for i, s in enumerate(streams):
    print("stream %2d (id %d): round trip %.1f us" % (i, s.stream_id, pingpong(s)), flush=True)
print("again:", " ".join("%.0f" % pingpong(s) for s in streams))
