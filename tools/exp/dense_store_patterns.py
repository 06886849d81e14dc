"""Experiment (round 4): store order of the dense siblings' output volume, stores only (tools/exp/dense_store_patterns.hip)."""
import ctypes, os, subprocess, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libdsp.so")
L = ctypes.CDLL(so)
L.run_pat.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, C, D, H, W = 4, 32, 48, 136, 240
vol = torch.empty(B * 2 * C * D * H * W, device=dev)
nbytes = vol.numel() * 4
def t(kind, tr, dchunk, pin, iters=10):
    for _ in range(2): L.run_pat(kind, tr, dchunk, pin, vol.data_ptr(), B, C, D, H, W, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): L.run_pat(kind, tr, dchunk, pin, vol.data_ptr(), B, C, D, H, W, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
us = t(0, 0, 48, 0); print("dense_warp order                          %7.1f us  %6.1f GB/s" % (us, nbytes / us / 1e3), flush=True)
for chunks in (1024, 2048, 4096, 16384):
    us = t(1, chunks, 48, 0); print("linear, %5d workgroup chunks           %7.1f us  %6.1f GB/s" % (chunks, us, nbytes / us / 1e3), flush=True)
for tr in (2, 4, 8, 136):
    for dchunk in (48, 12, 4):
        for pin in (1, 0):
            us = t(2, tr, dchunk, pin)
            print("runs of %3d rows, %2d candidates per wg, planes %s  %7.1f us  %6.1f GB/s" % (tr, dchunk, "inner" if pin else "outer", us, nbytes / us / 1e3), flush=True)

# K1's own pattern at 1 and 4 pairs (167 / 669 MB of the two stored halves)
L.run_k1.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p]
C, D = 128, 5
for Bk in (1, 4, 8):
    v2 = torch.empty(Bk * 2 * C * D * H * W, device=dev)
    for order in (0, 1, 2, 3):
        for _ in range(3): L.run_k1(order, v2.data_ptr(), Bk, C, D, H, W, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.run_k1(order, v2.data_ptr(), Bk, C, D, H, W, st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("K1 store pattern, %d pairs, %s: %7.1f us  %6.1f GB/s" % (Bk, ("rows outer", "planes outer, 4 rows of a plane back to back", "row pairs, the 2 rows of a plane back to back", "row pairs, 4 planes of row a then of row b")[order], us, v2.numel() * 4 / us / 1e3), flush=True)
