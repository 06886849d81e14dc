import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "libfill.so"))
L.fill_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
L.fill_k1_run.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
C, D, H, W = 128, 5, 136, 240
vol = torch.empty(2 * C * D * H * W, device=dev)
for order in (0, 1):
    for _ in range(5): L.fill_k1_run(order, vol.data_ptr(), C, D, H, W, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): L.fill_k1_run(order, vol.data_ptr(), C, D, H, W, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print("K1 store pattern order %d: %.1f us for %.1f MB = %.1f GB/s" % (order, us, vol.numel() * 4 / 1e6, vol.numel() * 4 / us / 1e3), flush=True)
names = ["plain", "nontemporal", "buf aux0", "buf nt", "buf sc0sc1", "buf sc0", "chunked"]
for mb in (200,):
    buf = torch.empty(mb * 1000 * 1000 // 4, device=dev)
    nbytes = buf.numel() * 4 // 16 * 16
    for kind, nm in enumerate(names):
        best = None
        for blocks in (256 * 4, 256 * 8, 256 * 16, 256 * 32):
            for _ in range(3): L.fill_run(kind, buf.data_ptr(), nbytes, blocks, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): L.fill_run(kind, buf.data_ptr(), nbytes, blocks, st)
            e1.record(); torch.cuda.synchronize()
            gbs = nbytes * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            if best is None or gbs > best[0]: best = (gbs, blocks)
        print("%5d MB  %-12s best %7.1f GB/s at %5d blocks" % (mb, nm, best[0], best[1]), flush=True)
