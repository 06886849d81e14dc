import ctypes, os, subprocess, sys, time, torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libgridbar.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "grid_barrier.hip")])
L = ctypes.CDLL(so)
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n, nops = 24480 * 32, 40          # a coarse-level activation: 32 channels x 12 x 34 x 60
a = torch.rand(n, device=dev); b = torch.zeros(n, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
ref = None
for blocks in (64, 128, 256):
    a0 = a.clone()
    L.run_separate(P(a0), P(b), n, nops, blocks, st); torch.cuda.synchronize(); r1 = a0.clone()
    a1 = a.clone()
    L.run_persistent(P(a1), P(b), n, nops, blocks, P(cnt), st); torch.cuda.synchronize()
    ok = torch.equal(a1, r1)
    ts = timeit(lambda: L.run_separate(P(a0), P(b), n, nops, blocks, st))
    tp = timeit(lambda: L.run_persistent(P(a1), P(b), n, nops, blocks, P(cnt), st))
    print("%3d workgroups, %d dependent ops: separate launches %.1f us/op, persistent + grid barrier %.1f us/op, results equal: %s"
          % (blocks, nops, ts / nops, tp / nops, ok), flush=True)
