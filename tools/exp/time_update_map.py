import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from temporalstereo_amd import temporal
dev = torch.device("cuda:0")
B, H, W = 4, 544, 960
prev = {"prev_disp": (torch.rand(B, 1, H, W, device=dev) * 100 + 1),
        "cost_memory": {"disp_sample": torch.rand(B, 2, H // 8, W // 8, device=dev) * 12, "cost_volume": torch.randn(B, 2, H // 8, W // 8, device=dev)},
        "local_map": torch.rand(B, 2, H // 8, W // 8, device=dev) * 12, "local_map_size": 3}
K = torch.from_numpy(synth.sceneflow_intrinsics(B, H, W)).to(dev)
T = torch.from_numpy(synth.small_motion(5, B)).to(dev)
Ti = torch.inverse(T)
def run():
    return temporal.update_map(dict(prev), K, T, Ti, 1.0, H, W, use_past_cost=True, local_map_size=3)
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): run()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("update_map: host %.3f ms, total %.3f ms per frame (B=%d)" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3, B))
for B1 in (1,):
    prev1 = {"prev_disp": prev["prev_disp"][:B1].contiguous(), "cost_memory": {k: v[:B1].contiguous() for k, v in prev["cost_memory"].items()},
             "local_map": prev["local_map"][:B1].contiguous()}
    K1, T1, Ti1 = K[:B1].contiguous(), T[:B1].contiguous(), Ti[:B1].contiguous()
    f = lambda: temporal.update_map(dict(prev1), K1, T1, Ti1, 1.0, H, W, use_past_cost=True, local_map_size=3)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for _ in range(50): f()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("B=%d: host %.3f ms, device span %.3f ms per frame" % (B1, (t1 - t0) / 50 * 1e3, e0.elapsed_time(e1) / 50))
