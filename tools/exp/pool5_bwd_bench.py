import sys; sys.path.insert(0, ".")
import torch
import temporalstereo_amd.functional as TF
dev = torch.device("cuda:0")
for shape in [(1, 32, 14, 34, 60), (1, 16, 7, 68, 120), (4, 32, 14, 34, 60)]:
    x = torch.randn(*shape, device=dev, requires_grad=True)
    a, m = TF.pool5_avgmax(x)
    ga, gm = torch.randn_like(a), torch.randn_like(m)
    for _ in range(5):
        x.grad = None; a, m = TF.pool5_avgmax(x); torch.autograd.backward([a, m], [ga, gm])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    a, m = TF.pool5_avgmax(x)
    e0.record()
    for _ in range(50):
        x.grad = None
        torch.autograd.backward([a, m], [ga, gm], retain_graph=True)
    e1.record(); e1.synchronize()
    print(shape, "pool5 backward (avg + max) %.1f us" % (e0.elapsed_time(e1) * 20))
