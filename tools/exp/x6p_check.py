"""Experiment / acceptance run of ig_conv_x6p_kernel (csrc/conv_x6p.hip): ts_conv3d_hw_x6_fwd against an fp64 convolution on ragged
shapes with the ping-pong form forced (TS_X6P_MIN_WGS=1, TS_X6P_HR=4|8) and the layer timings of tools/exp/x6_bench.py's shapes with
the form on / off.  Each configuration runs in a process of its own (the switches are read once per process).
    python tools/exp/x6p_check.py            # everything
    python tools/exp/x6p_check.py --child acc|time"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

ACC = [  # B, Cin, Cout, D, H, W, addend
    (1, 16, 32, 2, 24, 64, False), (2, 32, 32, 3, 37, 44, False), (1, 176, 24, 5, 34, 60, True), (1, 20, 24, 2, 16, 36, False),
    (1, 64, 64, 1, 40, 72, False), (1, 128, 32, 1, 68, 120, False), (1, 32, 144, 1, 24, 40, False), (1, 512, 17, 1, 32, 64, False),
    (3, 40, 33, 2, 1, 4, False), (1, 31, 80, 1, 5, 8, True), (2, 100, 48, 1, 19, 76, False), (1, 64, 32, 1, 272, 480, False),
    # 9..16 output channels: the row-paired form
    (1, 64, 16, 7, 68, 120, False), (2, 48, 12, 3, 37, 44, True), (1, 128, 16, 1, 68, 120, False), (1, 176, 9, 2, 33, 36, False), (1, 32, 16, 1, 272, 480, False),
]
TIME = [  # name, Cin, Cout, D, H, W
    ("unet 32->32 272x480", 32, 32, 1, 272, 480), ("unet 64->64 136x240", 64, 64, 1, 136, 240), ("unet 128->32 272x480", 128, 32, 1, 272, 480),
    ("unet 64->32 272x480", 64, 32, 1, 272, 480), ("unet 128->32 136x240", 128, 32, 1, 136, 240), ("unet 32->32 136x240", 32, 32, 1, 136, 240),
    ("coarse 128->32 D14 34x60", 128, 32, 14, 34, 60), ("coarse 32->32 D12 34x60", 32, 32, 12, 34, 60), ("conv 128->64 68x120", 128, 64, 1, 68, 120),
    ("fine 64->16 D7 68x120", 64, 16, 7, 68, 120), ("conv 128->16 68x120", 128, 16, 1, 68, 120), ("fine 48->16 D5 68x120", 48, 16, 5, 68, 120),
]


def child_acc():
    import torch
    import torch.nn.functional as F
    from temporalstereo_amd.aggregation import native as N
    N._X6_MIN_GRID = 1
    dev = torch.device("cuda:0")
    worst = 0.0
    for (B, Cin, Cout, D, H, W, add) in ACC:
        for act in (N.ACT_NONE, N.ACT_SILU):
            g = torch.Generator().manual_seed(Cin * 7 + Cout)
            x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
            w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
            f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, act, False, "hw")
            addend = torch.randn(B, Cout, 1, H, W, generator=g).to(dev) if add else None
            out = N.conv_hw(x, f, 1, 1, addend=addend)
            N.X6 = False
            o32 = N.conv_hw(x, f, 1, 1, addend=addend)
            N.X6 = True
            torch.cuda.synchronize()
            ref = F.conv3d(x.double(), w.double(), padding=(0, 1, 1))
            if add:
                ref = ref + addend.double()
            ref = ref * f.scale[:Cout].double().view(1, -1, 1, 1, 1) + f.shift[:Cout].double().view(1, -1, 1, 1, 1)
            if act == N.ACT_SILU:
                ref = F.silu(ref)
            sc = max(float(ref.abs().max()), 1.0)
            e6 = float((out.double() - ref).abs().max()) / sc
            e32 = float((o32.double() - ref).abs().max()) / sc
            worst = max(worst, e6)
            flag = "" if (e6 <= 1e-6 and torch.isfinite(out).all()) else "   <-- FAIL"
            print("acc B%d %3d->%3d D%d %3dx%3d add%d act%d   x6 %.3e   f32 %.3e%s" % (B, Cin, Cout, D, H, W, add, act, e6, e32, flag), flush=True)
    print("worst %.3e" % worst)


def child_time():
    import torch
    from temporalstereo_amd.aggregation import native as N
    N._X6_MIN_GRID = 1
    dev = torch.device("cuda:0")
    for B in (1, 4):
        for name, Cin, Cout, D, H, W in TIME:
            x = torch.randn(B, Cin, D, H, W, device=dev)
            w = torch.randn(Cout, Cin, 1, 3, 3, device=dev) / (9 * Cin) ** 0.5
            f = N.Folded(w, None, None, N.ACT_SILU, False, "hw")
            out = torch.empty(B, Cout, D, H, W, device=dev)
            for _ in range(5):
                N.conv_hw(x, f, 1, 1, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                N.conv_hw(x, f, 1, 1, out=out)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 10.0
            fl = 2.0 * B * Cin * Cout * 9 * D * H * W
            print("time B=%d %-28s %7.1f us  %6.1f TF eq." % (B, name, t, fl / t / 1e6), flush=True)


def child_fuzz():
    """Random geometries (one-row / four-column images, ragged tiles and channel counts, several planes and batch elements, 17..144 output
    channels) with the form forced: ts_conv3d_hw_x6_fwd against the f32-MFMA kernel, to fp32 rounding everywhere."""
    import numpy as np
    import torch
    from temporalstereo_amd.aggregation import native as N
    N._X6_MIN_GRID = 1
    N._X6_MIN_GRID_UNSPLIT = 1
    dev = torch.device("cuda:0")
    seed, cases = int(os.environ.get("TS_FUZZ_SEED", "20260930")), int(os.environ.get("TS_FUZZ_CASES", "60"))
    rng = np.random.RandomState(seed)
    worst = 0.0
    bad = 0
    for case in range(cases):
        B = int(rng.randint(1, 4)); Cin = int(rng.choice([32, 33, 40, 48, 64, 100, 128, 272])); Cout = int(rng.choice([17, 20, 32, 33, 48, 64, 80, 144]))
        D = int(rng.randint(1, 6)); H = int(rng.randint(1, 70)); W = 4 * int(rng.randint(1, 40))
        g = torch.Generator().manual_seed(seed % 100000 * 1000 + 3000 + case)
        x = torch.randn(B, Cin, D, H, W, generator=g).to(dev)
        view = int(rng.randint(0, 4)) if seed != 20260930 else 0      # (the default seed keeps the committed test's cases as they were)
        if view == 1:        # a channel slice of a larger tensor: batch stride > Cin planes, base offset
            big = torch.randn(B, Cin + 5, D, H, W, generator=g).to(dev)
            big[:, 3:3 + Cin] = x
            x = big[:, 3:3 + Cin]
        w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(dev)
        act = [N.ACT_NONE, N.ACT_SILU, N.ACT_RELU][case % 3]
        f = N.Folded(w, torch.randn(Cout, generator=g).to(dev), None, act, False, "hw")
        add = torch.randn(B, Cout, 1, H, W, generator=g).to(dev) if case % 4 == 0 else None
        outs = {}
        for x6 in (True, False):
            N.X6 = x6
            try:
                if view == 2:    # the output as a channel slice of a larger tensor
                    bigo = torch.full((B, Cout + 4, D, H, W), 7.0, device=dev)
                    N.conv_hw(x, f, 1, 1, addend=add, out=bigo[:, 2:2 + Cout])
                    assert bool((bigo[:, :2] == 7.0).all()) and bool((bigo[:, 2 + Cout:] == 7.0).all()), "wrote outside its channels"
                    outs[x6] = bigo[:, 2:2 + Cout].clone()
                else:
                    outs[x6] = N.conv_hw(x, f, 1, 1, addend=add)
            finally:
                N.X6 = True
        torch.cuda.synchronize()
        scale = max(float(outs[False].abs().max()), 1.0)
        err = float((outs[True] - outs[False]).abs().max()) / scale
        worst = max(worst, err)
        ok = err <= 4e-6 and bool(torch.isfinite(outs[True]).all())
        bad += not ok
        if not ok or cases <= 60:
            print("fuzz %2d B%d %3d->%3d D%d %2dx%3d act%d add%d view%d  %.3e%s" % (case, B, Cin, Cout, D, H, W, act, add is not None, view, err, "" if ok else "   <-- FAIL"), flush=True)
    print("seed %d: %d cases, %d failed, worst %.3e" % (seed, cases, bad, worst))


if __name__ == "__main__":
    if "--child" in sys.argv:
        {"acc": child_acc, "time": child_time, "fuzz": child_fuzz}[sys.argv[sys.argv.index("--child") + 1]]()
        sys.exit(0)
    runs = [("acc", {"TS_X6P_MIN_WGS": "1", "TS_X6P_HR": "8"}), ("acc", {"TS_X6P_MIN_WGS": "1", "TS_X6P_HR": "4"}),
            ("time", {"TS_X6P": "0"}), ("time", {"TS_X6P_MIN_WGS": "1"}), ("time", {"TS_X6P_MIN_WGS": "1", "TS_X6P_HR": "4"}),
            ("time", {"TS_X6P_MIN_WGS": "1", "TS_X6P_HR": "8"})]
    for kind, env in runs:
        print("==== %s %s" % (kind, env), flush=True)
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind], env=e, timeout=900)
        print("exit %d" % r.returncode, flush=True)
