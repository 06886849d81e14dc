"""Experiment: x6 error against an fp64 convolution over channel counts / shapes, next to the f32 kernel's (tests/test_conv_x6_gpu._run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TS_CONV_X6_MIN_GRID", "1")
from test_conv_x6_gpu import _run
from temporalstereo_amd.aggregation import native as N
for (B, Cin, Cout, D, H, W, dil, add) in [(1, 128, 16, 1, 34, 60, 1, False), (1, 176, 16, 1, 34, 60, 1, False), (1, 176, 12, 1, 34, 60, 1, False),
                                          (1, 176, 12, 5, 34, 60, 1, False), (1, 176, 12, 5, 34, 60, 1, True), (1, 176, 32, 5, 34, 60, 1, True),
                                          (1, 256, 16, 1, 34, 60, 1, False), (1, 512, 16, 1, 32, 64, 1, False), (1, 64, 16, 1, 32, 64, 1, False),
                                          (1, 16, 16, 1, 32, 64, 1, False)]:
    for act in (N.ACT_NONE,):
        e6, e32, scale = _run(B, Cin, Cout, D, H, W, dil, act, add, seed=Cin * 7 + Cout)
        print("Cin %4d Cout %3d D %d addend %-5s: x6 %.2e  f32 %.2e  scale %.1f  ratio %.2f" % (Cin, Cout, D, add, e6, e32, scale, e6 / e32))
