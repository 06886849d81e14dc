"""Experiment (CPU, numpy/torch): accuracy of fp32 products emulated by bf16 splits with fp32 accumulation, for the reduction
lengths of the wide convolution layers (K = Cin x 9 taps).  a = a0 + a1 + a2 (+ residual), each part a bf16:
  x3: a0 b0 + a0 b1 + a1 b0                      (the classic three-product split)
  x6: x3 + a1 b1 + a0 b2 + a2 b0                 (all terms down to 2^-16 of the product)
  x9: every pair
Errors are relative to the RMS of the exact (fp64) results; `fp32` is a plain fp32 dot product in the same order."""
import numpy as np
import torch

torch.manual_seed(0)


def split3(x):
    p0 = x.to(torch.bfloat16).to(torch.float32)
    r = x - p0
    p1 = r.to(torch.bfloat16).to(torch.float32)
    r = r - p1
    p2 = r.to(torch.bfloat16).to(torch.float32)
    return p0, p1, p2


def run(K, N=4096, M=64, dist="normal"):
    a = torch.randn(N, K) if dist == "normal" else torch.rand(N, K) * 4 - 1       # activations (post-SiLU like for 'uniform')
    b = torch.randn(K, M) / K ** 0.5
    exact = a.double() @ b.double()
    rms = float(exact.pow(2).mean().sqrt())
    out = {"fp32": (a @ b)}
    A, B = split3(a), split3(b)
    mm = lambda i, j: A[i] @ B[j]        # fp32 accumulate (torch CPU: fp32 sgemm)
    x3 = mm(0, 0) + (mm(0, 1) + mm(1, 0))
    x6 = x3 + (mm(1, 1) + mm(0, 2) + mm(2, 0))
    x9 = x6 + (mm(1, 2) + mm(2, 1) + mm(2, 2))
    out.update(x3=x3, x6=x6, x9=x9)
    return {k: (float((v.double() - exact).abs().max()) / rms, float((v.double() - exact).pow(2).mean().sqrt()) / rms) for k, v in out.items()}


print("%-22s %-8s %12s %12s" % ("case", "form", "max rel", "rms rel"))
for K in (288, 576, 1152):
    for dist in ("normal", "uniform"):
        r = run(K, dist=dist)
        for k in ("fp32", "x3", "x6", "x9"):
            print("%-22s %-8s %12.3e %12.3e" % ("K=%d %s" % (K, dist), k, r[k][0], r[k][1]))
