"""Oracle (test infrastructure): forward (soft-max) splatting, K2b of SURVEY.md section 8(a).

PARITY UNPINNED BY THE REFERENCE: the reference implements this op only as cupy/CUDA kernels
(architecture/modeling/layers/softsplat.py:8-177) and asserts/raises on CPU tensors
(:252, :269-270), so it cannot be executed in the build container.  This file restates the
kernel text (:14-52 forward; the backward kernels :63-105 and :116-176 are the exact adjoints
and are obtained here through autograd on this differentiable restatement) and the python
wrapper (:334-360).  It is pinned by analytic known-answer tests (tests/test_oracle_splat.py).
"""
import torch


def splat_sum(inp, flow):
    """Bilinear forward splat by summation (kernel_Softsplat_updateOutput, softsplat.py:14-52).

    inp [B,C,H,W], flow [B,2,H,W] (x then y).  Each source pixel deposits value*w on the four
    integer neighbours of (x+fx, y+fy); deposits outside the frame are dropped.  Accumulates in
    the input dtype, in a fixed (deterministic) order.
    """
    B, C, H, W = inp.shape
    if flow.shape != (B, 2, H, W):
        raise ValueError("flow must be [B,2,H,W] matching the input")
    dt = inp.dtype
    xs = torch.arange(W, dtype=dt).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=dt).view(1, H, 1).expand(B, H, W)
    ox = xs + flow[:, 0]
    oy = ys + flow[:, 1]
    x0 = torch.floor(ox)
    y0 = torch.floor(oy)
    x1 = x0 + 1
    y1 = y0 + 1
    taps = (
        (x0, y0, (x1 - ox) * (y1 - oy)),      # north-west
        (x1, y0, (ox - x0) * (y1 - oy)),      # north-east
        (x0, y1, (x1 - ox) * (oy - y0)),      # south-west
        (x1, y1, (ox - x0) * (oy - y0)),      # south-east
    )
    out = torch.zeros(B, C, H * W, dtype=dt)
    src = inp.reshape(B, C, H * W)
    for tx, ty, w in taps:
        ok = (tx >= 0) & (tx < W) & (ty >= 0) & (ty < H)
        lin = (ty.clamp(0, H - 1) * W + tx.clamp(0, W - 1)).long().reshape(B, 1, H * W)
        contrib = src * (w * ok.to(dt)).reshape(B, 1, H * W)
        out = out.scatter_add(2, lin.expand(B, C, H * W), contrib)
    return out.reshape(B, C, H, W)


def softsplat(ten_input, ten_flow, ten_metric, mode):
    """FunctionSoftsplat, softsplat.py:334-360: summation / average / linear / softmax."""
    if ten_metric is not None and ten_metric.shape[1] != 1:
        raise ValueError("metric must have one channel")
    if mode not in ('summation', 'average', 'linear', 'softmax'):
        raise ValueError("unknown splat type " + str(mode))
    x = ten_input
    if mode == 'average':
        x = torch.cat([x, x.new_ones(x.shape[0], 1, x.shape[2], x.shape[3])], 1)
    elif mode == 'linear':
        x = torch.cat([x * ten_metric, ten_metric], 1)
    elif mode == 'softmax':
        e = ten_metric.exp()
        x = torch.cat([x * e, e], 1)
    out = splat_sum(x, ten_flow)
    if mode != 'summation':
        out = out[:, :-1] / (out[:, -1:] + 1e-22)
    return out
