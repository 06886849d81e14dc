"""Random-shape sweep of the autograd wrappers against float64 torch on the CPU (forward and every gradient).

The unit tests pin chosen shapes; this draws them (seeded): odd sizes, one-pixel maps, channel counts that are not multiples of
anything, batch 1-3.  A mismatch prints the op, the shape and the error and the run exits 1.  Found in round 5 this way (through
a new unit test, then generalised here): K1's ragged-width LDS sizing.  `python tests/fuzz_ops.py [--n 40] [--seed 0] [--ops a,b]`; tests/test_fuzz_gpu.py runs a
short sweep of every op under `-m gpu` (this file imports oracle/, so it lives under tests/)."""
import argparse
import os
import sys
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F
import temporalstereo_amd.functional as TF

dev = torch.device("cuda:0")
FAILS = []


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float64) * scale


def check(op, desc, outs, refs, grads, rgrads, tol=2e-4, max_outliers=0, grad_floor=1e-5):
    for kind, a_list, b_list in (("out", outs, refs), ("grad", grads, rgrads)):
        for i, (a, b) in enumerate(zip(a_list, b_list)):
            if a is None and b is None:
                continue
            a = a.detach().cpu().double()
            b = b.detach().double()
            if a.shape != b.shape:
                FAILS.append((op, desc, "%s %d shape %s vs %s" % (kind, i, tuple(a.shape), tuple(b.shape))))
                continue
            scale = float(b.abs().max()) + 1e-6
            diff = (a - b).abs()
            err = float(diff.max()) if a.numel() else 0.0
            floor = 1e-6 if kind == "out" else grad_floor    # (gradients that cancel to ~0: fp32 noise of the terms, not of the sum)
            if max_outliers and int((diff > tol * scale + floor).sum()) <= max_outliers and torch.isfinite(a).all():
                continue                         # an op with a sign test: an element within rounding of the kink may fall either side
            if not (err <= tol * scale + floor) or not torch.isfinite(a).all():
                FAILS.append((op, desc, "%s %d: max err %.3g at scale %.3g" % (kind, i, err, scale)))


def run(op, desc, fn_gpu, fn_ref, inputs, tol=2e-4, ref_dtype=torch.float64, grad_floor=1e-5):
    """inputs: list of float64 CPU tensors (all differentiable).  ref_dtype float32: ops whose reference has kinks that the two
    precisions can land on different sides of (a warp's tap column), so the reference runs the same fp32 sequence."""
    ref_in = [x.detach().clone().to(ref_dtype).requires_grad_() for x in inputs]
    gpu_in = [x.detach().float().to(dev).requires_grad_() for x in inputs]
    try:
        outs = fn_gpu(*gpu_in)
    except NotImplementedError:
        return
    except Exception as e:                      # an op that raises on a shape the reference accepts is a finding too
        try:
            fn_ref(*ref_in)
        except RuntimeError:
            return                               # ... both refuse it (e.g. a 5^3 pool of a 3-deep volume): the same behaviour
        FAILS.append((op, desc, "raised %s: %s" % (type(e).__name__, str(e)[:200])))
        return
    try:
        refs = fn_ref(*ref_in)
    except RuntimeError:
        return                                   # the reference itself refuses this shape (e.g. a 5^3 pool of a 3-deep volume)
    if not isinstance(outs, (tuple, list)):
        outs, refs = [outs], [refs]
    g = torch.Generator().manual_seed(1234)
    gos = [rnd(g, *r.shape) if r.dim() else torch.tensor(1.7, dtype=torch.float64) for r in refs]
    torch.autograd.backward(list(refs), [go.to(ref_dtype) for go in gos])
    try:
        torch.autograd.backward(list(outs), [go.float().to(dev) for go in gos])
    except Exception as e:
        FAILS.append((op, desc, "backward raised %s: %s" % (type(e).__name__, str(e)[:200])))
        return
    check(op, desc, outs, refs, [x.grad for x in gpu_in], [x.grad for x in ref_in], tol, grad_floor=grad_floor)


def fuzz_conv3d(r, g):
    fam = r.choice(["hw", "hw_d2", "hw_s2", "hwT", "d3", "d5", "d1", "d_s2", "dT"])
    B, Cin, Cout = r.randint(1, 3), r.randint(1, 40), r.randint(1, 40)
    D, H, W = r.randint(1, 9), r.randint(1, 37), r.randint(1, 70)
    x = rnd(g, B, Cin, D, H, W)
    sc = 1.0 / (Cin * 9) ** 0.5
    bias = rnd(g, Cout) if r.random() < 0.5 else None
    if fam in ("hw", "hw_d2", "hw_s2"):
        s, d = (2, 1) if fam == "hw_s2" else (1, 2 if fam == "hw_d2" else 1)
        w = rnd(g, Cout, Cin, 1, 3, 3, scale=sc)
        kw = dict(stride=(1, s, s), padding=(0, d, d), dilation=(1, d, d))
        f_gpu = lambda x, w, *b: TF.conv3d(x, w, b[0] if b else None, **kw)
        f_ref = lambda x, w, *b: F.conv3d(x, w, b[0] if b else None, **kw)
    elif fam == "hwT":
        w = rnd(g, Cin, Cout, 1, 3, 3, scale=sc)
        kw = dict(stride=(1, 2, 2), padding=(0, 1, 1), output_padding=(0, 1, 1))
        f_gpu = lambda x, w, *b: TF.conv_transpose3d(x, w, b[0] if b else None, **kw)
        f_ref = lambda x, w, *b: F.conv_transpose3d(x, w, b[0] if b else None, **kw)
    elif fam in ("d3", "d5", "d1"):
        k = {"d3": 3, "d5": 5, "d1": 1}[fam]
        d = r.choice([1, 2]) if k == 3 else 1
        w = rnd(g, Cout, Cin, k, 1, 1, scale=sc)
        kw = dict(stride=(1, 1, 1), padding=(d * (k - 1) // 2, 0, 0), dilation=(d, 1, 1))
        f_gpu = lambda x, w, *b: TF.conv3d(x, w, b[0] if b else None, **kw)
        f_ref = lambda x, w, *b: F.conv3d(x, w, b[0] if b else None, **kw)
    elif fam == "d_s2":
        w = rnd(g, Cout, Cin, 3, 1, 1, scale=sc)
        kw = dict(stride=(2, 1, 1), padding=(1, 0, 0), dilation=(1, 1, 1))
        f_gpu = lambda x, w, *b: TF.conv3d(x, w, b[0] if b else None, **kw)
        f_ref = lambda x, w, *b: F.conv3d(x, w, b[0] if b else None, **kw)
    else:
        w = rnd(g, Cin, Cout, 3, 1, 1, scale=sc)
        kw = dict(stride=(2, 1, 1), padding=(1, 0, 0), output_padding=(1, 0, 0))
        f_gpu = lambda x, w, *b: TF.conv_transpose3d(x, w, b[0] if b else None, **kw)
        f_ref = lambda x, w, *b: F.conv_transpose3d(x, w, b[0] if b else None, **kw)
    ins = [x, w] + ([bias] if bias is not None else [])
    run("conv3d", "%s B%d %d->%d %dx%dx%d bias=%s" % (fam, B, Cin, Cout, D, H, W, bias is not None), f_gpu, f_ref, ins)


def fuzz_deconv2d(r, g):
    B, Cin, Cout, H, W = r.randint(1, 2), r.randint(1, 64), r.randint(1, 32), r.randint(1, 40), r.randint(1, 60)
    x, w = rnd(g, B, Cin, H, W), rnd(g, Cin, Cout, 4, 4, scale=1.0 / (Cin * 4) ** 0.5)
    bias = rnd(g, Cout) if r.random() < 0.5 else None
    ins = [x, w] + ([bias] if bias is not None else [])
    run("deconv2d_k4s2", "B%d %d->%d %dx%d bias=%s" % (B, Cin, Cout, H, W, bias is not None),
        lambda x, w, *b: TF.conv_transpose2d_k4s2(x, w, b[0] if b else None),
        lambda x, w, *b: F.conv_transpose2d(x, w, b[0] if b else None, stride=2, padding=1), ins)


def fuzz_block_cost(r, g):
    import oracle
    B, C, H, W, D = r.randint(1, 2), 8 * r.randint(1, 4), r.randint(4, 40), r.randint(4, 90), r.randint(2, 9)
    L, R = rnd(g, B, C, H, W), rnd(g, B, C, H, W)
    scales = r.choice([1, 2, 3])
    if r.random() < 0.5:
        disp = torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * (W + 4.0) - 3.0
        run("block_cost", "sampled B%d C%d %dx%d D%d s%d" % (B, C, H, W, D, scales), lambda l, rr, d: TF.block_cost(l, rr, d, scales),
            lambda l, rr, d: oracle.block_cost(l, rr, d, scales), [L, R, disp], tol=3e-4, ref_dtype=torch.float32)
    else:
        D = min(D, W)
        run("block_cost", "int B%d C%d %dx%d D%d s%d" % (B, C, H, W, D, scales), lambda l, rr: TF.block_cost(l, rr, D, scales),
            lambda l, rr: oracle.block_cost(l, rr, D, scales), [L, R], tol=3e-4, ref_dtype=torch.float32)


def fuzz_dense(r, g):
    import oracle.cost_volume as oracle
    # (D, H, W >= 2: with a single plane / row / column the reference's coordinate normalisation divides by zero)
    B, C, H, W, D = r.randint(1, 2), 8 * r.randint(1, 3), r.randint(2, 30), r.randint(2, 90), r.randint(2, 9)
    L, R = rnd(g, B, C, H, W), rnd(g, B, C, H, W)
    disp = torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * (W + 4.0) - 3.0
    which = r.choice(["cat_fms", "dif_fms"])
    with torch.no_grad():
        try:
            a = getattr(TF, which)(L.float().to(dev), R.float().to(dev), disp.float().to(dev))
        except Exception as e:
            FAILS.append((which, "B%d C%d %dx%d D%d" % (B, C, H, W, D), "raised %s: %s" % (type(e).__name__, str(e)[:200])))
            return
        b = getattr(oracle, which)(L.float(), R.float(), disp.float())       # fp32 oracle: dif_fms' mask is a sign test
    # dif_fms replaces elements whose warped value is not > 0 by the tensor-wide maximum (dif_fms.py:38-44): a warped value within
    # rounding of zero is a coin toss between the two fp32 evaluations
    check(which, "B%d C%d %dx%d D%d" % (B, C, H, W, D), [a], [b], [], [], tol=1e-4, max_outliers=2 if which == "dif_fms" else 0)


def fuzz_pool_resize(r, g):
    B, C, D, H, W = r.randint(1, 2), r.randint(1, 20), r.randint(1, 12), r.randint(1, 30), r.randint(1, 50)
    x = rnd(g, B, C, D, H, W)
    # fp32 reference: two window values that round to ONE float are a tie there and not in fp64 -- the gradient then goes to the first
    # of them in scan order (max_pool3d's rule, and the kernel's)
    run("pool5_avgmax", "B%d C%d %dx%dx%d" % (B, C, D, H, W), TF.pool5_avgmax,
        lambda x: (F.avg_pool3d(x, 5, 1, 2), F.max_pool3d(x, 5, 1, 2)), [x], ref_dtype=torch.float32)
    d2, h2, w2 = r.randint(1, 12), r.randint(1, 30), r.randint(1, 50)
    a, add = rnd(g, B, C, d2, h2, w2), rnd(g, B, C, D, H, W)
    run("resize_add_silu", "B%d C%d %dx%dx%d -> %dx%dx%d" % (B, C, d2, h2, w2, D, H, W), TF.resize_add_silu,
        lambda a, add: F.silu(F.interpolate(a, size=add.shape[-3:], mode="trilinear", align_corners=True) + add), [a, add])


def fuzz_regress(r, g):
    import oracle
    B, D, H, W = r.randint(1, 2), r.randint(2, 20), r.randint(1, 30), r.randint(1, 60)
    cost = rnd(g, B, D, H, W, scale=3.0)
    ds = torch.sort(torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * 50, dim=1)[0]
    run("soft_argmin", "B%d D%d %dx%d" % (B, D, H, W), lambda c, d: TF.soft_argmin(c, d, 1.0, True),
        lambda c, d: (torch.softmax(c, dim=1) * d).sum(1, keepdim=True), [cost, ds])



def _convex_ref(logits, disp, r, scale):
    B, C, H, W = disp.shape
    w = torch.softmax(logits.view(B, 1, 9, r, r, H, W), dim=2)
    nb = F.unfold(disp * scale, kernel_size=(3, 3), padding=(1, 1)).view(B, C, 9, 1, 1, H, W)
    return torch.sum(w * nb, dim=2).permute(0, 1, 4, 2, 5, 3).contiguous().reshape(B, C, H * r, W * r)


def _unet_ref(mask, disp):
    mask = F.softmax(mask, dim=1)
    b, _, h, w = mask.shape
    dh, dw = disp.shape[-2:]
    nb = F.unfold(disp, kernel_size=(3, 3), padding=(1, 1)).reshape(b, 9, dh, dw)
    full = F.interpolate(nb * w / dw, size=(h, w), mode='bilinear', align_corners=True)
    return torch.sum(full * mask, dim=1, keepdim=True)


def fuzz_upsample(r, g):
    B, H, W, f = r.randint(1, 3), r.randint(1, 20), r.randint(1, 40), r.choice([2, 4])
    logits, disp = rnd(g, B, 9 * f * f, H, W, scale=2.0), torch.rand(B, 1, H, W, generator=g, dtype=torch.float64) * 30
    run("convex_upsample", "B%d %dx%d r%d" % (B, H, W, f), lambda l, d: TF.convex_upsample(l, d, f, float(f)),
        lambda l, d: _convex_ref(l, d, f, float(f)), [logits, disp], tol=5e-4)
    H, W = max(H, 2), max(W, 2)             # (the reference's bilinear resize to one row / column is a broadcast: covered by H, W = 2)
    mask, disp = rnd(g, B, 9, H * f, W * f, scale=2.0), torch.rand(B, 1, H, W, generator=g, dtype=torch.float64) * 40
    run("unet_upsample", "B%d %dx%d f%d" % (B, H, W, f), TF.unet_upsample, _unet_ref, [mask, disp], tol=2e-3)


def fuzz_topk(r, g):
    import oracle
    B, D, H, W = r.randint(1, 2), r.randint(2, 20), r.randint(1, 30), r.randint(1, 60)
    k = r.randint(1, min(D, 8))
    cost, samp = rnd(g, B, D, H, W, scale=2.0), torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * 48
    off = torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * 2 - 1
    cs = torch.sort(cost.float(), dim=1)[0]
    if bool((cs[:, 1:] == cs[:, :-1]).any()):
        return                                   # an exact tie of two fp32 costs: torch.topk's order among equals is unspecified
    # fp32 reference: top-k SELECTION is a comparison of fp32 costs
    run("topk_softargmax", "B%d D%d %dx%d k%d" % (B, D, H, W, k), lambda c, s_, o: TF.topk_softargmax(c, s_, o, k=k),
        lambda c, s_, o: oracle.topk_softargmax(c, s_, o, k=k), [cost, samp, off], tol=3e-4, ref_dtype=torch.float32)


def fuzz_correlation(r, g):
    from oracle import correlation as oc
    B, C, H, W = r.randint(1, 2), r.randint(1, 40), r.randint(1, 20), r.randint(1, 90)
    l, rr = rnd(g, B, C, H, W), rnd(g, B, C, H, W)

    def near_kink(out):                          # leaky_relu's corner: a sum within rounding of zero takes either slope in fp32
        return bool(((out != 0) & (out.abs() < 2e-5)).any())      # (exact zeros = outside the image: the same on both sides)
    if r.random() < 0.6:
        D = r.randint(1, 60)
        if near_kink(oc.correlation1d(l, rr, D)):
            return
        run("correlation1d", "B%d C%d %dx%d D%d" % (B, C, H, W, D), lambda a, b: TF.correlation1d(a, b, D),
            lambda a, b: oc.correlation1d(a, b, D), [l, rr], tol=3e-4)
    else:
        p = r.choice([1, 3, 5, 9])
        if near_kink(oc.correlation(l, rr, p)):
            return
        run("correlation", "B%d C%d %dx%d p%d" % (B, C, H, W, p), lambda a, b: TF.correlation(a, b, p),
            lambda a, b: oc.correlation(a, b, p), [l, rr], tol=3e-4)


def fuzz_sort_gather(r, g):
    B, C, D, H, W = r.randint(1, 2), r.randint(1, 20), r.randint(1, 16), r.randint(1, 20), r.randint(1, 40)
    vol = rnd(g, B, C, D, H, W)
    samp = torch.rand(B, D, H, W, generator=g, dtype=torch.float64) * 40
    if r.random() < 0.5:
        samp = torch.round(samp)                # ties: the sort is stable (coarse.py:103-105)

    def ref(v, s_):
        ss, order = torch.sort(s_, dim=1, stable=True)
        return torch.gather(v, 2, order.unsqueeze(1).expand_as(v)), ss
    run("sort_gather", "B%d C%d D%d %dx%d" % (B, C, D, H, W), TF.sort_gather, ref, [vol, samp], ref_dtype=torch.float32)


def fuzz_conv_bn_act(r, g):
    """conv -> BatchNorm (batch statistics) -> activation as ONE node (layers.Conv3d with norm + activation) against the same three
    torch ops in fp64, incl. the running statistics."""
    from temporalstereo_amd import layers as TL
    fam = r.choice(["hw", "d"])
    B, Cin, Cout = r.randint(1, 3), r.randint(1, 24), r.randint(1, 24)
    D, H, W = r.randint(1, 8), r.randint(1, 20), r.randint(2, 40)
    if B * D * H * W < 2:
        W += 1
    act = r.choice([None, "SiLU", "ReLU"])
    ks, pad = ((1, 3, 3), (0, 1, 1)) if fam == "hw" else ((3, 1, 1), (1, 0, 0))
    torch.manual_seed(r.randint(0, 1 << 30))
    m = TL.Conv3d(Cin, Cout, kernel_size=ks, stride=1, padding=pad, bias=False, norm=("BN3d", Cout), activation=act).to(dev).train()
    x = rnd(g, B, Cin, D, H, W)
    w = m.weight.detach().double().cpu()
    gamma, beta = rnd(g, Cout).abs() + 0.5, rnd(g, Cout)
    with torch.no_grad():
        m.norm.weight.copy_(gamma.float()); m.norm.bias.copy_(beta.float())

    def ours(x_):
        return m(x_)

    def ref(x_):
        y = F.conv3d(x_, w, None, 1, pad)
        y = F.batch_norm(y, None, None, gamma, beta, True, 0.1, m.norm.eps)
        return y if act is None else (F.silu(y) if act == "SiLU" else F.relu(y))
    if act == "ReLU":
        with torch.no_grad():
            pre = F.batch_norm(F.conv3d(x, w, None, 1, pad), None, None, gamma, beta, True, 0.1, m.norm.eps)
        if bool((pre.abs() < 2e-5).any()):
            return                               # a pre-activation within rounding of ReLU's corner
    run("conv_bn_act", "%s B%d %d->%d %dx%dx%d %s" % (fam, B, Cin, Cout, D, H, W, act), ours, ref, [x], tol=2e-3)



def fuzz_splat(r, g):
    import oracle
    import temporalstereo_amd as ts
    B, C, H, W = r.randint(1, 3), r.randint(1, 6), r.randint(1, 30), r.randint(1, 60)
    mode = r.choice(["summation", "softmax", "average", "linear"])
    x, f, m = rnd(g, B, C, H, W), rnd(g, B, 2, H, W, scale=r.choice([0.5, 2.0, 8.0])), rnd(g, B, 1, H, W)
    if mode == "linear":
        m = m.abs() + 0.1
    if mode in ("summation", "average"):
        run("softsplat", "%s B%d C%d %dx%d" % (mode, B, C, H, W), lambda a, b: ts.FunctionSoftsplat(a, b, None, mode),
            lambda a, b: oracle.softsplat(a, b, None, mode), [x, f], tol=3e-3, ref_dtype=torch.float32, grad_floor=1e-4)
    else:
        run("softsplat", "%s B%d C%d %dx%d" % (mode, B, C, H, W), lambda a, b, c: ts.FunctionSoftsplat(a, b, c, mode),
            lambda a, b, c: oracle.softsplat(a, b, c, mode), [x, f, m], tol=3e-3, ref_dtype=torch.float32, grad_floor=1e-4)


def fuzz_losses(r, g):
    from oracle import losses as ol
    from temporalstereo_amd import losses as TL
    s_ = r.choice([1, 2, 4, 8, 16])
    B, h, w, D = r.randint(1, 3), r.randint(1, 12), r.randint(1, 20), r.randint(1, 16)
    H, W = h * s_ + r.randint(0, s_ - 1), w * s_ + r.randint(0, s_ - 1)         # (the full-resolution rescale is a bilinear resize)
    md = 192.0
    sparse = r.random() < 0.4
    gt = (torch.rand(B, 1, H, W, generator=g, dtype=torch.float64) * 235.0 - 5.0).float()
    if sparse:
        gt = gt * (torch.rand(B, 1, H, W, generator=g) < 0.3)
    c, o = rnd(g, B, D, h, w, scale=3.0), rnd(g, B, D, h, w, scale=0.3)
    sm = torch.rand(B, D, h, w, generator=g, dtype=torch.float64) * md / s_
    desc = "B%d D%d %dx%d -> %dx%d sparse=%s" % (B, D, h, w, H, W, sparse)
    run("wasserstein", desc, lambda a, b, d: TL.wasserstein_loss_per_level(a, b, d, gt.to(dev), md, 0, sparse),
        lambda a, b, d: ol.wasserstein_loss_per_level(a, b, d, gt, md, 0, sparse), [c, o, sm], tol=3e-4, ref_dtype=torch.float32)
    e = torch.rand(B, 1, h, w, generator=g, dtype=torch.float64) * md / s_
    run("smooth_l1", desc, lambda a: TL.smooth_l1_loss_per_level(a, gt.to(dev), md, 0),
        lambda a: ol.smooth_l1_loss_per_level(ol.rescale_to_full(a, (H, W)), gt, md, 0), [e], tol=3e-4, ref_dtype=torch.float32)



def fuzz_glue(r, g):
    """The element-wise glue of the levels: the five range candidates behind the resized local map (fine.py:82-93), the offset head
    (module.py:384-390), argmin selection."""
    B, H, W = r.randint(1, 3), r.randint(1, 30), r.randint(1, 60)
    low = torch.rand(B, 1, H, W, generator=g, dtype=torch.float64) * 40 - 4
    high = low + torch.rand(B, 1, H, W, generator=g, dtype=torch.float64) * 16 - 2        # (high < low happens: |high - low| and min)
    nl = r.choice([0, 1, 3])
    lm = None
    if nl:
        h, w = r.randint(1, 30), r.randint(2, 60)
        lm = torch.rand(B, nl, h, w, generator=g, dtype=torch.float64) * 30

    def ref(lo, hi):
        c = [torch.abs(hi - lo) * k / 8.0 + torch.minimum(lo, hi) for k in (0, 3, 4, 5, 8)]
        out = torch.cat(c, dim=1)
        if lm is not None:
            m = F.interpolate(lm.to(lo.dtype) * W / lm.shape[-1], size=(H, W), mode="bilinear", align_corners=True)
            out = torch.cat([m, out], dim=1)
        return out
    lmg = lm.float().to(dev) if lm is not None else None
    run("candidates_in_range", "B%d %dx%d local=%d" % (B, H, W, nl), lambda lo, hi: TF.candidates_in_range(lo, hi, lmg), ref, [low, high],
        ref_dtype=torch.float32)
    x = rnd(g, B, r.randint(1, 14), H, W, scale=r.choice([1.0, 50.0, 300.0]))
    delta = r.choice([1.0, 0.5, 2.0])
    run("offset_head", "B%d %dx%d delta=%g" % (B, H, W, delta), lambda a: TF.offset_head(a, delta),
        lambda a: torch.tanh(a / 100.0).clamp(-1, 1) * delta, [x])


OPS = dict(conv3d=fuzz_conv3d, deconv2d=fuzz_deconv2d, block_cost=fuzz_block_cost, dense=fuzz_dense, pool_resize=fuzz_pool_resize,
           regress=fuzz_regress, upsample=fuzz_upsample, topk=fuzz_topk, correlation=fuzz_correlation, sort_gather=fuzz_sort_gather,
           conv_bn_act=fuzz_conv_bn_act, splat=fuzz_splat, losses=fuzz_losses, glue=fuzz_glue)

def sweep(name, n, seed):
    """-> the findings of `n` seeded cases of op family `name`."""
    r = random.Random(seed * 1000 + sum(map(ord, name)))
    g = torch.Generator().manual_seed(seed)
    del FAILS[:]
    for _ in range(n):
        OPS[name](r, g)
    return list(FAILS)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ops", default=",".join(OPS))
    a = ap.parse_args()
    bad = []
    for name in a.ops.split(","):
        try:
            found = sweep(name, a.n, a.seed)
        except Exception as e:
            found = list(FAILS) + [(name, "-", "the sweep itself raised %s: %s" % (type(e).__name__, e))]
        print("%-12s %d cases, %d findings" % (name, a.n, len(found)), flush=True)
        bad += found
    for f in bad:
        print("FINDING", *f)
    sys.exit(1 if bad else 0)
