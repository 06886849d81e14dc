"""GPU: the training-side kernels of csrc/train_ops.hip against the framework ops they replace (float64 on the CPU as arbiter).
  candidates_in_range   fine.py:82-93 / precise.py:73-78
  offset_head           module.py:384-390
  ConvTranspose2d(4, stride 2, padding 1) forward / backward (UNet.deconv4 with BatchNorm + ReLU, UNet.deconv2 with bias), module.py:453-457
  ClipRMSprop           clip_grad_norm_ + torch.optim.RMSprop (sceneflow.yaml:21-24, dist_train.py:94)"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("nl,ties", [(0, False), (3, False), (1, True)])
def test_candidates_in_range_both_ways(nl, ties):
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.aggregation.levels import _candidates_in_range
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    B, H, W = 2, 17, 29
    low = torch.randn(B, 1, H, W, device=dev) * 5 + 20
    high = low + torch.randn(B, 1, H, W, device=dev) * 3           # either order
    if ties:
        high[:, :, ::3] = low[:, :, ::3]
    lm = torch.rand(B, nl, 9, 15, device=dev) * 10 if nl else None
    g = torch.randn(B, nl + 5, H, W, device=dev)
    lo, hi = low.clone().requires_grad_(True), high.clone().requires_grad_(True)
    out = TF.candidates_in_range(lo, hi, lm)
    out.backward(g)
    lo64, hi64 = low.double().cpu().requires_grad_(True), high.double().cpu().requires_grad_(True)
    ref = _candidates_in_range(lo64, hi64)
    if nl:
        ref = torch.cat([F.interpolate(lm.double().cpu() * W / lm.shape[-1], size=(H, W), mode="bilinear", align_corners=True), ref], 1)
    ref.backward(g.double().cpu())
    assert _rel(out, ref) < 1e-6
    assert _rel(lo.grad, lo64.grad) < 1e-6 and _rel(hi.grad, hi64.grad) < 1e-6


def test_offset_head_both_ways():
    from temporalstereo_amd import functional as TF
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    x = torch.randn(2, 1, 7, 11, 13, device=dev) * 200
    g = torch.randn_like(x)
    xh = x.clone().requires_grad_(True)
    y = TF.offset_head(xh, 1.5)
    y.backward(g)
    x64 = x.double().cpu().requires_grad_(True)
    r = torch.tanh(x64 / 100).clamp(-1, 1) * 1.5
    r.backward(g.double().cpu())
    assert _rel(y, r) < 1e-6 and _rel(xh.grad, x64.grad) < 1e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,bn", [(2, 32, 32, 12, 20, True), (1, 32, 9, 16, 24, False), (2, 16, 9, 9, 14, False), (1, 32, 32, 34, 60, True)])
def test_conv_transpose2d_k4s2_training_form(B, Cin, Cout, H, W, bn):
    """Forward and all gradients of ConvTranspose2d(4, 2, 1) (+ BatchNorm(train) + ReLU) against float64 framework ops."""
    from temporalstereo_amd.layers import ConvTranspose2d
    from temporalstereo_amd import functional as TF
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = ConvTranspose2d(Cin, Cout, kernel_size=4, stride=2, padding=1, norm=("BN", Cout) if bn else None, activation="ReLU" if bn else None).to(dev).train()
    x = torch.randn(B, Cin, H, W, device=dev)
    g = torch.randn(B, Cout, 2 * H, 2 * W, device=dev)
    xh = x.clone().requires_grad_(True)
    y = m(xh) if bn else TF.conv_transpose2d_k4s2(xh, m.weight, m.bias)
    y.backward(g)
    w64, b64 = m.weight.detach().double().cpu().requires_grad_(True), m.bias.detach().double().cpu().requires_grad_(True)
    x64 = x.double().cpu().requires_grad_(True)
    r = F.conv_transpose2d(x64, w64, b64, 2, 1)
    if bn:
        gam, bet = m.norm.weight.detach().double().cpu().requires_grad_(True), m.norm.bias.detach().double().cpu().requires_grad_(True)
        r = F.relu(F.batch_norm(r, None, None, gam, bet, True, 0.0, m.norm.eps))
    r.backward(g.double().cpu())
    assert _rel(y, r) < 2e-6
    assert _rel(xh.grad, x64.grad) < 5e-6
    assert _rel(m.weight.grad, w64.grad) < 5e-6
    if bn:
        assert _rel(m.norm.weight.grad, gam.grad) < 5e-6 and _rel(m.norm.bias.grad, bet.grad) < 5e-6
    else:
        assert _rel(m.bias.grad, b64.grad) < 5e-6


def test_clip_rmsprop_matches_the_framework_optimizer():
    from temporalstereo_amd.train import ClipRMSprop
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    shapes = [(32, 352, 1, 3, 3), (32,), (16, 16, 3, 1, 1), (1,), (9, 32, 4, 4), (64, 64, 3, 3)]
    ours = [torch.randn(*s, device=dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in ours]
    opt = ClipRMSprop(ours, lr=1e-3, max_norm=0.1)
    ropt = torch.optim.RMSprop(ref, lr=1e-3)
    for step in range(4):
        for i, (p, r) in enumerate(zip(ours, ref)):
            if i == 3 and step % 2 == 0:            # a parameter that receives no gradient in some steps
                p.grad = r.grad = None
                continue
            gval = torch.randn_like(p) * (10.0 if step == 1 else 0.01)      # clipped and unclipped steps
            p.grad, r.grad = gval.clone(), gval.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_(ref, 0.1)
        ropt.step()
        opt.step()
        assert abs(float(opt.total_norm()) - float(norm_ref)) < 1e-5 * float(norm_ref)
        for p, r in zip(ours, ref):
            assert _rel(p, r) < 1e-5


def test_eval_batchnorm_folds_equal_the_unfused_wrappers():
    """Inside `with BNFolds():` an eval-mode, gradient-free conv -> BatchNorm -> activation wrapper is ONE convolution launch with
    the BatchNorm folded into its epilogue (the previous frames of a training step); it must equal the two-launch form, follow the
    running statistics when they move (refresh), and leave train-mode / gradient-carrying calls alone."""
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.layers import Conv2d, Conv3d, ConvTranspose2d, ConvTranspose3d
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    layers = [(Conv3d(16, 32, (1, 3, 3), 1, (0, 1, 1), bias=True, norm=("BN3d", 32), activation="SiLU"), (2, 16, 5, 12, 20)),
              (Conv3d(32, 32, (3, 1, 1), (2, 1, 1), (1, 0, 0), bias=False, norm=("BN3d", 32), activation=None), (1, 32, 6, 9, 12)),
              (ConvTranspose3d(16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), (0, 1, 1), bias=False, norm=("BN3d", 16), activation="SiLU"), (1, 16, 3, 8, 10)),
              (Conv2d(8, 64, 3, 1, 1, bias=False, norm=("BN", 64), activation="ReLU"), (2, 8, 16, 24)),
              (ConvTranspose2d(32, 32, kernel_size=4, stride=2, padding=1, norm=("BN", 32), activation="ReLU"), (1, 32, 10, 12))]
    folds = TF.BNFolds()
    for m, shape in layers:
        m = m.to(dev)
        with torch.no_grad():
            m.norm.running_mean.normal_(0, 0.3); m.norm.running_var.uniform_(0.5, 2.0)
            m.norm.weight.uniform_(0.5, 1.5); m.norm.bias.normal_(0, 0.2)
        m.eval()
        x = torch.randn(*shape, device=dev)
        with torch.no_grad():
            want = m(x)
            with folds:
                got = m(x)
            assert _rel(got, want) < 2e-6
            m.norm.running_mean.add_(0.5)                       # the statistics move (a training step in between) ...
            want2 = m(x)
            with folds:
                stale = m(x)
                folds.refresh()                                 # ... and one launch re-folds every registered layer
                got2 = m(x)
            assert _rel(stale, want) < 2e-6 and _rel(got2, want2) < 2e-6 and _rel(want2, want) > 1e-3
        n_before = len(folds.entries)
        xg = x.clone().requires_grad_(True)
        with folds:
            y = m(xg)                                           # gradients enabled: the ordinary autograd node
        assert y.requires_grad and len(folds.entries) == n_before
    assert len(folds.entries) == len(layers)


def test_channel_sum_both_forms():
    """A convolution's bias gradient (ts_channel_sum_fwd): one launch for small tensors (a 1024-thread workgroup per channel), two
    deterministic stages otherwise -- both against torch in double, and bit-identical from call to call."""
    from temporalstereo_amd import functional as TF
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    for shape in ((1, 2, 14, 34, 60), (2, 3, 5, 17, 33), (1, 9, 544, 960), (3, 8, 1, 1, 1), (1, 16, 7, 68, 120), (4, 2, 32768 // 4 + 1)):
        dy = torch.randn(*shape, generator=g).to(dev)
        got = TF._channel_sum(dy)
        want = dy.double().transpose(0, 1).reshape(shape[1], -1).sum(1)
        scale = float(dy.double().abs().transpose(0, 1).reshape(shape[1], -1).sum(1).max())
        assert float((got.double() - want).abs().max()) <= 2e-7 * scale, shape
        assert torch.equal(got, TF._channel_sum(dy))
