"""ts_conv3d_hw_warp_fwd / ts_block_cost_sampled_corr_fwd: the first (1,3,3) layer of a sampled level with the warped half of
its input volume contracted over channels BEFORE the warp (reference precise.py:88-91, fine.py:96-103
over block_cost.py:47-81).

Arbiter: the oracle's block_cost (pinned to the reference's own output, tests/test_oracle_golden.py) followed by a float64
framework convolution over the full [left x D | warped | corr] volume -- i.e. literally what the reference computes.  The HIP path
must be as close to it as the materialised HIP path of rounds 1-3 (volume without its reference half + convolution + left addend)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(C, cout, scales, seed, dev, act):
    from temporalstereo_amd.aggregation import native as N
    g = torch.Generator().manual_seed(seed)
    cin = 2 * C + scales * (C // 8)
    w = torch.randn(cout, cin, 1, 3, 3, generator=g) * (2.0 / (9 * cout)) ** 0.5
    bn = torch.nn.BatchNorm3d(cout)
    bn.running_mean.copy_(0.1 * torch.randn(cout, generator=g)); bn.running_var.copy_(0.5 + torch.rand(cout, generator=g))
    bn.weight.data.copy_(0.5 + torch.rand(cout, generator=g)); bn.bias.data.copy_(0.1 * torch.randn(cout, generator=g))
    bn.eval()
    f0 = N.Folded(w.to(dev), None, bn.to(dev), act, False, "hw")
    return w, bn.cpu(), f0


def _reference(left, right, disp, w, bn, act, dil, scales):
    import oracle
    vol = oracle.block_cost(left.double(), right.double(), disp.double(), scales)
    y = torch.nn.functional.conv3d(vol, w.double(), None, 1, (0, dil, dil), (1, dil, dil))
    y = bn.double()(y)
    return torch.nn.functional.silu(y) if act else y


CASES = [
    # B, C, cout, D, H, W, dilation, act, disparity range
    (1, 16, 8, 5, 20, 36, 1, 1, (-3.0, 12.0)),
    (2, 32, 16, 6, 24, 40, 1, 1, (-1.5, 44.0)),       # taps leave the row on both sides
    (1, 16, 8, 3, 17, 29, 1, 0, (0.0, 6.0)),          # odd sizes: ragged tiles, scalar K1 path
    (2, 16, 8, 5, 20, 36, 2, 1, (-2.0, 9.0)),         # dilation 2
    (1, 64, 8, 5, 36, 60, 1, 1, (0.0, 20.0)),
    (1, 64, 16, 8, 20, 44, 1, 1, (0.0, 30.0)),        # fine level with local-map candidates
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_C%d_co%d_D%d_%dx%d_dil%d" % c[:7])
def test_first_layer_with_the_warped_half_contracted_before_the_warp(case):
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.aggregation import native as N
    B, C, cout, D, H, W, dil, act, (lo, hi) = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234 + C + H)
    left, right = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    disp = lo + (hi - lo) * torch.rand(B, D, H, W, generator=g)
    disp[:, 0, :2, :3] = 0.0                                  # exact integer positions
    disp[:, -1, -1, :] = float(W + 3)                         # a whole row of taps outside the image
    w, bn, f0 = _layer(C, cout, 3, 7 + cout, dev, N.ACT_SILU if act else N.ACT_NONE)
    exp = _reference(left, right, disp, w, bn, act, dil, 3)

    lg, rg, dg = left.to(dev), right.to(dev), disp.to(dev).contiguous()
    fl, rest, corr, q = N.split_sampled_first_layer(f0, 3)
    lterm = N.conv_hw(lg.unsqueeze(2), fl, 1, dil)
    # rounds 1-3: the volume without its reference half, convolved
    old = N.conv_hw(TF.block_cost_warped(lg, rg, dg, 3), rest, 1, dil, addend=lterm)
    # this round: correlation blocks + Q + gather
    cvol = N.block_cost_corr(lg, rg, dg, 3)
    np.testing.assert_allclose(cvol.cpu().numpy(), TF.block_cost(lg, rg, dg, 3)[:, 2 * C:].cpu().numpy(), rtol=0, atol=0)
    Q = N.conv_d(rg.unsqueeze(2), q, 1).squeeze(2)
    new = N.conv_hw_warp(cvol, corr, Q, dg, lterm.squeeze(2), dil)
    torch.cuda.synchronize()
    scale = float(exp.abs().max())
    e_old = float((old.cpu().double() - exp).abs().max()) / scale
    e_new = float((new.cpu().double() - exp).abs().max()) / scale
    assert e_new < 5e-6, "pre-contracted first layer: max error %.3g of the output's magnitude (materialised path %.3g)" % (e_new, e_old)
    assert e_new <= 2.0 * e_old + 2e-7, "pre-contracted form is further from the exact layer (%.3g) than the materialised one (%.3g)" % (e_new, e_old)


def test_corr_only_volume_is_the_tail_of_block_cost_at_full_size():
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.aggregation import native as N
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, C, H, W, D = 2, 128, 136, 240, 5
    left, right = torch.randn(B, C, H, W, generator=g).to(dev), torch.randn(B, C, H, W, generator=g).to(dev)
    disp = (40.0 * torch.rand(B, D, H, W, generator=g)).to(dev)
    full = TF.block_cost_warped(left, right, disp, 3)
    corr = N.block_cost_corr(left, right, disp, 3)
    assert torch.equal(full[:, C:], corr)
