"""Building blocks of the 3-D aggregation pyramid, K3/K5 of SURVEY.md section 8(a).

Same constructor arguments, forward contracts and state-dict names as
architecture/modeling/aggregation/TemporalStereo/module.py of the reference (cited per class), so a
reference checkpoint loads strictly.  The forward passes here are the *trainable* definition (torch
convolutions + BatchNorm with autograd); inference goes through aggregation.engine, which runs the
same graph on the fused HIP kernels of csrc/conv3d.hip.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as TF
from ..layers import Conv2d, Conv3d, ConvTranspose2d, ConvTranspose3d, _hip_conv as _on_hip


def _triple(k, s):
    """(1,k,k)-then-(k,1,1) factorisation of an isotropic 3-D argument."""
    return (1, s, s), (s, 1, 1)


class DepthwiseConv3D(nn.Module):
    """module.py:111-147.  Despite the name a *separable* pair: (1,k,k) conv then (k,1,1) conv,
    each followed by norm + activation."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding, dilation=1, bias=False,
                 norm='BN3d', activation='SiLU'):
        super().__init__()
        k, s, p, d = kernel_size, stride, padding, dilation
        self.conv = nn.Sequential(
            Conv3d(in_planes, out_planes, kernel_size=(1, k, k), stride=(1, s, s), padding=(0, p, p),
                   dilation=(1, d, d), bias=bias, norm=(norm, out_planes), activation=activation),
            Conv3d(out_planes, out_planes, kernel_size=(k, 1, 1), stride=(s, 1, 1), padding=(p, 0, 0),
                   dilation=(d, 1, 1), bias=bias, norm=(norm, out_planes), activation=activation),
        )

    def forward(self, x):
        return self.conv(x)


class DepthwiseConvTranspose3D(nn.Module):
    """module.py:149-184: transposed separable pair."""

    def __init__(self, in_planes, out_planes, kernel_size, stride, padding, output_padding, bias=False,
                 norm='BN3d', activation='SiLU'):
        super().__init__()
        k, s, p, o = kernel_size, stride, padding, output_padding
        self.conv = nn.Sequential(
            ConvTranspose3d(in_planes, out_planes, kernel_size=(1, k, k), stride=(1, s, s), padding=(0, p, p),
                            output_padding=(0, o, o), bias=bias, norm=(norm, out_planes), activation=activation),
            ConvTranspose3d(out_planes, out_planes, kernel_size=(k, 1, 1), stride=(s, 1, 1), padding=(p, 0, 0),
                            output_padding=(o, 0, 0), bias=bias, norm=(norm, out_planes), activation=activation),
        )

    def forward(self, x):
        return self.conv(x)


class ResidualBlock3D(nn.Module):
    """module.py:187-297: two-level hourglass; each up-step = transposed pair -> trilinear resize to
    the skip's (D,H,W) -> SiLU(up + shortcut(skip))."""

    def __init__(self, in_planes, kernel_size=3, stride=2, padding=1, norm='BN3d', activation='SiLU'):
        super().__init__()
        C, k, s, p = in_planes, kernel_size, stride, padding
        self.in_planes = C
        kw = dict(bias=False, norm=norm)
        self.conv1 = DepthwiseConv3D(C, 2 * C, k, s, p, activation=activation, **kw)
        self.conv2 = DepthwiseConv3D(2 * C, 2 * C, k, 1, p, activation=activation, **kw)
        self.conv3 = DepthwiseConv3D(2 * C, 2 * C, k, s, p, activation=activation, **kw)
        self.conv4 = DepthwiseConv3D(2 * C, 2 * C, k, 1, p, activation=None, **kw)
        self.conv5 = DepthwiseConvTranspose3D(2 * C, 2 * C, k, s, p, p, activation=None, **kw)
        self.conv6 = DepthwiseConvTranspose3D(2 * C, C, k, s, p, p, activation=None, **kw)
        self.shortcut5 = DepthwiseConv3D(2 * C, 2 * C, k, 1, p, activation=None, **kw)
        self.shortcut6 = DepthwiseConv3D(C, C, k, 1, p, activation=None, **kw)

    def forward(self, x):
        pre = self.conv2(self.conv1(x))
        if _on_hip(x):      # F.silu(conv4(...)) with the SiLU inside conv4's last conv -> BatchNorm node (one launch each way less)
            c4 = self.conv4.conv
            out = c4[1].forward_then(c4[0](self.conv3(pre)), "SiLU")
        else:
            out = F.silu(self.conv4(self.conv3(pre)))
        if _on_hip(x):      # GPU: resize + add + SiLU as one HIP kernel each way
            out = TF.resize_add_silu(self.conv5(out), self.shortcut5(pre))
            return TF.resize_add_silu(self.conv6(out), self.shortcut6(x))
        out = F.interpolate(self.conv5(out), size=pre.shape[-3:], mode='trilinear', align_corners=True)
        out = F.silu(out + self.shortcut5(pre))
        out = F.interpolate(self.conv6(out), size=x.shape[-3:], mode='trilinear', align_corners=True)
        return F.silu(out + self.shortcut6(x))


class ConvexUpsample(nn.Module):
    """module.py:300-353: x`upscale_factor` upsampling as a softmax-weighted combination of the
    window_size^2 neighbours, weights predicted from a feature map."""

    def __init__(self, in_planes, upscale_factor=2, window_size=3):
        super().__init__()
        self.in_planes, self.upscale_factor, self.window_size = in_planes, upscale_factor, window_size
        self.mask = nn.Sequential(
            nn.Conv2d(in_planes, 64, (3, 3), (1, 1), (1, 1), bias=True),
            nn.BatchNorm2d(64),
            nn.SiLU(inplace=True),
            nn.Conv2d(64, (window_size ** 2) * (upscale_factor ** 2), kernel_size=(1, 1), bias=True),
        )

    def forward(self, input, disp, disp_scale=None):
        B, C, H, W = disp.shape
        r, k = self.upscale_factor, self.window_size
        if k % 2 != 1:
            raise ValueError("window_size must be odd, got %d" % k)
        m0, bn, m3 = self.mask[0], self.mask[1], self.mask[3]
        # the HIP form needs both convolutions inside the kernels' channel limits (k*k*r*r <= 64 logits: the shipped r = 2, k = 3
        # has 36) and a BatchNorm with a fixed momentum; anything else takes the framework's modules
        hip_ok = _on_hip(input) and bn.momentum is not None and \
            TF.conv3d_supported(tuple(m0.weight.unsqueeze(2).shape), (1, 1, 1), (0, 1, 1), (1, 1, 1), 1) == "hw" and \
            TF.conv3d_supported(tuple(m3.weight.unsqueeze(2).shape), (1, 1, 1), (0, 0, 0), (1, 1, 1), 1) == "d"
        if hip_ok:                # conv 3x3 + BatchNorm + SiLU as one fused node, the 1x1 conv on the (k,1,1) family
            m = TF.conv_bn_act(input.unsqueeze(2), m0.weight.unsqueeze(2), m0.bias, bn, "SiLU", "hw", (1, 1, False))
            logits = TF.conv3d(m, m3.weight.unsqueeze(2), m3.bias, (1, 1, 1), (0, 0, 0), (1, 1, 1)).squeeze(2)
            if k == 3 and C == 1:             # softmax + 3x3 gather + weighted sum: one kernel each way
                return TF.convex_upsample(logits, disp, r, r if disp_scale is None else disp_scale)
        else:
            logits = self.mask(input)
        w = torch.softmax(logits.view(B, 1, k * k, r, r, H, W), dim=2)
        scale = r if disp_scale is None else disp_scale
        nb = F.unfold(disp * scale, kernel_size=(k, k), padding=(k // 2, k // 2)).view(B, C, k * k, 1, 1, H, W)
        up = torch.sum(w * nb, dim=2).permute(0, 1, 4, 2, 5, 3).contiguous()
        return up.reshape(B, C, H * r, W * r)


class PredictionHeads(nn.Module):
    """module.py:356-398: cost head and tanh-bounded offset head, each (3,1,1)+BN+act -> (1,3,3)."""

    def __init__(self, in_planes, delta=1, norm='BN3d', activation='SiLU'):
        super().__init__()
        self.in_planes, self.delta = in_planes, delta

        def head():
            return nn.Sequential(
                Conv3d(in_planes, in_planes, (3, 1, 1), 1, (1, 0, 0), bias=False, norm=(norm, in_planes),
                       activation=activation),
                Conv3d(in_planes, 1, (1, 3, 3), 1, (0, 1, 1), bias=False, norm=None, activation=None),
            )
        self.cost_head = head()
        self.off_head = head()

    def regress_offset(self, off):
        if _on_hip(off):
            return TF.offset_head(off, self.delta)
        return torch.tanh(off / 100).clamp(-1, 1) * self.delta

    def forward(self, init_cost):
        off = self.regress_offset(self.off_head(init_cost)).squeeze(dim=1)
        cost = self.cost_head(init_cost).squeeze(dim=1)
        return cost, off


class PyramidFusion(nn.Module):
    """module.py:401-421: cat[x, conv(5,1,1), avg-pool 5^3, max-pool 5^3] -> separable conv 4C->C."""

    def __init__(self, in_planes, norm='BN3d', activation='SiLU'):
        super().__init__()
        self.conv_5x5 = Conv3d(in_planes, in_planes, (5, 1, 1), 1, (2, 0, 0), bias=False,
                               norm=('BN3d', in_planes), activation=activation)
        self.conv_fuse = DepthwiseConv3D(4 * in_planes, in_planes, kernel_size=3, stride=1, padding=1,
                                         bias=False, norm=norm, activation=None)

    def forward(self, cost):
        if _on_hip(cost):
            avg, mx = TF.pool5_avgmax(cost)
        else:
            avg = F.avg_pool3d(cost, kernel_size=5, stride=1, padding=2)
            mx = F.max_pool3d(cost, kernel_size=5, stride=1, padding=2)
        return self.conv_fuse(torch.cat([cost, self.conv_5x5(cost), avg, mx], dim=1))


class UNet(nn.Module):
    """module.py:424-492: image encoder (1/2, 1/4 features) and the decoder that predicts the 9-tap
    weights of the final x4 upsampling.  The reference hard-codes ReLU here (:432)."""

    def __init__(self, in_planes=3, out_planes=48, norm='BN', activation='SiLU'):
        super().__init__()
        self.in_planes = in_planes
        C, act = 32, 'ReLU'

        def c3(i, o, s):
            return Conv2d(i, o, kernel_size=3, stride=s, padding=1, bias=False, norm=(norm, o), activation=act)
        self.conv2 = nn.Sequential(c3(in_planes, C, 2), c3(C, C, 1))
        self.conv4 = nn.Sequential(c3(C, out_planes, 2), c3(out_planes, out_planes, 1))
        self.fuse = nn.Sequential(c3(out_planes * 2, C, 1), c3(C, C, 1))
        self.deconv4 = ConvTranspose2d(C, C, kernel_size=4, stride=2, padding=1, norm=(norm, C), activation=act)
        self.concat = c3(C * 2, C, 1)
        self.deconv2 = nn.ConvTranspose2d(C, 9, kernel_size=(4, 4), stride=(2, 2), padding=(1, 1))

    def encoder(self, imL, imR):
        s2l = self.conv2(imL); s4l = self.conv4(s2l)
        s2r = self.conv2(imR); s4r = self.conv4(s2r)
        return [s2l, s4l], [s2r, s4r]

    def upsample(self, mask, disp):
        if _on_hip(mask) and mask.shape[1] == 9 and disp.shape[1] == 1:
            return TF.unet_upsample(mask, disp)
        mask = F.softmax(mask, dim=1)
        b, _, h, w = mask.shape
        dh, dw = disp.shape[-2:]
        nb = F.unfold(disp, kernel_size=(3, 3), padding=(1, 1)).reshape(b, 9, dh, dw)
        full = F.interpolate(nb * w / dw, size=(h, w), mode='bilinear', align_corners=True)
        return torch.sum(full * mask, dim=1, keepdim=True)

    def decoder(self, disp, feat, feat2x):
        feat = self.deconv4(self.fuse(feat))
        feat = self.concat(torch.cat([feat, feat2x], dim=1))
        if _on_hip(feat):      # nn.ConvTranspose2d(4, stride 2, padding 1) on the HIP kernels, forward and backward (module.py:457)
            return self.upsample(TF.conv_transpose2d_k4s2(feat, self.deconv2.weight, self.deconv2.bias), disp)
        return self.upsample(self.deconv2(feat), disp)
