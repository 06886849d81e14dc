"""The three pyramid levels and the registered aggregator, K1+K3+K4 wiring of SURVEY.md 3.3.

Mirrors (constructor arguments, forward signature, outputs, state-dict names)
  CoarseAggregation   architecture/modeling/aggregation/TemporalStereo/coarse.py:14-116
  FineAggregation     .../fine.py:13-132
  PreciseAggregation  .../precise.py:11-105
  TEMPORALSTEREO      .../TemporalStereo.py:14-135
with block_cost / predict_disp running on the HIP kernels (functional.block_cost,
functional.topk_softargmax) and the candidate merge using a stable sort (ties: original order --
SURVEY.md Appendix B.2; torch.sort is unstable by default on the GPU).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as TF
from ..layers import Conv3d
from ..registry import AGGREGATION_REGISTRY, configurable
from .blocks import (ConvexUpsample, DepthwiseConv3D, PredictionHeads, PyramidFusion, ResidualBlock3D, UNet)


def _reference_init(module):
    """coarse.py:52-67: conv ~ N(0, sqrt(2 / (k * Cout))), BatchNorm gamma=1 beta=0."""
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.Conv3d)):
            n = m.out_channels
            for k in m.kernel_size:
                n *= k
            m.weight.data.normal_(0, math.sqrt(2. / n))
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.bias.data.zero_()


def _init3d(cost_planes, C, norm, activation):
    return nn.Sequential(
        DepthwiseConv3D(cost_planes, C, 3, 1, 1, bias=True, norm=norm, activation=activation),
        ResidualBlock3D(in_planes=C, kernel_size=3, stride=2, padding=1, norm=norm, activation=activation),
        DepthwiseConv3D(C, C, 3, 1, padding=2, dilation=2, bias=False, norm=norm, activation=activation),
    )


def _candidates_in_range(low, high):
    """fine.py:82-87 / precise.py:73-78: five candidates at {0,3,4,5,8}/8 of the search range."""
    span, base = torch.abs(high - low), torch.min(low, high)
    # {0,3,4,5,8}/8 are exact in fp32, so span * s + base equals the reference's broadcast form; no
    # host-built tensor is involved (keeps the pass capturable into a hipGraph)
    return torch.cat([span * s + base for s in (0.0, 0.375, 0.5, 0.625, 1.0)], dim=1)


import os as _os
_SPLIT_FIRST = _os.environ.get("TS_SPLIT_FIRST_LAYER", "1") != "0"


class _Level(nn.Module):
    def weight_init(self):
        _reference_init(self)

    def _sampled_init3d(self, left, right, disp_sample):
        """init3d(block_cost(left, right, disp_sample)) of the sampled levels (fine.py:96-103, precise.py:88-91).  On the HIP path the
        first (1,3,3) layer takes the volume WITHOUT its D-fold repeat of `left` and that half's share as a per-pixel term
        (functional.first_layer_split): same function, 42 % fewer channels through the volume, the layer, and their three backward
        kernels.  TS_SPLIT_FIRST_LAYER=0: the materialised form."""
        first = self.init3d[0].conv[0]
        if _SPLIT_FIRST and left.is_cuda and left.dtype == torch.float32 and first.stride[1] == 1 and first._fusable() is not False \
                and TF.conv3d_supported(tuple(first.weight.shape), first.stride, first.padding, first.dilation, first.groups) == "hw":
            vol = TF.block_cost_warped_ag(left, right, disp_sample, self.block_cost_scale)
            y = self.init3d[0].conv[1](TF.first_layer_split(first, left, vol))
            return self.init3d[2](self.init3d[1](y))
        return self.init3d(TF.block_cost(left, right, disp_sample, self.block_cost_scale))

    def predict_disp(self, cost, disp_sample, off, k=2):
        """coarse.py:69-75 -> (disp_map, topk_disp, topk_cost), on ts_topk_softargmax_*."""
        return TF.topk_softargmax(cost, disp_sample, off, k=k)

    def _merge_memory(self, init_cost, disp_sample, prev_info, resize_to=None):
        """coarse.py:84-105 / fine.py:105-122: append the k memory candidates, sort along D and
        permute the volume accordingly."""
        memory = prev_info.get('cost_memory', None)
        hip = init_cost.is_cuda and init_cost.dtype == torch.float32
        if memory is None or not prev_info.get('use_past_cost', False):
            if hip:     # shared read-only zeros instead of two fill launches per level and frame
                mem_s = TF.const_zeros((disp_sample.shape[0], self.topk) + tuple(disp_sample.shape[2:]), disp_sample.device)
                mem_v = mem_s.unsqueeze(dim=1)
            else:
                mem_s = torch.zeros_like(disp_sample[:, :self.topk])
                mem_v = torch.zeros_like(mem_s).unsqueeze(dim=1)
        else:
            mem_s, mem_v = memory['disp_sample'], memory['cost_volume']
            if resize_to is not None:
                H, W = resize_to
                if hip and mem_s.shape == mem_v.shape:       # both maps in one launch (the value scale inside)
                    mem_s, mem_v = TF.resize_bilinear_pair(mem_s, mem_v, (H, W), float(W) / mem_s.shape[-1], 1.0)
                else:
                    mem_s = F.interpolate(mem_s * W / mem_s.shape[-1], size=(H, W), mode='bilinear', align_corners=True)
                    mem_v = F.interpolate(mem_v, size=(H, W), mode='bilinear', align_corners=True)
            mem_v = mem_v.unsqueeze(dim=1)
        mem_v = self.past_conv(mem_v)
        disp_sample = torch.cat([disp_sample, mem_s], dim=1)
        init_cost = torch.cat([init_cost, mem_v], dim=2)
        if init_cost.is_cuda and init_cost.dtype == torch.float32:
            return TF.sort_gather(init_cost, disp_sample)            # stable rank sort + gather, one kernel each way
        disp_sample, order = torch.sort(disp_sample, dim=1, stable=True)
        init_cost = torch.gather(init_cost, dim=2, index=order.unsqueeze(dim=1).expand(-1, self.C, -1, -1, -1))
        return init_cost.contiguous(), disp_sample


class CoarseAggregation(_Level):
    def __init__(self, in_planes, C, num_sample, delta=1, block_cost_scale=3, topk=2, spatial_fusion=True,
                 norm='BN3d', activation='SiLU'):
        super().__init__()
        self.in_planes, self.C, self.num_sample, self.delta = in_planes, C, num_sample, delta
        self.block_cost_scale, self.topk, self.spatial_fusion = block_cost_scale, topk, spatial_fusion
        self.norm, self.activation = norm, activation
        self.init3d = _init3d(in_planes + block_cost_scale * in_planes // 8, C, norm, activation)
        self.past_conv = Conv3d(1, C, 1, 1, 0, bias=False, norm=(norm, C), activation=activation)
        if spatial_fusion:
            self.fuse = PyramidFusion(in_planes=C, norm=norm, activation=activation)
        self.pred_heads = PredictionHeads(in_planes=C, delta=delta, norm=norm, activation=activation)
        self.convex_upsample = ConvexUpsample(in_planes=in_planes, upscale_factor=2, window_size=3)
        self.weight_init()

    def forward(self, left, right, prev_info: dict):
        B, _, H, W = left.shape
        raw_cost = TF.block_cost(left, right, int(self.num_sample), self.block_cost_scale)
        if left.is_cuda and left.dtype == torch.float32:
            disp_sample = TF.const_arange(self.num_sample, left.device)
        else:
            disp_sample = torch.arange(self.num_sample, device=left.device, dtype=left.dtype)
        disp_sample = disp_sample.view(1, -1, 1, 1).expand(B, self.num_sample, H, W)
        init_cost = self.init3d(raw_cost)
        init_cost, disp_sample = self._merge_memory(init_cost, disp_sample, prev_info, resize_to=(H, W))
        if self.spatial_fusion:
            init_cost = self.fuse(init_cost)
        final_cost, off = self.pred_heads(init_cost)
        disp, _, _ = self.predict_disp(final_cost, disp_sample, off, k=self.topk)
        disp = self.convex_upsample(left, disp)
        return disp, final_cost, off, disp_sample, prev_info


class FineAggregation(_Level):
    def __init__(self, in_planes, C, num_sample, delta=1, block_cost_scale=3, topk=2, spatial_fusion=True,
                 norm='BN3d', activation='SiLU'):
        super().__init__()
        self.in_planes, self.C, self.num_sample, self.delta = in_planes, C, num_sample, delta
        self.block_cost_scale, self.topk, self.spatial_fusion = block_cost_scale, topk, spatial_fusion
        self.norm, self.activation = norm, activation
        self.phi = nn.Parameter(torch.Tensor([0.0, ]), requires_grad=True)        # fine.py:34 (unused by forward)
        self.init3d = _init3d(2 * in_planes + block_cost_scale * in_planes // 8, C, norm, activation)
        self.past_conv = Conv3d(1, C, 1, 1, 0, bias=False, norm=(norm, C), activation=activation)
        if spatial_fusion:
            self.fuse = PyramidFusion(in_planes=C, norm=norm, activation=activation)
        self.pred_heads = PredictionHeads(in_planes=C, delta=delta, norm=norm, activation=activation)
        self.convex_upsample = ConvexUpsample(in_planes=in_planes, upscale_factor=2, window_size=3)
        self.weight_init()

    def generate_disparity_sample(self, low_disparity, high_disparity, num_sample, prev_info):
        local_map = prev_info.get('local_map', None)
        if low_disparity.is_cuda and low_disparity.dtype == torch.float32 and not (local_map is not None and local_map.requires_grad):
            use_map = local_map is not None and prev_info.get('local_map_size', 0) > 0
            return TF.candidates_in_range(low_disparity, high_disparity, local_map if use_map else None)      # one launch each way
        disp_sample = _candidates_in_range(low_disparity, high_disparity)
        if local_map is not None and prev_info.get('local_map_size', 0) > 0:      # fine.py:89-93
            H, W = low_disparity.shape[-2:]
            local_map = F.interpolate(local_map * W / local_map.shape[-1], size=(H, W), mode='bilinear',
                                      align_corners=True)
            disp_sample = torch.cat([local_map, disp_sample], dim=1)
        return disp_sample

    def forward(self, left, right, low_disparity, high_disparity, prev_info: dict):
        disp_sample = self.generate_disparity_sample(low_disparity, high_disparity, self.num_sample, prev_info)
        init_cost = self._sampled_init3d(left, right, disp_sample)
        init_cost, disp_sample = self._merge_memory(init_cost, disp_sample, prev_info)
        if self.spatial_fusion:
            init_cost = self.fuse(init_cost)
        final_cost, off = self.pred_heads(init_cost)
        disp, _, _ = self.predict_disp(final_cost, disp_sample, off, k=self.topk)
        disp = self.convex_upsample(left, disp)
        return disp, final_cost, off, disp_sample, prev_info


class PreciseAggregation(_Level):
    def __init__(self, in_planes, C, num_sample, delta=1, block_cost_scale=3, topk=2, norm='BN3d', activation='SiLU'):
        super().__init__()
        self.in_planes, self.C, self.num_sample, self.delta = in_planes, C, num_sample, delta
        self.block_cost_scale, self.topk = block_cost_scale, topk
        self.norm, self.activation = norm, activation
        self.init3d = _init3d(4 * in_planes + block_cost_scale * 2 * in_planes // 8, C, norm, activation)
        self.pred_heads = PredictionHeads(in_planes=C, delta=delta, norm=norm, activation=activation)
        self.refinement = UNet(in_planes=3, out_planes=in_planes)
        self.weight_init()

    def generate_disparity_sample(self, low_disparity, high_disparity, num_sample):
        if low_disparity.is_cuda and low_disparity.dtype == torch.float32:
            return TF.candidates_in_range(low_disparity, high_disparity)
        return _candidates_in_range(low_disparity, high_disparity)

    def forward(self, left, right, low_disparity, high_disparity, left_image, right_image, prev_info: dict):
        (spx2l, spx4l), (_, spx4r) = self.refinement.encoder(left_image, right_image)
        left, right = torch.cat([left, spx4l], dim=1), torch.cat([right, spx4r], dim=1)
        disp_sample = self.generate_disparity_sample(low_disparity, high_disparity, self.num_sample)
        init_cost = self._sampled_init3d(left, right, disp_sample)
        final_cost, off = self.pred_heads(init_cost)
        disp, memory_sample, memory_volume = self.predict_disp(final_cost, disp_sample, off, k=self.topk)
        full_disp = self.refinement.decoder(disp, left, spx2l)
        prev_info['prev_disp'] = full_disp.detach()
        if memory_sample.is_cuda and memory_sample.dtype == torch.float32:         # precise.py:100-103, both maps in one launch
            h2, w2 = memory_sample.shape[-2] // 2, memory_sample.shape[-1] // 2      # F.interpolate(scale_factor=1/2): floor
            ms, mv = TF.resize_bilinear_pair(memory_sample, memory_volume, (h2, w2), 0.5, 1.0)
            prev_info['cost_memory'] = {'disp_sample': ms, 'cost_volume': mv}
            return full_disp, disp, final_cost, off, disp_sample, prev_info
        prev_info['cost_memory'] = {                                               # precise.py:100-103
            'disp_sample': F.interpolate(memory_sample / 2, scale_factor=1 / 2, mode='bilinear', align_corners=True),
            'cost_volume': F.interpolate(memory_volume, scale_factor=1 / 2, mode='bilinear', align_corners=True),
        }
        return full_disp, disp, final_cost, off, disp_sample, prev_info


@AGGREGATION_REGISTRY.register()
class TEMPORALSTEREO(nn.Module):
    """Coarse -> fine -> precise cost aggregation; drop-in for the reference class of the same name."""

    @configurable
    def __init__(self, coarse, fine, precise, norm='BN', activation='SiLU'):
        super().__init__()
        self.norm, self.activation = norm, activation
        self.coarse, self.fine, self.precise = coarse, fine, precise

    @classmethod
    def from_config(cls, cfg):
        """TemporalStereo.py:38-78 (same keys and defaults)."""
        A = cfg.MODEL.AGGREGATION

        def common(node, in_planes, C, ns):
            return dict(in_planes=node.get('IN_PLANES', in_planes), C=node.get('C', C),
                        num_sample=node.get('NUM_SAMPLE', ns), delta=node.get('DELTA', 1),
                        block_cost_scale=node.get('BLOCK_COST_SCALE', 3), topk=node.get('TOPK', 2),
                        norm=node.get('NORM', 'BN3d'), activation=node.get('ACTIVATION', 'SiLU'))
        coarse = CoarseAggregation(spatial_fusion=A.COARSE.get('SPATIAL_FUSION', True), **common(A.COARSE, 192, 32, 12))
        fine = FineAggregation(spatial_fusion=A.FINE.get('SPATIAL_FUSION', True), **common(A.FINE, 64, 16, 5))
        precise = PreciseAggregation(**common(A.PRECISE, 48, 8, 5))
        return {'coarse': coarse, 'fine': fine, 'precise': precise,
                'norm': A.get('NORM', 'BN'), 'activation': A.get('ACTIVATION', 'SiLU')}

    def weight_init(self):
        _reference_init(self)

    def forward(self, left_feats, right_feats, left_image, right_image, prev_info: dict):
        disp_range = 4                                                             # TemporalStereo.py:103
        l4, l8, l16 = left_feats
        r4, r8, r16 = right_feats
        disps, costs, offs, samples, ranges = [], [], [], [], []

        disp, cost, off, samp, prev_info = self.coarse(l16, r16, prev_info)
        low, high = disp - disp_range, disp + disp_range
        disps.append(disp); costs.append(cost); offs.append(off); samples.append(samp)
        ranges.append({'low': low, 'high': high})

        disp, cost, off, samp, prev_info = self.fine(l8, r8, low, high, prev_info)
        low, high = disp - disp_range, disp + disp_range
        disps.append(disp); costs.append(cost); offs.append(off); samples.append(samp)
        ranges.append({'low': low, 'high': high})

        full, disp, cost, off, samp, prev_info = self.precise(l4, r4, low, high, left_image, right_image, prev_info)
        disps += [disp, full]; costs.append(cost); offs.append(off); samples.append(samp)
        return disps[::-1], costs[::-1], samples[::-1], offs[::-1], ranges[::-1], prev_info
