"""Loss-side counterparts of the path's outputs, on the kernels of csrc/losses.hip.

Same constructor arguments, call signatures and result dictionaries as the reference's loss objects
  DispSmoothL1Loss            architecture/modeling/losses/smooth_l1_loss.py:9-94
  WarssersteinDistanceLoss    architecture/modeling/losses/warsserstein_distance_loss.py:9-113
so `TemporalStereo.training_step` (projects/TemporalStereo/TemporalStereo.py:130-168) can use them unchanged.  One addition:
`DispSmoothL1Loss(..., rescale=True)` takes the aggregation's disparities at their NATIVE resolutions and evaluates the wrapper's
`F.interpolate(d * full_w / dw, (full_h, full_w), bilinear, align_corners=True)` (TemporalStereo.py:305-309) inside the loss kernel
instead of materialising four full-resolution maps first.  fp32, GPU only, deterministic; differentiable in the disparities /
costs / offsets / samples.
"""
import torch

from . import _lib
from .functional import _require_gpu, _stream


def rescale_to_full(disp, full_size):
    """F.interpolate(disp * full_w / dw, size=full_size, mode='bilinear', align_corners=True) (TemporalStereo.py:305-309), forward
    only: what validation / inference report.  Training goes through DispSmoothL1Loss(rescale=True) instead."""
    _require_gpu(disp)
    disp = _lib.contiguous(disp)
    B, C, h, w = disp.shape
    H, W = full_size
    out = torch.empty((B, C, H, W), device=disp.device, dtype=torch.float32)
    _lib.check(_lib.lib().ts_resize_bilinear_fwd(_lib.ptr(disp), _lib.ptr(out), B, C, h, w, H, W, float(W) / w, out.stride(0), _stream()),
               "ts_resize_bilinear_fwd")
    return out


class _Wasserstein(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, off, sample, gt, max_disp, start_disp, sparse):
        _require_gpu(cost, off, sample, gt)
        if cost.dim() != 4 or cost.shape != off.shape or cost.shape != sample.shape:
            raise ValueError("cost, offset and disparity samples must be [B,D,H,W] of one shape")
        if gt.dim() != 4 or gt.shape[1] != 1 or gt.shape[0] != cost.shape[0]:
            raise ValueError("gtDisp must be [B,1,H,W]")
        cost, off, sample, gt = (_lib.contiguous(t) for t in (cost, off, sample, gt))
        B, D, H, W = cost.shape
        Hg, Wg = gt.shape[-2:]
        L = _lib.lib()
        loss = torch.empty(1, device=cost.device, dtype=torch.float32)
        gts = torch.empty((B, H, W), device=cost.device, dtype=torch.float32)
        ws = torch.empty(int(L.ts_wasserstein_loss_workspace_bytes(B, H, W)), device=cost.device, dtype=torch.uint8)
        _lib.check(L.ts_wasserstein_loss_fwd(_lib.ptr(cost), _lib.ptr(off), _lib.ptr(sample), _lib.ptr(gt), _lib.ptr(loss), _lib.ptr(gts),
                                             _lib.ptr(ws), B, D, H, W, Hg, Wg, float(max_disp), float(start_disp), int(bool(sparse)),
                                             _stream()), "ts_wasserstein_loss_fwd")
        ctx.save_for_backward(cost, off, sample, gts)
        ctx.meta = (B, D, H, W, Wg, float(max_disp), float(start_disp))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        cost, off, sample, gts = ctx.saved_tensors
        B, D, H, W, Wg, max_disp, start_disp = ctx.meta
        need_c = ctx.needs_input_grad[0]
        need_o = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gc = torch.empty_like(cost) if need_c else None
        go = torch.empty_like(cost) if need_o else None
        g = _lib.contiguous(g.reshape(1).float())
        _lib.check(_lib.lib().ts_wasserstein_loss_bwd(_lib.ptr(cost), _lib.ptr(off), _lib.ptr(sample), _lib.ptr(gts), _lib.ptr(g),
                                                      _lib.ptr(gc), _lib.ptr(go), B, D, H, W, Wg, max_disp, start_disp, _stream()),
                   "ts_wasserstein_loss_bwd")
        return gc, go if ctx.needs_input_grad[1] else None, go if ctx.needs_input_grad[2] else None, None, None, None, None


class _SmoothL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, est, gt, max_disp, start_disp):
        _require_gpu(est, gt)
        if est.dim() != 4 or est.shape[1] != 1 or gt.dim() != 4 or gt.shape[1] != 1 or est.shape[0] != gt.shape[0]:
            raise ValueError("estDisp and gtDisp must be [B,1,H,W]")
        est, gt = _lib.contiguous(est), _lib.contiguous(gt)
        B, _, h, w = est.shape
        Hg, Wg = gt.shape[-2:]
        L = _lib.lib()
        out = torch.empty(2, device=est.device, dtype=torch.float32)
        ws = torch.empty(int(L.ts_disp_smooth_l1_workspace_bytes(B, Hg, Wg)), device=est.device, dtype=torch.uint8)
        _lib.check(L.ts_disp_smooth_l1_fwd(_lib.ptr(est), _lib.ptr(gt), _lib.ptr(out), _lib.ptr(ws), B, h, w, Hg, Wg, float(max_disp),
                                           float(start_disp), _stream()), "ts_disp_smooth_l1_fwd")
        ctx.save_for_backward(est, gt, out)
        ctx.meta = (B, h, w, Hg, Wg, float(max_disp), float(start_disp))
        return out[0].reshape(())

    @staticmethod
    def backward(ctx, g):
        est, gt, out = ctx.saved_tensors
        B, h, w, Hg, Wg, max_disp, start_disp = ctx.meta
        ge = torch.empty_like(est)
        g = _lib.contiguous(g.reshape(1).float())
        _lib.check(_lib.lib().ts_disp_smooth_l1_bwd(_lib.ptr(est), _lib.ptr(gt), _lib.ptr(g), _lib.ptr(out), _lib.ptr(ge), B, h, w, Hg, Wg,
                                                    max_disp, start_disp, _stream()), "ts_disp_smooth_l1_bwd")
        return ge, None, None, None


def wasserstein_loss_per_level(estCost, estOffset, dispSample, gtDisp, max_disp=192, start_disp=0, sparse=False):
    """WarssersteinDistanceLoss.loss_per_level (warsserstein_distance_loss.py:52-78)."""
    return _Wasserstein.apply(estCost, estOffset, dispSample, gtDisp, max_disp, start_disp, sparse)


def smooth_l1_loss_per_level(estDisp, gtDisp, max_disp=192, start_disp=0):
    """DispSmoothL1Loss.loss_per_level (smooth_l1_loss.py:49-76) of `estDisp` rescaled to gtDisp's size when it is smaller."""
    return _SmoothL1.apply(estDisp, gtDisp, max_disp, start_disp)


class DispSmoothL1Loss(object):
    """smooth_l1_loss.py:9-94.  rescale=True: estDisp are the aggregation's native-resolution disparities (see module docstring);
    rescale=False (the reference's contract): estDisp have the ground truth's size already."""

    def __init__(self, max_disp, start_disp=0, global_weight=1.0, weights=None, sparse=False, rescale=False):
        self.max_disp, self.start_disp, self.global_weight = max_disp, start_disp, global_weight
        self.weights, self.sparse, self.rescale = weights, sparse, rescale

    @classmethod
    def from_config(cls, cfg):
        return {"max_disp": cfg.get("MAX_DISP", 192), "start_disp": cfg.get("START_DISP", 0),
                "weights": cfg.get("WEIGHTS", None), "sparse": cfg.get("SPARSE", False)}      # no global_weight: as the reference (:41-47)

    def loss_per_level(self, estDisp, gtDisp):
        if not self.rescale and estDisp.shape[-2:] != gtDisp.shape[-2:]:
            raise NotImplementedError("DispSmoothL1Loss on a disparity smaller than the ground truth pools the ground truth in the "
                                      "reference; the model wrapper never does that (it rescales first): pass rescale=True")
        return smooth_l1_loss_per_level(estDisp, gtDisp, self.max_disp, self.start_disp)

    def __call__(self, estDisp, gtDisp):
        if not isinstance(estDisp, (list, tuple)):
            estDisp = [estDisp]
        if self.weights is None:
            self.weights = [1.0] * len(estDisp)
        # weights[i] * loss * global_weight of the reference (smooth_l1_loss.py:86-92) with the two factors multiplied on the host: one
        # device multiplication per level instead of two (and one instead of two in backward); differs by at most one rounding
        return {"l1_loss_lvl{}".format(i): self.loss_per_level(d, gtDisp) * (float(self.weights[i]) * float(self.global_weight))
                for i, d in enumerate(estDisp)}

    @property
    def name(self):
        return 'SmoothL1Loss'


class WarssersteinDistanceLoss(object):
    """warsserstein_distance_loss.py:9-113."""

    def __init__(self, max_disp, start_disp=0, global_weight=1.0, weights=None, sparse=False):
        self.max_disp, self.start_disp, self.global_weight = max_disp, start_disp, global_weight
        self.weights, self.sparse = weights, sparse

    @classmethod
    def from_config(cls, cfg):
        return {"max_disp": cfg.get("MAX_DISP", 192), "start_disp": cfg.get("START_DISP", 0),
                "global_weight": cfg.get("GLOBAL_WEIGHT", 1.0), "weights": cfg.get("WEIGHTS", None),
                "sparse": cfg.get("SPARSE", False)}

    def loss_per_level(self, estCost, estOffset, dispSample, gtDisp):
        return wasserstein_loss_per_level(estCost, estOffset, dispSample, gtDisp, self.max_disp, self.start_disp, self.sparse)

    def __call__(self, estCosts, estOffsets, dispSamples, gtDisp):
        if not isinstance(estCosts, (list, tuple)):
            estCosts = [estCosts]
        if not isinstance(estOffsets, (list, tuple)):
            estOffsets = [estOffsets]
        if not isinstance(dispSamples, (list, tuple)):
            dispSamples = [dispSamples] * len(estCosts)
        assert len(estCosts) == len(estOffsets), "{}, {}".format(len(estCosts), len(estOffsets))
        if self.weights is None:
            self.weights = [1.0] * len(estCosts)
        out = {}
        for i, (c, o, s) in enumerate(zip(estCosts, estOffsets, dispSamples)):
            assert s.shape == c.shape, "sample shape: {}, cost shape: {}".format(s.shape, c.shape)
            assert o.shape == c.shape, "sample shape: {}, cost shape: {}".format(o.shape, c.shape)
            out["wars_loss_lvl{}".format(i)] = self.loss_per_level(c, o, s, gtDisp) * (float(self.weights[i]) * float(self.global_weight))
        return out

    @property
    def name(self):
        return 'WarssersteinDistanceLoss'
