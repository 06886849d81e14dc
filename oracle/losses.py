"""Oracle (test infrastructure): the two losses that consume the path's outputs, plus the wrapper's full-resolution rescale.

Restates  architecture/modeling/losses/warsserstein_distance_loss.py:52-78   (loss_per_level)
          architecture/modeling/losses/smooth_l1_loss.py:49-76              (loss_per_level)
          projects/TemporalStereo/TemporalStereo.py:305-309                 (rescale of every disparity)
with the torch ops the reference uses.  Pinned by tests/golden/loss_*.npz, recorded from the imported reference classes
(tools/gen_golden.py: loss_cases)."""
import torch
import torch.nn.functional as F


def _scaled_gt(gt, H, W, sparse):
    scale = 1.0
    if gt.shape[-2] != H or gt.shape[-1] != W:
        scale = gt.shape[-1] / (W * 1.0)
        pool = F.adaptive_max_pool2d if sparse else F.adaptive_avg_pool2d
        gt = pool(gt / scale, (H, W))
    return gt, scale


def wasserstein_loss_per_level(cost, off, sample, gt, max_disp=192, start_disp=0, sparse=False):
    N, D, H, W = cost.shape
    prob = torch.softmax(cost, dim=1)
    g, scale = _scaled_gt(gt, H, W, sparse)
    mask = (g > start_disp) & (g < (max_disp / scale))
    if mask.sum() < 1.0:
        return (prob * torch.abs(off + sample - g) * mask.float()).sum(dim=1).mean()
    return ((prob * 1.0 + 0.25) * torch.abs(off + sample - g) * mask.float()).sum(dim=1).mean()


def rescale_to_full(disp, full_size):
    H, W = full_size
    return F.interpolate(disp * W / disp.shape[-1], size=(H, W), mode='bilinear', align_corners=True)


def smooth_l1_loss_per_level(est, gt, max_disp=192, start_disp=0, sparse=False):
    N, C, H, W = est.shape
    g, scale = _scaled_gt(gt, H, W, sparse)
    mask = (g > start_disp) & (g < (max_disp / scale))
    if mask.sum() < 1.0:
        return (torch.abs(est - g) * mask.float()).mean()
    return F.smooth_l1_loss(est[mask], g[mask], reduction='mean')
