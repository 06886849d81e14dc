"""Oracle (test infrastructure): cost-volume construction, K1a/K1b of SURVEY.md section 8(a).

Restates  architecture/modeling/aggregation/utils/block_cost.py:6-83  and
          architecture/modeling/layers/inverse_warp_3d.py:4-58
with the same torch CPU primitives the reference reaches (unfold-equivalent integer shift,
5-D grid_sample, avg_pool3d, trilinear interpolate) so that rounding behaviour is the
reference's, not ours.
"""
import torch
import torch.nn.functional as F

GROUP = 8  # channels per correlation group, block_cost.py:8


def groupwise_neg_sqdiff(a, b):
    """-sum over groups of 8 channels of (a-b)^2.  block_cost.py:6-13.

    a, b: [B, C, D, H, W] -> [B, C//8, D, H, W]
    """
    B, C, D, H, W = a.shape
    if C % GROUP != 0:
        raise ValueError("channel count must be a multiple of 8")
    sq = torch.pow(a - b, 2.0)
    return -sq.view(B, C // GROUP, GROUP, D, H, W).sum(dim=2)


def warp_candidates(right, disp):
    """Per-candidate horizontal resampling of the right feature map.

    Follows inverse_warp_3d.py:4-58 as it is called from block_cost.py:56 (with -disp_sample):
    source x = x + (-disp); coordinates are normalised to [-1,1] over (D-1, H-1, W-1) and handed
    to grid_sample(trilinear, zeros padding, align_corners=True) on a D-expanded view.

    right: [B, C, H, W]; disp: [B, D, H, W]  ->  [B, C, D, H, W]
    """
    B, D, H, W = disp.shape
    C = right.shape[1]
    dt = disp.dtype
    vol = right.unsqueeze(2).expand(B, C, D, H, W)
    zs = torch.linspace(0, D - 1, D, dtype=dt).view(1, D, 1, 1).expand(B, D, H, W)
    ys = torch.linspace(0, H - 1, H, dtype=dt).view(1, 1, H, 1).expand(B, D, H, W)
    xs = torch.linspace(0, W - 1, W, dtype=dt).view(1, 1, 1, W).expand(B, D, H, W)
    xs = xs + (-disp)                                   # inverse_warp_3d.py:41
    gz = (zs / (D - 1) * 2) - 1                         # :45-47
    gy = (ys / (H - 1) * 2) - 1
    gx = (xs / (W - 1) * 2) - 1
    grid = torch.stack((gx, gy, gz), dim=4)             # :50-53 (w, h, d order)
    return F.grid_sample(vol, grid, padding_mode='zeros', align_corners=True)


def _multiscale_groups(ref5, tgt5, scales):
    """block_cost.py:66-78: per scale s pool by (1,2^s,2^s) (floor), group-correlate, resize back."""
    B, C, D, H, W = ref5.shape
    out = []
    for s in range(int(scales)):
        kh, kw = min(2 ** s, H), min(2 ** s, W)
        pr = F.avg_pool3d(ref5, kernel_size=(1, kh, kw), stride=(1, kh, kw))
        pt = F.avg_pool3d(tgt5, kernel_size=(1, kh, kw), stride=(1, kh, kw))
        g = groupwise_neg_sqdiff(pr, pt)
        g = F.interpolate(g, size=(D, H, W), mode='trilinear', align_corners=True)
        out.append(g.reshape(B, C // GROUP, D, H, W).contiguous())
    return out


def cost_volume_int(left, right, num_disp, scales=3):
    """Integer-candidate path, block_cost.py:34-45 + :66-81.

    target_d[x] = right[x-d] (0 for x<d); cost = -(left-target_d)^2 on C channels followed by
    `scales` blocks of C//8 group-correlation channels.   -> [B, C + scales*C//8, D, H, W]
    """
    B, C, H, W = left.shape
    D = int(num_disp)
    # pad-left by D-1 then take the D windows == unfold + flip in the reference (:36-41)
    padded = F.pad(right, (D - 1, 0, 0, 0))
    tgt = torch.stack([padded[..., D - 1 - d: D - 1 - d + W] for d in range(D)], dim=2)
    ref = left.reshape(B, C, 1, H, W).repeat(1, 1, D, 1, 1)
    cost = -(ref - tgt) ** 2
    return torch.cat([cost] + _multiscale_groups(ref, tgt, scales), dim=1)


def cost_volume_sampled(left, right, disp, scales=3):
    """Sampled-candidate path, block_cost.py:47-58 + :66-81.

    cost = cat[left broadcast over D, right warped by each candidate] (2C channels) followed by
    the group-correlation blocks.   -> [B, 2C + scales*C//8, D, H, W]
    """
    B, C, H, W = left.shape
    D = disp.shape[1]
    ref = left.unsqueeze(2).expand(B, C, D, H, W)
    tgt = warp_candidates(right, disp)
    return torch.cat([ref, tgt] + _multiscale_groups(ref, tgt, scales), dim=1)


def block_cost(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    """Same call signature as the reference's block_cost (block_cost.py:16)."""
    if isinstance(disp_sample, int):
        return cost_volume_int(reference_fm, target_fm, disp_sample, block_cost_scale)
    return cost_volume_sampled(reference_fm, target_fm, disp_sample, block_cost_scale)


def cat_fms(reference_fm, target_fm, disp_sample):
    """aggregation/utils/cat_fms.py:5-36: cat[left repeated over D, right warped by each candidate] -> [B,2C,D,H,W]."""
    B, C, H, W = reference_fm.shape
    D = disp_sample.shape[1]
    ref = reference_fm.unsqueeze(2).expand(B, C, D, H, W)
    return torch.cat([ref, warp_candidates(target_fm, disp_sample)], dim=1)


def dif_fms(reference_fm, target_fm, disp_sample):
    """aggregation/utils/dif_fms.py:5-44: |left - warped right|, with every element whose warped value is
    not > 0 (out of frame -- or simply non-positive) replaced by the maximum difference of the whole tensor."""
    B, C, H, W = reference_fm.shape
    D = disp_sample.shape[1]
    ref = reference_fm.unsqueeze(2).expand(B, C, D, H, W)
    tgt = warp_candidates(target_fm, disp_sample)
    dif = torch.abs(ref - tgt)
    max_dif = dif.max()                                   # :38, taken BEFORE masking
    keep = (tgt > 0).to(dif.dtype)                        # :40
    return dif * keep + (1 - keep) * torch.ones_like(dif) * max_dif
