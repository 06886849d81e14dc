"""Oracle (test infrastructure): disparity regression, K4a/K4b of SURVEY.md section 8(a)."""
import torch
import torch.nn.functional as F


def topk_softargmax(cost, disp_sample, off, k=2):
    """Top-k soft-argmax with learned per-candidate offset.

    Follows architecture/modeling/aggregation/TemporalStereo/coarse.py:69-75 (identical in
    fine.py:70-76, precise.py:61-67).  cost/disp_sample/off: [B, D, H, W].
    Returns (disp [B,1,H,W], topk_disp [B,k,H,W], topk_cost [B,k,H,W]).
    """
    top_cost, idx = torch.topk(cost, k=k, dim=1)
    w = torch.softmax(top_cost, dim=1)
    top_disp = torch.gather(disp_sample + off, dim=1, index=idx)
    disp = torch.sum(w * top_disp, dim=1, keepdim=True)
    return disp, top_disp, top_cost


def soft_argmin(cost_volume, disp_sample, temperature=1.0, normalize=True):
    """Full softmax-over-D regression, architecture/modeling/prediction/soft_argmin.py:38-59."""
    if cost_volume.dim() != 4:
        raise ValueError('expected 4D input (got {}D input)'.format(cost_volume.dim()))
    c = cost_volume * temperature
    p = F.softmax(c, dim=1) if normalize else c
    if p.shape != disp_sample.shape:
        raise ValueError('disparity samples and cost volume must have the same shape')
    return torch.sum(p * disp_sample, dim=1, keepdim=True)


def argmin_select(cost_volume, disp_sample, dim=1):
    """Hard selection of the best candidate, architecture/modeling/prediction/argmin.py:35-46
    (the reference takes the MAX of the similarity volume)."""
    if cost_volume.shape != disp_sample.shape:
        raise ValueError('shape mismatch')
    _, idx = torch.max(cost_volume, dim=dim, keepdim=True)
    return torch.gather(disp_sample, dim=dim, index=idx)
