"""Oracle (test infrastructure): temporal state update, K2c of SURVEY.md section 8(a).

Restates the closures of `update_map` in projects/TemporalStereo/TemporalStereo.py:326-461:
  pose composition          :333-338
  update_local_map          :340-384
  update_past_cost          :386-426
  state bookkeeping         :428-461
Pinned: tests/golden/temporal_update_*.npz are outputs of the reference's own update_map (its definition compiled
from the reference file by tools/gen_golden.py) -- with the one substitution that the CUDA-only FunctionSoftsplat is
oracle.splat.softsplat, whose arithmetic stays pinned analytically (parity unpinned by reference output, see splat.py).
"""
import torch
import torch.nn.functional as F

from .geometry import project_to_3d
from .splat import softsplat

EXPMAX = 50  # projects/TemporalStereo/TemporalStereo.py:5


def _scaled_intrinsics(K, factor):
    """:349-355 / :396-402: rows 0,1 of K divided by the downscale factor."""
    down_K = torch.cat((K[:, 0:1, :] / factor, K[:, 1:2, :] / factor, K[:, 2:, :]), dim=1)
    return down_K, torch.inverse(down_K), down_K[:, 0, 0].view(-1, 1, 1, 1)


def _metric(prev_disp):
    return (prev_disp[:, :1] - prev_disp[:, :1].mean()).clamp(-EXPMAX, EXPMAX)


def update_past_cost(prev_disp, memory, K, T_past_to_now, baseline, full_w):
    """:386-426.  memory: {'disp_sample','cost_volume'} each [B,k,h,w]."""
    ds = memory['disp_sample'].detach()
    cv = memory['cost_volume'].detach()
    k, h, w = ds.shape[1:]
    down_K, down_inv_K, f = _scaled_intrinsics(K, full_w / w)
    pd = F.interpolate(prev_disp * w / prev_disp.shape[-1], size=(h, w), mode='bilinear', align_corners=True)
    depth = baseline * f / (pd + 1e-5)
    flow = project_to_3d(depth, down_K, down_inv_K, T_past_to_now)['optical_flow'][:, :2]
    sample_depth = baseline * f / (ds + 1e-5)
    moved = project_to_3d(sample_depth, down_K, down_inv_K, T_past_to_now)['triangular_depth']
    moved_ds = baseline * f / (moved + 1e-5)
    warped = softsplat(torch.cat([moved_ds, cv], dim=1), flow, _metric(pd), 'softmax')
    return {'disp_sample': warped[:, :k].detach(), 'cost_volume': warped[:, k:].detach()}


def update_local_map(prev_disp, local_map, K, T_past_to_now, baseline, full_h, full_w, local_map_size):
    """:340-384."""
    if local_map is not None:
        h, w = local_map.shape[-2:]
    else:
        h, w = full_h // 8, full_w // 8
    down_K, down_inv_K, f = _scaled_intrinsics(K, full_w / w)
    pd = F.interpolate(prev_disp * w / prev_disp.shape[-1], size=(h, w), mode='bilinear', align_corners=True)
    depth = baseline * f / (pd + 1e-5)
    proj = project_to_3d(depth, down_K, down_inv_K, T_past_to_now)
    flow = proj['optical_flow'][:, :2]
    moved_pd = baseline * f / (proj['triangular_depth'] + 1e-5)
    warp_disp = softsplat(moved_pd, flow, _metric(pd), 'softmax')
    if local_map is None:
        out = warp_disp
    else:
        lm = torch.cat([pd, local_map], dim=1)[:, :local_map_size]
        ld = baseline * f / (lm + 1e-5)
        lp = project_to_3d(ld, down_K, down_inv_K, T_past_to_now)
        moved = baseline * f / (lp['triangular_depth'] + 1e-5)
        out = softsplat(moved, lp['optical_flow'][:, :2], _metric(pd), 'softmax')
    return out.detach()


def update_map(prev_info, K, T_now, inv_T_past, baseline, full_h, full_w,
               use_past_cost=True, local_map_size=0):
    """:326-338 + :428-461.  Mutates and returns prev_info."""
    T_past_to_now = prev_info.get('T_past_to_now', None)
    if T_past_to_now is None:
        T_past_to_now = torch.bmm(T_now, inv_T_past)
    prev_disp = prev_info['prev_disp'].detach()
    memory = prev_info.get('cost_memory', None)
    if use_past_cost and memory is not None:
        memory = update_past_cost(prev_disp, memory, K, T_past_to_now, baseline, full_w)
    elif not use_past_cost:
        memory = None
    prev_info['cost_memory'] = memory
    prev_info['use_past_cost'] = use_past_cost
    if local_map_size > 0:
        prev_info['local_map'] = update_local_map(prev_disp, prev_info.get('local_map', None), K, T_past_to_now,
                                                  baseline, full_h, full_w, local_map_size)
        prev_info['local_map_size'] = local_map_size
    return prev_info
