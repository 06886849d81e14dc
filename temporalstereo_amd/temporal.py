"""Temporal state update between frames, K2c of SURVEY.md section 8(a).

Own counterpart of the `update_map` method of the reference's LightningModule
(projects/TemporalStereo/TemporalStereo.py:326-461; closures update_local_map :340-384 and
update_past_cost :386-426): the previous frame's disparity gives a rigid flow (pose + depth), the
top-k (disparity candidate, cost) memory and the local disparity map are re-projected into the
current frame and forward-splatted with softmax weighting.  Re-projection and splat run on the HIP
kernels (functional.project_to_3d, functional.FunctionSoftsplat); the remaining glue is a handful of
bilinear resizes of 1/8-resolution maps.
"""
import torch
import torch.nn.functional as F

from . import functional as TF

EXPMAX = 50          # clamp of the splat metric, projects/TemporalStereo/TemporalStereo.py:5


def _scaled_intrinsics(K, factor):
    down_K = torch.cat((K[:, 0:1, :] / factor, K[:, 1:2, :] / factor, K[:, 2:, :]), dim=1)
    return down_K, torch.inverse(down_K), down_K[:, 0, 0].view(-1, 1, 1, 1)


def _resize_disp(disp, h, w):
    return F.interpolate(disp * w / disp.shape[-1], size=(h, w), mode='bilinear', align_corners=True)


def _metric(prev_disp):
    return (prev_disp[:, :1] - prev_disp[:, :1].mean()).clamp(-EXPMAX, EXPMAX)


@torch.no_grad()
def update_past_cost(prev_disp, memory, K, T_past_to_now, baseline, full_w):
    """:386-426 -> {'disp_sample','cost_volume'} warped into the current frame (detached)."""
    ds, cv = memory['disp_sample'].detach(), memory['cost_volume'].detach()
    k, h, w = ds.shape[1:]
    down_K, down_inv_K, f = _scaled_intrinsics(K, full_w / w)
    pd = _resize_disp(prev_disp, h, w)
    flow = TF.project_to_3d(baseline * f / (pd + 1e-5), down_K, down_inv_K, T_past_to_now)['optical_flow'][:, :2]
    moved = TF.project_to_3d(baseline * f / (ds + 1e-5), down_K, down_inv_K, T_past_to_now)['triangular_depth']
    moved_ds = baseline * f / (moved + 1e-5)
    warped = TF.FunctionSoftsplat(torch.cat([moved_ds, cv], dim=1), flow.contiguous(), _metric(pd), 'softmax')
    return {'disp_sample': warped[:, :k].contiguous(), 'cost_volume': warped[:, k:].contiguous()}


@torch.no_grad()
def update_local_map(prev_disp, local_map, K, T_past_to_now, baseline, full_h, full_w, local_map_size):
    """:340-384 -> local disparity map [B, <=local_map_size, h, w] in the current frame (detached)."""
    if local_map is not None:
        h, w = local_map.shape[-2:]
    else:
        h, w = full_h // 8, full_w // 8
    down_K, down_inv_K, f = _scaled_intrinsics(K, full_w / w)
    pd = _resize_disp(prev_disp, h, w)
    if local_map is None:
        planes = pd
    else:
        planes = torch.cat([pd, local_map], dim=1)[:, :local_map_size]
    proj = TF.project_to_3d(baseline * f / (planes + 1e-5), down_K, down_inv_K, T_past_to_now)
    moved = baseline * f / (proj['triangular_depth'] + 1e-5)
    return TF.FunctionSoftsplat(moved, proj['optical_flow'][:, :2].contiguous(), _metric(pd), 'softmax')


@torch.no_grad()
def update_map(prev_info, K, T_now, inv_T_past, baseline, full_h, full_w, use_past_cost=True, local_map_size=0):
    """:326-338 + :428-461.  Mutates and returns prev_info (keys as in the reference).

    K [B,4,4] full-resolution intrinsics; T_now / inv_T_past [B,4,4]; baseline [B,1,1,1] or scalar.
    """
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
