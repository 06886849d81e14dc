"""Temporal state update between frames, K2c of SURVEY.md section 8(a).

Own counterpart of the `update_map` method of the reference's LightningModule
(projects/TemporalStereo/TemporalStereo.py:326-461; closures update_local_map :340-384 and
update_past_cost :386-426): the previous frame's disparity gives a rigid flow (pose + depth), the
top-k (disparity candidate, cost) memory and the local disparity map are re-projected into the
current frame and forward-splatted with softmax weighting.

The whole update is ONE native call (functional.reproject_memory -> ts_reproject_memory_fwd, three
launches): resize of the previous disparity, scaled intrinsics and their inverse, pose composition,
re-projection of every plane and the shared soft-max splat.  Issued op by op (the reference's ~60
framework calls; this module's first version used ~25) the update is host-bound at ~0.75 ms per frame
on MI355X -- half an aggregation pass; fused it is ~0.05 ms.
"""
import torch

from . import functional as TF

EXPMAX = 50          # clamp of the splat metric, projects/TemporalStereo/TemporalStereo.py:5 (baked into the kernel)


def _local_hw(local_map, full_h, full_w):
    if local_map is not None:
        return tuple(local_map.shape[-2:])
    return full_h // 8, full_w // 8                 # :343-347


@torch.no_grad()
def update_past_cost(prev_disp, memory, K, T_past_to_now, baseline, full_w, T_b=None):
    """:386-426 -> {'disp_sample','cost_volume'} warped into the current frame (detached)."""
    ds, cv = memory['disp_sample'].detach(), memory['cost_volume'].detach()
    h, w = ds.shape[-2:]
    out_d, out_c, _ = TF.reproject_memory(prev_disp, ds, cv, None, 0, K, T_past_to_now, T_b, baseline, full_w / w, h, w)
    return {'disp_sample': out_d, 'cost_volume': out_c}


@torch.no_grad()
def update_local_map(prev_disp, local_map, K, T_past_to_now, baseline, full_h, full_w, local_map_size, T_b=None):
    """:340-384 -> local disparity map [B, <=local_map_size, h, w] in the current frame (detached)."""
    h, w = _local_hw(local_map, full_h, full_w)
    n_out = 1 if local_map is None else local_map_size
    return TF.reproject_memory(prev_disp, None, None, local_map, n_out, K, T_past_to_now, T_b, baseline, full_w / w, h, w)[2]


@torch.no_grad()
def update_map(prev_info, K, T_now, inv_T_past, baseline, full_h, full_w, use_past_cost=True, local_map_size=0):
    """:326-338 + :428-461.  Mutates and returns prev_info (keys as in the reference).

    K [B,4,4] full-resolution intrinsics; T_now / inv_T_past [B,4,4]; baseline [B,1,1,1] or scalar.
    """
    T_a, T_b = prev_info.get('T_past_to_now', None), None
    if T_a is None:
        T_a, T_b = T_now, inv_T_past                # composed inside the kernel (:333-338)
    prev_disp = prev_info['prev_disp'].detach()
    memory = prev_info.get('cost_memory', None)
    move_memory = use_past_cost and memory is not None
    local_map = prev_info.get('local_map', None)
    if local_map is not None:
        local_map = local_map.detach()
    same_grid = move_memory and local_map_size > 0 and \
        tuple(memory['disp_sample'].shape[-2:]) == _local_hw(local_map, full_h, full_w)
    if same_grid:                                   # the normal case: both live on the 1/8 grid -> one call
        ds, cv = memory['disp_sample'].detach(), memory['cost_volume'].detach()
        h, w = ds.shape[-2:]
        n_out = 1 if local_map is None else local_map_size
        out_d, out_c, out_l = TF.reproject_memory(prev_disp, ds, cv, local_map, n_out, K, T_a, T_b, baseline,
                                                  full_w / w, h, w)
        memory = {'disp_sample': out_d, 'cost_volume': out_c}
        prev_info['local_map'] = out_l
    else:
        if move_memory:
            memory = update_past_cost(prev_disp, memory, K, T_a, baseline, full_w, T_b)
        if local_map_size > 0:
            prev_info['local_map'] = update_local_map(prev_disp, local_map, K, T_a, baseline, full_h, full_w,
                                                      local_map_size, T_b)
    if not use_past_cost:
        memory = None
    prev_info['cost_memory'] = memory
    prev_info['use_past_cost'] = use_past_cost
    if local_map_size > 0:
        prev_info['local_map_size'] = local_map_size
    return prev_info


# ---------------------------------------------------------------------------------------------------- backbone feature memory
class _ChannelSplice(torch.autograd.Function):
    """cat([memory, input[:, mc:]], 1) as one launch (ts_channel_splice_fwd); adjoint: memory <- g[:, :mc], input <- (0 | g[:, mc:])."""

    @staticmethod
    def forward(ctx, inp, memory):
        from . import _lib
        from . import functional as TF
        TF._require_gpu(inp, memory)
        inp, memory = inp.contiguous(), memory.contiguous()
        B, C = inp.shape[0], inp.shape[1]
        mc = memory.shape[1]
        N = inp[0, 0].numel()
        out = torch.empty_like(inp)
        _lib.check(_lib.lib().ts_channel_splice_fwd(_lib.ptr(memory), _lib.ptr(inp), _lib.ptr(out), B, C, mc, N, mc * N, C * N, C * N,
                                                    TF._stream()), "ts_channel_splice_fwd")
        ctx.mc = mc
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        from . import functional as TF
        g = g.contiguous()
        B, C = g.shape[0], g.shape[1]
        N = g[0, 0].numel()
        g_in = torch.empty_like(g)
        _lib.check(_lib.lib().ts_channel_splice_fwd(None, _lib.ptr(g), _lib.ptr(g_in), B, C, ctx.mc, N, 0, C * N, C * N, TF._stream()),
                   "ts_channel_splice_fwd")
        return g_in, g[:, :ctx.mc].contiguous()


def exchange_feature_memory(inp, memory=None, memory_percent=-1.0):
    """The memory plumbing in front of every residual block of the reference's backbone
    (architecture/modeling/backbone/TemporalStereo.py:183-197, :218 -- SURVEY.md section 8(f)-4):

        mc = int(C * memory_percent); the block sees cat([memory, input[:, mc:]], 1) -- the previous frame's first mc channels in
        place of this frame's -- and this frame's first mc channels become the next frame's memory.

    Returns (x, new_memory).  memory None (first frame): x is `inp` itself.  A memory of the wrong width raises, as the reference's
    assert does."""
    C = inp.shape[1]
    mc = int(C * memory_percent)
    new_memory = inp[:, :mc]
    if memory is None:
        return inp, new_memory
    if memory.shape[1] != mc:
        raise AssertionError("input shape: {}; memory shape: {}!".format(tuple(inp.shape), tuple(memory.shape)))
    if mc == 0:
        return inp, new_memory
    return _ChannelSplice.apply(inp, memory), new_memory
