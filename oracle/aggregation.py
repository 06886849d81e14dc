"""Oracle (test infrastructure): the coarse -> fine -> precise aggregation pyramid on CPU.

A *functional* restatement, driven by a flat state dict whose keys are the reference's
parameter names (so weights produced by the reference modules load without renaming):

  architecture/modeling/aggregation/TemporalStereo/module.py   (blocks)
  .../TemporalStereo/coarse.py:77-116, fine.py:78-132, precise.py:69-105   (levels)
  .../TemporalStereo/TemporalStereo.py:97-135                   (wiring, +-4 px search range)
  architecture/modeling/layers/basic_layers.py:194-235,340-388 (conv -> norm -> activation)

Inference semantics (BatchNorm uses running statistics) unless `training=True`, in which case
batch statistics are used exactly like nn.BatchNorm*d.train() (running buffers not updated).
"""
import torch
import torch.nn.functional as F

from .cost_volume import block_cost
from .regress import topk_softargmax


class StateView:
    """Prefix-scoped read access to a flat {name: tensor} dict."""

    def __init__(self, sd, prefix="", training=False):
        self.sd, self.prefix, self.training = sd, prefix, training

    def sub(self, name):
        return StateView(self.sd, self.prefix + name + ".", self.training)

    def get(self, name, default=None):
        return self.sd.get(self.prefix + name, default)

    def __getitem__(self, name):
        return self.sd[self.prefix + name]


def _norm_act(sv, x, act):
    """basic_layers.py:231-234: optional BatchNorm (params under '<conv>.norm.') then activation."""
    g = sv.get("norm.weight")
    if g is not None:
        x = F.batch_norm(x, sv["norm.running_mean"], sv["norm.running_var"], g, sv["norm.bias"],
                         training=sv.training, momentum=0.0, eps=1e-5)
    if act == "SiLU":
        x = F.silu(x)
    elif act == "ReLU":
        x = F.relu(x)
    elif act is not None:
        raise ValueError(act)
    return x


def conv3d(sv, x, stride=1, padding=0, dilation=1, act="SiLU"):
    x = F.conv3d(x, sv["weight"], sv.get("bias"), stride, padding, dilation)
    return _norm_act(sv, x, act)


def deconv3d(sv, x, stride, padding, output_padding, act=None):
    x = F.conv_transpose3d(x, sv["weight"], sv.get("bias"), stride, padding, output_padding)
    return _norm_act(sv, x, act)


def conv2d(sv, x, stride=1, padding=0, act="SiLU"):
    x = F.conv2d(x, sv["weight"], sv.get("bias"), stride, padding)
    return _norm_act(sv, x, act)


def sep_conv3d(sv, x, k=3, stride=1, padding=1, dilation=1, act="SiLU"):
    """module.py:111-147 'DepthwiseConv3D' = (1,k,k) conv then (k,1,1) conv, each + BN + act."""
    x = conv3d(sv.sub("conv.0"), x, (1, stride, stride), (0, padding, padding), (1, dilation, dilation), act)
    return conv3d(sv.sub("conv.1"), x, (stride, 1, 1), (padding, 0, 0), (dilation, 1, 1), act)


def sep_deconv3d(sv, x, k=3, stride=2, padding=1, output_padding=1, act=None):
    """module.py:149-184."""
    x = deconv3d(sv.sub("conv.0"), x, (1, stride, stride), (0, padding, padding), (0, output_padding, output_padding), act)
    return deconv3d(sv.sub("conv.1"), x, (stride, 1, 1), (padding, 0, 0), (output_padding, 0, 0), act)


def hourglass3d(sv, x):
    """module.py:272-297 ResidualBlock3D.forward (kernel 3, stride 2, padding 1)."""
    out = sep_conv3d(sv.sub("conv1"), x, stride=2)
    pre = sep_conv3d(sv.sub("conv2"), out)
    out = sep_conv3d(sv.sub("conv3"), pre, stride=2)
    out = sep_conv3d(sv.sub("conv4"), out, act=None)
    out = F.silu(out)
    out = sep_deconv3d(sv.sub("conv5"), out)
    out = F.interpolate(out, size=pre.shape[-3:], mode='trilinear', align_corners=True)
    out = F.silu(out + sep_conv3d(sv.sub("shortcut5"), pre, act=None))
    out = sep_deconv3d(sv.sub("conv6"), out)
    out = F.interpolate(out, size=x.shape[-3:], mode='trilinear', align_corners=True)
    return F.silu(out + sep_conv3d(sv.sub("shortcut6"), x, act=None))


def init3d(sv, raw_cost):
    """coarse.py:36-40 (same in fine.py:40-44, precise.py:33-37)."""
    x = sep_conv3d(sv.sub("0"), raw_cost)
    x = hourglass3d(sv.sub("1"), x)
    return sep_conv3d(sv.sub("2"), x, padding=2, dilation=2)


def pyramid_fusion(sv, cost):
    """module.py:412-421."""
    cost = torch.cat([
        cost,
        conv3d(sv.sub("conv_5x5"), cost, 1, (2, 0, 0)),
        F.avg_pool3d(cost, kernel_size=5, stride=1, padding=2),
        F.max_pool3d(cost, kernel_size=5, stride=1, padding=2),
    ], dim=1)
    return sep_conv3d(sv.sub("conv_fuse"), cost, act=None)


def prediction_heads(sv, x, delta):
    """module.py:380-398: returns (cost, off) both [B,D,H,W]."""
    def head(h):
        y = conv3d(h.sub("0"), x, 1, (1, 0, 0))
        return conv3d(h.sub("1"), y, 1, (0, 1, 1), act=None)
    off = torch.tanh(head(sv.sub("off_head")) / 100).clamp(-1, 1) * delta
    cost = head(sv.sub("cost_head"))
    return cost.squeeze(1), off.squeeze(1)


def convex_upsample(sv, feat, disp, factor=2, window=3):
    """module.py:318-353."""
    B, C, H, W = disp.shape
    m = F.conv2d(feat, sv["mask.0.weight"], sv["mask.0.bias"], 1, 1)
    m = F.batch_norm(m, sv["mask.1.running_mean"], sv["mask.1.running_var"], sv["mask.1.weight"],
                     sv["mask.1.bias"], training=sv.training, momentum=0.0, eps=1e-5)
    m = F.conv2d(F.silu(m), sv["mask.3.weight"], sv["mask.3.bias"])
    m = torch.softmax(m.view(B, 1, window ** 2, factor, factor, H, W), dim=2)
    u = F.unfold(disp * factor, kernel_size=(window, window), padding=(window // 2, window // 2))
    u = u.view(B, C, window ** 2, 1, 1, H, W)
    u = torch.sum(m * u, dim=2).permute(0, 1, 4, 2, 5, 3).contiguous()
    return u.reshape(B, C, H * factor, W * factor)


def merge_memory(sv, init_cost, disp_sample, prev_info, topk, resize_to=None):
    """coarse.py:84-105 / fine.py:105-122: append the top-k memory slots and sort along D."""
    memory = prev_info.get('cost_memory', None)
    if memory is None or not prev_info.get('use_past_cost', False):
        mem_s = torch.zeros_like(disp_sample[:, :topk])
        mem_v = torch.zeros_like(mem_s).unsqueeze(1)
    else:
        mem_s, mem_v = memory['disp_sample'], memory['cost_volume']
        if resize_to is not None:                                   # coarse only (:91-96)
            H, W = resize_to
            mw = mem_s.shape[-1]
            mem_s = F.interpolate(mem_s * W / mw, size=(H, W), mode='bilinear', align_corners=True)
            mem_v = F.interpolate(mem_v, size=(H, W), mode='bilinear', align_corners=True)
        mem_v = mem_v.unsqueeze(1)
    mem_v = conv3d(sv.sub("past_conv"), mem_v, 1, 0)
    C = init_cost.shape[1]
    disp_sample = torch.cat([disp_sample, mem_s], dim=1)
    vol = torch.cat([init_cost, mem_v], dim=2)
    disp_sample, order = torch.sort(disp_sample, dim=1, stable=True)
    vol = torch.gather(vol, 2, order.unsqueeze(1).repeat(1, C, 1, 1, 1)).contiguous()
    return vol, disp_sample, order


def coarse_level(sv, left, right, prev_info, num_sample, delta=1.0, scales=3, topk=2, fusion=True, trace=None):
    """coarse.py:77-116."""
    B, _, H, W = left.shape
    raw = block_cost(left, right, int(num_sample), scales)
    ds0 = torch.linspace(0, num_sample - 1, num_sample, dtype=left.dtype).view(1, num_sample, 1, 1).expand(B, num_sample, H, W)
    init = init3d(sv.sub("init3d"), raw)
    merged, ds, order = merge_memory(sv, init, ds0, prev_info, topk, resize_to=(H, W))
    vol = pyramid_fusion(sv.sub("fuse"), merged) if fusion else merged
    cost, off = prediction_heads(sv.sub("pred_heads"), vol, delta)
    disp, _, _ = topk_softargmax(cost, ds, off, k=topk)
    up = convex_upsample(sv.sub("convex_upsample"), left, disp)
    if trace is not None:
        trace.update(coarse_raw=raw, coarse_order=order, coarse_disp_lowres=disp, coarse_init=init, coarse_ds0=ds0,
                     coarse_merged=merged, coarse_fused=vol, coarse_cost=cost, coarse_off=off, coarse_ds=ds, coarse_up=up)
    return up, cost, off, ds


def candidates_in_range(low, high):
    """fine.py:82-87 / precise.py:73-78: |high-low| * {0,3,4,5,8}/8 + min(low,high)."""
    steps = torch.tensor([0., 3., 4., 5., 8.], dtype=low.dtype)
    steps = (steps / steps.max()).view(1, 5, 1, 1)
    return torch.abs(high - low) * steps + torch.min(low, high)


def fine_level(sv, left, right, low, high, prev_info, delta=1.0, scales=3, topk=2, fusion=True, trace=None):
    """fine.py:97-132 (+ local-map candidates :89-93)."""
    H, W = left.shape[-2:]
    ds = candidates_in_range(low, high)
    lm = prev_info.get('local_map', None)
    if lm is not None and prev_info.get('local_map_size', 0) > 0:
        lm = F.interpolate(lm * W / lm.shape[-1], size=(H, W), mode='bilinear', align_corners=True)
        ds = torch.cat([lm, ds], dim=1)
    ds0 = ds
    raw = block_cost(left, right, ds0, scales)
    init = init3d(sv.sub("init3d"), raw)
    merged, ds, order = merge_memory(sv, init, ds0, prev_info, topk)
    vol = pyramid_fusion(sv.sub("fuse"), merged) if fusion else merged
    cost, off = prediction_heads(sv.sub("pred_heads"), vol, delta)
    disp, _, _ = topk_softargmax(cost, ds, off, k=topk)
    up = convex_upsample(sv.sub("convex_upsample"), left, disp)
    if trace is not None:
        trace.update(fine_raw=raw, fine_order=order, fine_disp_lowres=disp, fine_low=low, fine_high=high, fine_ds0=ds0,
                     fine_init=init, fine_merged=merged, fine_fused=vol, fine_cost=cost, fine_off=off, fine_ds=ds, fine_up=up)
    return up, cost, off, ds


def unet_encoder(sv, img):
    """module.py:459-466 (one image)."""
    x = conv2d(sv.sub("conv2.0"), img, 2, 1, "ReLU")
    s2 = conv2d(sv.sub("conv2.1"), x, 1, 1, "ReLU")
    x = conv2d(sv.sub("conv4.0"), s2, 2, 1, "ReLU")
    s4 = conv2d(sv.sub("conv4.1"), x, 1, 1, "ReLU")
    return s2, s4


def unet_decoder(sv, disp, feat, feat2x):
    """module.py:468-492: image-guided 9-tap x4 upsampling."""
    f = conv2d(sv.sub("fuse.0"), feat, 1, 1, "ReLU")
    f = conv2d(sv.sub("fuse.1"), f, 1, 1, "ReLU")
    d4 = sv.sub("deconv4")
    f = F.conv_transpose2d(f, d4["weight"], d4.get("bias"), 2, 1)
    f = _norm_act(d4, f, "ReLU")
    f = conv2d(sv.sub("concat"), torch.cat([f, feat2x], dim=1), 1, 1, "ReLU")
    mask = F.conv_transpose2d(f, sv["deconv2.weight"], sv["deconv2.bias"], 2, 1)
    mask = F.softmax(mask, dim=1)
    b, _, h, w = mask.shape
    _, _, dh, dw = disp.shape
    nb = F.unfold(disp, kernel_size=(3, 3), padding=(1, 1)).reshape(b, 9, dh, dw)
    full = F.interpolate(nb * w / dw, size=(h, w), mode='bilinear', align_corners=True)
    return torch.sum(full * mask, dim=1, keepdim=True)


def precise_level(sv, left, right, low, high, left_img, right_img, prev_info, delta=1.0, scales=3, topk=2, trace=None):
    """precise.py:81-105."""
    ref = sv.sub("refinement")
    s2l, s4l = unet_encoder(ref, left_img)
    _, s4r = unet_encoder(ref, right_img)
    left = torch.cat([left, s4l], dim=1)
    right = torch.cat([right, s4r], dim=1)
    ds = candidates_in_range(low, high)
    raw = block_cost(left, right, ds, scales)
    vol = init3d(sv.sub("init3d"), raw)
    cost, off = prediction_heads(sv.sub("pred_heads"), vol, delta)
    disp, mem_s, mem_v = topk_softargmax(cost, ds, off, k=topk)
    full = unet_decoder(ref, disp, left, s2l)
    prev_info['prev_disp'] = full.detach()
    prev_info['cost_memory'] = {
        'disp_sample': F.interpolate(mem_s / 2, scale_factor=1 / 2, mode='bilinear', align_corners=True),
        'cost_volume': F.interpolate(mem_v, scale_factor=1 / 2, mode='bilinear', align_corners=True),
    }
    if trace is not None:
        trace.update(precise_raw=raw, precise_low=low, precise_high=high, precise_left=left, precise_right=right,
                     precise_s2l=s2l, precise_init=vol, precise_cost=cost, precise_off=off, precise_ds=ds,
                     precise_disp_lowres=disp, precise_mem_s=mem_s, precise_mem_v=mem_v, precise_full=full)
    return full, disp, cost, off, ds


def aggregate(sd, left_feats, right_feats, left_image, right_image, prev_info, cfg=None,
              training=False, trace=None):
    """TEMPORALSTEREO.forward, TemporalStereo.py:97-135.

    sd: flat state dict with keys 'coarse.*', 'fine.*', 'precise.*'.
    left_feats/right_feats: [f4, f8, f16].  cfg: dict(coarse=dict(num_sample,delta,scales,topk,fusion), ...)
    Returns (disps[4], costs[3], disp_samples[3], offs[3], search_ranges[2], prev_info), fine->coarse.
    """
    cfg = cfg or {}
    cc = dict(num_sample=12, delta=1.0, scales=3, topk=2, fusion=True); cc.update(cfg.get('coarse', {}))
    fc = dict(delta=1.0, scales=3, topk=2, fusion=True); fc.update(cfg.get('fine', {}))
    pc = dict(delta=1.0, scales=3, topk=2); pc.update(cfg.get('precise', {}))
    sv = StateView(sd, "", training)
    rng = 4
    l4, l8, l16 = left_feats
    r4, r8, r16 = right_feats
    disps, costs, offs, samples, ranges = [], [], [], [], []

    d, c, o, s = coarse_level(sv.sub("coarse"), l16, r16, prev_info, trace=trace, **cc)
    low, high = d - rng, d + rng
    disps.append(d); costs.append(c); offs.append(o); samples.append(s)
    ranges.append({'low': low, 'high': high})

    d, c, o, s = fine_level(sv.sub("fine"), l8, r8, low, high, prev_info, trace=trace, **fc)
    low, high = d - rng, d + rng
    disps.append(d); costs.append(c); offs.append(o); samples.append(s)
    ranges.append({'low': low, 'high': high})

    full, d, c, o, s = precise_level(sv.sub("precise"), l4, r4, low, high, left_image, right_image,
                                     prev_info, trace=trace, **pc)
    disps += [d, full]; costs.append(c); offs.append(o); samples.append(s)
    return disps[::-1], costs[::-1], samples[::-1], offs[::-1], ranges[::-1], prev_info
