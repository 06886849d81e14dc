"""torch.autograd.Function wrappers over the C ABI (include/ts_hip.h).

Function-level drop-ins for the reference's hot-path ops:
  block_cost        architecture/modeling/aggregation/utils/block_cost.py:16
All ops require fp32 CUDA(HIP) tensors on an MI355X and raise otherwise: there is no CPU path.
"""
import torch

from . import _lib


def _stream():
    return _lib.current_stream_handle()


_QUERY = {}


def _q(name, *args):
    """A size query of the C ABI (pure function of its integer arguments), remembered: a training step asks ~1,300 of them."""
    key = (name,) + args
    v = _QUERY.get(key)
    if v is None:
        v = _QUERY[key] = int(getattr(_lib._real_lib(), name)(*args))
    return v


# Measurement hook (bench.py): when set, called as probe(key, launch) around the K1 C-ABI call so
# that HIP events can bracket exactly the library's launches on the launch stream.
_k1_probe = None


def _require_gpu(*tensors):
    """fp32 tensors of ONE GPU, and that GPU is the current device: the launch stream is the current device's
    current stream (`_stream`), so an op on tensors of another device would be enqueued on the wrong GPU's queue,
    unordered against the framework's work on theirs.  Callers on a non-current device wrap the call in
    `with torch.cuda.device_of(tensor):` (NativeAggregator / InferenceEngine do)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("temporalstereo_amd ops run on the GPU only (got a %s tensor); "
                               "there is deliberately no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise TypeError("temporalstereo_amd ops are fp32 (got %s)" % t.dtype)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("temporalstereo_amd op got tensors on different devices (%s and %s)" % (dev, t.device))
    if dev is not None and dev.index != torch.cuda.current_device():
        raise RuntimeError("temporalstereo_amd op on %s while the current device is cuda:%d: wrap the call in "
                           "`with torch.cuda.device_of(tensor):`" % (dev, torch.cuda.current_device()))


class _BlockCost(torch.autograd.Function):
    """K1.  forward: ts_block_cost_{int,sampled}_fwd; backward: ts_block_cost_{int,sampled}_bwd."""

    @staticmethod
    def forward(ctx, left, right, disp, num_disp, scales):
        _require_gpu(left, right, disp)
        if left.dim() != 4 or left.shape != right.shape:
            raise ValueError("reference_fm / target_fm must be [B,C,H,W] of equal shape")
        left = _lib.contiguous(left)
        right = _lib.contiguous(right)
        B, C, H, W = left.shape
        L = _lib.lib()
        sampled = disp is not None
        if sampled:
            if disp.dim() != 4 or disp.shape[0] != B or disp.shape[2:] != left.shape[2:]:
                raise ValueError("disp_sample must be [B,D,H,W] matching the feature maps")
            disp = _lib.contiguous(disp)
            D = disp.shape[1]
            ctot = 2 * C + scales * (C // 8)
        else:
            D = int(num_disp)
            ctot = C + scales * (C // 8)
        if C % 8 != 0:
            raise ValueError("channel count must be a multiple of 8 (block_cost.py:9)")
        out = torch.empty((B, ctot, D, H, W), device=left.device, dtype=torch.float32)
        ws = torch.empty(max(int(L.ts_block_cost_workspace_bytes(B, C, H, W, D, scales)), 256),
                         device=left.device, dtype=torch.uint8)
        def launch():
            if sampled:
                return L.ts_block_cost_sampled_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out),
                                                   _lib.ptr(ws), B, C, H, W, D, scales, _stream())
            return L.ts_block_cost_int_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(out), _lib.ptr(ws),
                                           B, C, H, W, D, scales, _stream())
        rc = launch() if _k1_probe is None else _k1_probe((B, C, H, W, D, sampled), launch)
        _lib.check(rc, "ts_block_cost_%s_fwd" % ("sampled" if sampled else "int"))
        ctx.save_for_backward(left, right, disp if sampled else None)
        ctx.meta = (B, C, H, W, D, scales, sampled)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        left, right, disp = ctx.saved_tensors
        B, C, H, W, D, scales, sampled = ctx.meta
        L = _lib.lib()
        grad_out = _lib.contiguous(grad_out)
        need_l, need_r, need_d = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        gl = torch.empty_like(left) if need_l else None
        gr = torch.empty_like(right) if need_r else None
        gd = torch.empty_like(disp) if (sampled and need_d) else None
        ws = torch.empty(max(int(L.ts_block_cost_bwd_workspace_bytes(B, C, H, W, D, scales)), 256),
                         device=left.device, dtype=torch.uint8)
        if sampled:
            rc = L.ts_block_cost_sampled_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(grad_out),
                                             _lib.ptr(gl), _lib.ptr(gr), _lib.ptr(gd), _lib.ptr(ws),
                                             B, C, H, W, D, scales, _stream())
        else:
            rc = L.ts_block_cost_int_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(grad_out),
                                         _lib.ptr(gl), _lib.ptr(gr), _lib.ptr(ws),
                                         B, C, H, W, D, scales, _stream())
        _lib.check(rc, "ts_block_cost_%s_bwd" % ("sampled" if sampled else "int"))
        return gl, gr, gd, None, None


def _fms_args(reference_fm, target_fm, disp_sample):
    _require_gpu(reference_fm, target_fm, disp_sample)
    if reference_fm.dim() != 4 or reference_fm.shape != target_fm.shape:
        raise ValueError("reference_fm / target_fm must be [B,C,H,W] of equal shape")
    B, C, H, W = reference_fm.shape
    if disp_sample.dim() != 4 or disp_sample.shape[0] != B or disp_sample.shape[2:] != reference_fm.shape[2:]:
        raise ValueError("disp_sample must be [B,D,H,W] matching the feature maps")
    if C % 8 != 0:
        raise ValueError("the HIP kernels work on 8-channel groups: C=%d is not a multiple of 8" % C)
    return _lib.contiguous(reference_fm), _lib.contiguous(target_fm), _lib.contiguous(disp_sample), B, C, H, W, disp_sample.shape[1]


def _fms_backward(l, r, d, g_ref, g_warp, needs):
    """Gradients of [ref repeated | right warped by every candidate] w.r.t. (left, right, candidates): that volume is the
    first 2C channels of the sampled block cost, so its backward is ts_block_cost_sampled_bwd at scales = 1 with a zero
    gradient on the correlation planes (one padded copy of the incoming gradient; these ops are not on the shipped path)."""
    B, C, H, W = l.shape
    D = d.shape[1]
    L = _lib.lib()
    gpad = torch.zeros((B, 2 * C + C // 8, D, H, W), device=l.device, dtype=torch.float32)
    if g_ref is not None:
        gpad[:, :C] = g_ref
    gpad[:, C:2 * C] = g_warp
    gl = torch.empty_like(l) if needs[0] else None
    gr = torch.empty_like(r) if needs[1] else None
    gd = torch.empty_like(d) if needs[2] else None
    ws = torch.empty(max(int(L.ts_block_cost_bwd_workspace_bytes(B, C, H, W, D, 1)), 256), device=l.device, dtype=torch.uint8)
    rc = L.ts_block_cost_sampled_bwd(_lib.ptr(l), _lib.ptr(r), _lib.ptr(d), _lib.ptr(gpad), _lib.ptr(gl), _lib.ptr(gr), _lib.ptr(gd),
                                     _lib.ptr(ws), B, C, H, W, D, 1, _stream())
    _lib.check(rc, "ts_block_cost_sampled_bwd")
    return gl, gr, gd


class _CatFms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reference_fm, target_fm, disp_sample):
        l, r, d, B, C, H, W, D = _fms_args(reference_fm, target_fm, disp_sample)
        out = torch.empty((B, 2 * C, D, H, W), device=l.device, dtype=torch.float32)
        rc = _lib.lib().ts_cat_fms_fwd(_lib.ptr(l), _lib.ptr(r), _lib.ptr(d), _lib.ptr(out), B, C, H, W, D, _stream())
        _lib.check(rc, "ts_cat_fms_fwd")
        ctx.save_for_backward(l, r, d)
        return out

    @staticmethod
    def backward(ctx, g):
        l, r, d = ctx.saved_tensors
        C = l.shape[1]
        g = _lib.contiguous(g)
        return _fms_backward(l, r, d, g[:, :C], g[:, C:], ctx.needs_input_grad)


class _DifFms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reference_fm, target_fm, disp_sample):
        l, r, d, B, C, H, W, D = _fms_args(reference_fm, target_fm, disp_sample)
        out = torch.empty((B, C, D, H, W), device=l.device, dtype=torch.float32)
        ws = torch.empty(int(_lib.lib().ts_dif_fms_workspace_bytes()), device=l.device, dtype=torch.uint8)
        rc = _lib.lib().ts_dif_fms_fwd(_lib.ptr(l), _lib.ptr(r), _lib.ptr(d), _lib.ptr(out), _lib.ptr(ws), B, C, H, W, D, _stream())
        _lib.check(rc, "ts_dif_fms_fwd")
        ctx.save_for_backward(l, r, d)
        return out

    @staticmethod
    def backward(ctx, g):
        # dif_fms.py:33-41 differentiated by hand: |e| where the warped value is > 0, the tensor-wide maximum elsewhere
        # (torch.max() hands its gradient to the maximal element, split evenly among ties); the warp itself through
        # the cost-volume backward kernel
        l, r, d = ctx.saved_tensors
        B, C, H, W = l.shape
        with torch.no_grad():
            warped = _CatFms.apply(l, r, d)[:, C:]
            e = l.unsqueeze(2) - warped
            a = e.abs()
            keep = (warped > 0).to(g.dtype)
            g_a = g * keep
            g_max = (g * (1 - keep)).sum()
            top = (a == a.max()).to(g.dtype)
            g_a = g_a + top * (g_max / top.sum())
            g_e = g_a * torch.sign(e)
            gl, gr, gd = _fms_backward(l, r, d, None, -g_e, (False,) + tuple(ctx.needs_input_grad[1:]))
            gl = g_e.sum(2) if ctx.needs_input_grad[0] else None
        return gl, gr, gd


def cat_fms(reference_fm, target_fm, disp_sample):
    """aggregation/utils/cat_fms.py:5 (same arguments): [B,2C,D,H,W] = cat[left repeated over D, right warped by every
    candidate] (the shipped configs use block_cost).  Differentiable in all three arguments."""
    return _CatFms.apply(reference_fm, target_fm, disp_sample)


def dif_fms(reference_fm, target_fm, disp_sample):
    """aggregation/utils/dif_fms.py:5 (same arguments): [B,C,D,H,W] = |left - warped right| with the elements whose warped
    value is not > 0 filled with the tensor-wide maximum difference.  Differentiable in all three arguments."""
    return _DifFms.apply(reference_fm, target_fm, disp_sample)


class _InverseWarp3d(torch.autograd.Function):
    """img [B,C,H,W] (C % 8 == 0), disp [B,D,H,W] -> [B,C,D,H,W] sampled at x + disp (ts_inverse_warp_3d_fwd).  Backward: the warped
    half of ts_block_cost_sampled_bwd at scales = 1 (that op warps by -disp: the candidates go in negated, their gradient comes
    back negated)."""

    @staticmethod
    def forward(ctx, img, disp):
        B, C, H, W = img.shape
        D = disp.shape[1]
        out = torch.empty((B, C, D, H, W), device=img.device, dtype=torch.float32)
        rc = _lib.lib().ts_inverse_warp_3d_fwd(_lib.ptr(img), _lib.ptr(disp), _lib.ptr(out), B, C, H, W, D, _stream())
        _lib.check(rc, "ts_inverse_warp_3d_fwd")
        ctx.save_for_backward(img, disp)
        return out

    @staticmethod
    def backward(ctx, g):
        img, disp = ctx.saved_tensors
        _, gi, gd = _fms_backward(img, img, -disp, None, _lib.contiguous(g), (False, ctx.needs_input_grad[0], ctx.needs_input_grad[1]))
        return gi, (-gd if gd is not None else None)


def inverse_warp_3d(img, disp, padding_mode='zeros', disp_Y=None):
    """layers/inverse_warp_3d.py:4 (same arguments): img [B,C,H,W] or [B,C,D,H,W], disp [B,D,H,W] -> img sampled at x + disp,
    [B,C,D,H,W], zeros outside the row.  Differentiable in img and disp.  `padding_mode` other than 'zeros' and a `disp_Y` are not
    what any caller in the reference asks for (block_cost.py:56, cat_fms.py:31, dif_fms.py:31) and are refused; wrong ranks raise the
    reference's own ValueError (inverse_warp_3d.py:31-33)."""
    if img.dim() not in (4, 5):
        raise ValueError('image is only allowed with 4 or 5 dimensions, but got {} dimensions!'.format(img.dim()))
    if disp.dim() != 4:
        raise ValueError("disp must be [B, D, H, W]")
    if padding_mode != 'zeros' or disp_Y is not None:
        raise RuntimeError("inverse_warp_3d: padding_mode=%r / disp_Y are not supported by the HIP kernel (TS_ERR_UNSUPPORTED): the reference "
                           "only ever calls it with 'zeros' and no disp_Y" % (padding_mode,))
    _require_gpu(img, disp)
    B, D, H, W = disp.shape
    if img.dim() == 5:
        if img.shape[2] != D:
            raise AssertionError('The disparity number should be same between image and disparity map!')
        if img.stride(2) == 0 or D == 1:
            img = img[:, :, 0]                       # an expanded view (cat_fms.py:28-31): every plane is the same image
        else:
            # plane d is warped by candidate d: B * D images with one candidate each (the kernel wants two: the row is doubled)
            C = img.shape[1]
            flat = img.permute(0, 2, 1, 3, 4).reshape(B * D, C, img.shape[3], img.shape[4])
            dd = disp.reshape(B * D, 1, H, W).expand(B * D, 2, H, W)
            return inverse_warp_3d(flat, dd)[:, :, 0].reshape(B, D, C, H, W).permute(0, 2, 1, 3, 4)
    if img.shape[0] != B or img.shape[2:] != disp.shape[2:]:
        raise ValueError("img [B,C,H,W] and disp [B,D,H,W] must agree in B, H, W")
    C = img.shape[1]
    if D == 1:
        return inverse_warp_3d(img, disp.expand(B, 2, H, W))[:, :, :1]
    pad = (-C) % 8
    x = _lib.contiguous(img.float())
    if pad:
        x = torch.cat([x, x.new_zeros(B, pad, H, W)], 1)
    out = _InverseWarp3d.apply(x, _lib.contiguous(disp.float()))
    return out[:, :C] if pad else out


class _Correlation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, ph, pw, keep):
        _require_gpu(left, right)
        if left.dim() != 4 or left.shape != right.shape:
            raise ValueError("reference_fm / target_fm must be [B,C,H,W] of equal shape")
        left, right = _lib.contiguous(left), _lib.contiguous(right)
        B, C, H, W = left.shape
        out = torch.empty((B, keep, H, W), device=left.device, dtype=torch.float32)
        _lib.check(_lib.lib().ts_correlation_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(out), B, C, H, W, ph, pw, keep, _stream()),
                   "ts_correlation_fwd")
        ctx.save_for_backward(left, right, out)
        ctx.meta = (ph, pw, keep)
        return out

    @staticmethod
    def backward(ctx, g):
        left, right, out = ctx.saved_tensors
        ph, pw, keep = ctx.meta
        B, C, H, W = left.shape
        gl = torch.empty_like(left) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(right) if ctx.needs_input_grad[1] else None
        g = _lib.contiguous(g)
        _lib.check(_lib.lib().ts_correlation_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(out), _lib.ptr(g),
                                                 _lib.ptr(gl), _lib.ptr(gr), B, C, H, W, ph, pw, keep, _stream()), "ts_correlation_bwd")
        return gl, gr, None, None, None


def _corr_args(kernel_size, stride, padding, dilation, dilation_patch):
    if (kernel_size, stride, padding, dilation, dilation_patch) != (1, 1, 0, 1, 1):
        raise NotImplementedError("native correlation: kernel_size=1, stride=1, padding=0, dilation=1, dilation_patch=1 "
                                  "(what the reference's callers pass)")


def correlation(reference_fm, target_fm, patch_size=1, kernel_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
    """aggregation/utils/correlation.py:10-29 (same arguments): [B, patch_size^2, H, W], plane ph*p+pw = correlation with the
    right pixel at (y + ph - p//2, x + pw - p//2), leaky_relu(0.1) applied.  Native replacement of the third-party sampler."""
    _corr_args(kernel_size, stride, padding, dilation, dilation_patch)
    if patch_size < 1 or patch_size % 2 != 1:
        raise ValueError("patch_size must be odd")
    return _Correlation.apply(reference_fm, target_fm, patch_size, patch_size, patch_size * patch_size)


def correlation1d(reference_fm, target_fm, max_disp=1, kernel_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
    """aggregation/utils/correlation.py:32-57 (same arguments): [B, max_disp, H, W]; plane k correlates the left pixel x with
    the right pixel x + k - (max_disp - 1), i.e. disparity max_disp - 1 - k."""
    _corr_args(kernel_size, stride, padding, dilation, dilation_patch)
    if max_disp < 1:
        raise ValueError("max_disp must be >= 1")
    return _Correlation.apply(reference_fm, target_fm, 1, 2 * max_disp - 1, max_disp)


# --------------------------------------------------------------------------------------- K3 convolutions
def _pad_last(t, n):
    if t.shape[-1] == n:
        return t.contiguous()
    out = t.new_zeros(t.shape[:-1] + (n,))
    out[..., :t.shape[-1]] = t
    return out


_CPAD = {}


def _cpad(n):
    if n not in _CPAD:
        _CPAD[n] = int(_lib.lib().ts_conv_cout_pad(n))
    return _CPAD[n]


class WeightLayouts:
    """Kernel layouts of the convolution weights of one training step, refreshed in ONE launch.

    Every convolution call needs its weight as [A][taps][pad(B)] (forward and backward-data each their own): issued one by one that
    is ~280 launches of `ts_conv_weight_layout` per T=2 step.  Inside `with layouts:` the first request for a (weight, order) pair
    runs that launch and registers the pair; `refresh()` -- called by the step after the optimizer has changed the weights -- then
    re-lays every registered pair with one `ts_conv_weight_layout_many` launch and later requests are dictionary look-ups.
    Keys are (data_ptr, shape, order): in-place updates (the optimizers') keep them valid; a weight that is REPLACED gets a new
    entry (the stale one is re-laid for nothing until `clear()`)."""

    def __init__(self, lazy=False):
        """lazy: `refresh()` launches nothing; a layout is redone by its first request after it (one small launch, as without the
        object) and shared by the later requests of the step -- for a replayed graph, where the one-launch refresh leaves the
        layouts cold in the cache by the time the convolutions read them (train.py)."""
        self.entries = {}           # key -> [weight, out, descriptor tuple, epoch]
        self.custom = {}            # key -> [weight, out, make, epoch]
        self._table = None
        self._dirty = False
        self._blocks = 1
        self.lazy = bool(lazy)
        self.epoch = 0

    def __enter__(self):
        global _LAYOUTS
        self._outer, _LAYOUTS = _LAYOUTS, self
        return self

    def __exit__(self, *exc):
        global _LAYOUTS
        _LAYOUTS = self._outer
        return False

    def clear(self):
        self.entries, self.custom, self._table, self._dirty = {}, {}, None, False

    def get(self, weight, a_dim, b_dim, flip):
        key = (weight.data_ptr(), tuple(weight.shape), a_dim, b_dim, bool(flip))
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= 4096:       # someone feeds temporaries (each kept alive below): start over rather than grow
                self.clear()
            out, desc = _layout_now(weight, a_dim, b_dim, flip)
            self.entries[key] = e = [weight, out, desc, self.epoch]
            self._dirty = True
        elif self.lazy and e[3] != self.epoch:
            A, T, nb, bpad, sa, sb, st, fl = e[2]
            _lib.check(_lib.lib().ts_conv_weight_layout(_lib.ptr(e[0]), _lib.ptr(e[1]), A, T, nb, bpad, sa, sb, st, fl, _stream()),
                       "ts_conv_weight_layout")
            e[3] = self.epoch
        return e[1]

    def get_custom(self, weight, tag, make):
        """A kernel form of `weight` that is not a plain [A][taps][pad(B)] re-layout (the 3x3 form of a 4x4 transposed
        convolution): `make(weight, out=None) -> out` fills it; kept and refreshed (one launch each) like the table entries."""
        key = (weight.data_ptr(), tuple(weight.shape), tag)
        e = self.custom.get(key)
        if e is None:
            e = self.custom[key] = [weight, make(weight, None), make, self.epoch]
        elif self.lazy and e[3] != self.epoch:
            make(e[0], e[1])
            e[3] = self.epoch
        return e[1]

    def refresh(self):
        """Re-lay every registered weight from its current values (one launch on the current stream; lazy: on next use)."""
        self.epoch += 1
        if not self.lazy:
            for e in self.custom.values():
                e[2](e[0], e[1])
        if self.lazy or not self.entries:
            return
        if self._dirty:
            import numpy as np
            rec = np.zeros(len(self.entries), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("A", "<i4"), ("T", "<i4"), ("nb", "<i4"),
                                                               ("bpad", "<i4"), ("sa", "<i8"), ("sb", "<i8"), ("st", "<i8"),
                                                               ("flip", "<i4"), ("reserved", "<i4")]))
            most = 1
            for i, (weight, out, d, _) in enumerate(self.entries.values()):
                rec[i] = (weight.data_ptr(), out.data_ptr()) + d + (0,)
                most = max(most, out.numel())
            dev = next(iter(self.entries.values()))[1].device
            self._table = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev)
            self._blocks = min(64, (most + 255) // 256)
            self._dirty = False
        _lib.check(_lib.lib().ts_conv_weight_layout_many(_lib.ptr(self._table), len(self.entries), self._blocks, _stream()),
                   "ts_conv_weight_layout_many")


_LAYOUTS = None


def _layout_now(weight, a_dim, b_dim, flip):
    weight = weight.detach()
    if not weight.is_contiguous():
        weight = weight.contiguous()
    d1 = weight.shape[1]
    T = weight[0, 0].numel()
    strides = (d1 * T, T)
    A, nb = weight.shape[a_dim], weight.shape[b_dim]
    bpad = _cpad(nb)
    out = torch.empty((A, T, bpad), device=weight.device, dtype=torch.float32)
    desc = (A, T, nb, bpad, strides[a_dim], strides[b_dim], 1, int(flip))
    _lib.check(_lib.lib().ts_conv_weight_layout(_lib.ptr(weight), _lib.ptr(out), A, T, nb, bpad, strides[a_dim], strides[b_dim], 1,
                                                int(flip), _stream()), "ts_conv_weight_layout")
    return out, desc


def _layout(weight, a_dim, b_dim, flip=False):
    """[A][taps][pad(B)] kernel layout of a conv weight [d0, d1, taps...] (ts_conv_weight_layout): A / B are which of the first two
    dimensions goes outermost / innermost.  Inside a `WeightLayouts` context the layouts are kept and refreshed together."""
    # only LEAVES (parameters) are registered: a temporary made per step -- the two halves of first_layer_split's weight -- would add a
    # fresh entry every step, each pinning its tensor, its layout and a table rebuild (+ pageable upload) until the 4096-entry clear
    if _LAYOUTS is not None and weight.is_contiguous() and weight.is_leaf:
        return _LAYOUTS.get(weight, a_dim, b_dim, flip)
    return _layout_now(weight, a_dim, b_dim, flip)[0]


def _shift_vec(bias, n):
    """[coutp] epilogue shift holding the bias (the kernels add it to the raw sum: scale stays 1)."""
    if bias is None:
        return None
    if bias.numel() == n and bias.is_contiguous():
        return bias.detach()                    # Cout is its own bucket (8, 16, 32, 64): nothing to pad, no launch
    v = torch.zeros(n, device=bias.device, dtype=torch.float32)
    v[:bias.numel()] = bias.detach()
    return v


# Train-mode (1,3,3) stride-1 layers on the bf16-split matrix form (csrc/conv3d.hip ig_conv_x6_kernel; fp32 products from six bf16
# products, error below the f32-input MFMA chain's: tests/test_conv_x6_gpu.py), forward and input gradient, where the inference
# engine would choose it too: >= 16 input channels, more than 8 output channels, a grid that fills the chip.  The split copy of the
# weights is made per call from the kernel layout (the parameters move every step).  TS_TRAIN_X6=0 for the A/B.
_TRAIN_X6 = __import__("os").environ.get("TS_TRAIN_X6", "1") != "0"
_TRAIN_X6_MIN_GRID = int(__import__("os").environ.get("TS_TRAIN_X6_MIN_GRID", "256"))


def _x6_conv(x, weight, wspec, y, B, Cin, Cout, D, H, W, dilation, shift=None, scale=None, act=0, addend=None):
    """y = act(scale * conv(x) + shift) through ts_conv3d_hw_x6_fwd when the shape qualifies; False when it does not.
    `weight`: the framework's (contiguous) parameter; wspec = (stride_ci, stride_co, stride_tap, flip) of the convolution being run
    inside it -- layout and bf16 split are ONE launch (ts_conv3d_hw_x6_weight_split_from)."""
    if not _TRAIN_X6 or Cin < 32 or not weight.is_contiguous():
        return False
    L = _lib.lib()
    if not L.ts_conv3d_hw_x6_supported(Cin, Cout, W, 1, dilation, 0):
        return False
    grid = ((H + 7) // 8) * ((W + 31) // 32) * D * B * ((Cout + 31) // 32)
    wsb = _q("ts_conv3d_hw_x6_workspace_bytes", B, Cin, Cout, D, H, W)
    if grid < _TRAIN_X6_MIN_GRID and not wsb:
        return False
    w6 = torch.empty(_q("ts_conv3d_hw_x6_weight_bytes", Cin, Cout), device=x.device, dtype=torch.uint8)
    _lib.check(L.ts_conv3d_hw_x6_weight_split_from(_lib.ptr(weight.detach()), _lib.ptr(w6), Cin, Cout, wspec[0], wspec[1], wspec[2], wspec[3],
                                                   _stream()), "ts_conv3d_hw_x6_weight_split_from")
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8) if wsb else None
    Ho, Wo = H, W
    rc = L.ts_conv3d_hw_x6_fwd(_lib.ptr(x), _lib.ptr(w6), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y), B, Cin, Cout, D, H, W, dilation,
                               int(act), 0.0, x.stride(0), x.stride(1), y.stride(0), y.stride(1), _lib.ptr(addend),
                               (Cout * Ho * Wo) if addend is not None else 0, _lib.ptr(ws), wsb, _stream())
    _lib.check(rc, "ts_conv3d_hw_x6_fwd")
    return True


def _hw_forward(x, weight, stride, dilation, transposed, bias=None, fold=None, act=0, addend=None):
    """Raw Conv3d (1,3,3) [padding == dilation] / ConvTranspose3d (1,3,3) stride 2, padding 1, output_padding 1 (+ bias); with
    `fold` = (scale, shift) the epilogue applies act(y * scale + shift) (an eval-mode BatchNorm folded in, bias included)."""
    _require_gpu(x, weight)
    x = x.contiguous()
    B, Cin, D, H, W = x.shape
    if transposed:                                   # weight [Cin, Cout, 1, 3, 3]
        Cout = weight.shape[1]
        Ho, Wo = 2 * H, 2 * W
    else:                                            # weight [Cout, Cin, 1, 3, 3]
        Cout = weight.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Cout, D, Ho, Wo), device=x.device, dtype=torch.float32)
    sc, sh = fold if fold is not None else (None, _shift_vec(bias, _cpad(Cout)))
    if addend is not None:          # [B, Cout, 1, Ho, Wo] (or [B, Cout, Ho, Wo]): added to every depth plane's raw sum
        addend = _lib.contiguous(addend)
        if addend.numel() != B * Cout * Ho * Wo or transposed:
            raise ValueError("conv addend: one [B, Cout, Ho, Wo] plane per batch item (stride-1 / stride-2 forms only)")
    # (element (ci, co, tap) of [Cout, Cin, 1, 3, 3]: ci * 9 + co * Cin * 9 + tap)
    if stride == 1 and not transposed and _x6_conv(x, weight, (9, Cin * 9, 1, 0), y, B, Cin, Cout, D, H, W, dilation, sh, sc, act, addend):
        return x, y, (B, Cin, Cout, D, H, W, stride, dilation, transposed)
    w_t = _layout(weight, 0, 1) if transposed else _layout(weight, 1, 0)          # [ci][t][co]
    L = _lib.lib()
    wsb = _q("ts_conv3d_hw_workspace_bytes", B, Cin, Cout, D, H, W, stride, int(transposed))
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8) if wsb else None
    rc = L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(w_t), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(y), B, Cin, Cout, D, H, W, stride, dilation,
                            int(transposed), int(act), 0.0, x.stride(0), x.stride(1), y.stride(0), y.stride(1),
                            _lib.ptr(addend), (Cout * Ho * Wo) if addend is not None else 0, _lib.ptr(ws), wsb, _stream())
    _lib.check(rc, "ts_conv3d_hw_fwd")
    return x, y, (B, Cin, Cout, D, H, W, stride, dilation, transposed)


def _channel_sum(dy):
    """dy [B, C, ...] -> [C]: sum over batch and pixels (a convolution's bias gradient) on ts_channel_sum_fwd."""
    dy = dy if dy.is_contiguous() else dy.contiguous()
    B, C = dy.shape[0], dy.shape[1]
    N = dy.numel() // (B * C)
    out = torch.empty(C, device=dy.device, dtype=torch.float32)
    ws = torch.empty(_q("ts_bn_workspace_bytes", B, C, N), device=dy.device, dtype=torch.uint8)
    _lib.check(_lib.lib().ts_channel_sum_fwd(_lib.ptr(dy), _lib.ptr(out), _lib.ptr(ws), B, C, N, C * N, N, _stream()), "ts_channel_sum_fwd")
    return out


class WgradDefer:
    """The ~95 wgrad_finish launches of a training step's backward as ONE (ts_conv_wgrad_finish_many, csrc/conv3d.hip).

    Inside `with defer:` the weight-gradient calls of convolutions whose weight was used ONCE in the step's forward pass leave
    their partial sums in their workspaces (kept alive here) and return a dw that is written only by `flush()` -- called by the step
    after backward(), before anything reads a gradient.  Weights used more than once (the UNet image encoder runs on both views) are
    accumulated by autograd as soon as the second gradient arrives, so they keep the immediate finish: uses are counted during the
    forward pass.  Not for steps whose gradient buckets go out during backward (dist.GradientBuckets reads .grad from hooks)."""

    def __init__(self):
        self.uses = {}              # weight.data_ptr() -> forward uses in this step
        self.keep = []              # partial-sum workspaces awaiting the flush
        self._host = [None, None]
        self._dev = None
        self._turn = 0
        self._copied = [None, None]
        self.cap = 512 * 48

    def __enter__(self):
        global _WGRAD_DEFER
        self._outer, _WGRAD_DEFER = _WGRAD_DEFER, self
        self.uses = {}
        self.discard()             # nothing of an earlier, aborted step may reach this step's finish launch
        return self

    def __exit__(self, *exc):
        global _WGRAD_DEFER
        _WGRAD_DEFER = self._outer
        # flush() has run in a completed step; after an exception between the first deferred call and the flush (an OOM-skip loop, an
        # interrupt) the descriptors still name workspaces and dw tensors that are about to be freed: drop them WITHOUT launching
        self.discard()
        return False

    def discard(self):
        """Drops every pending deferred finish (the library's descriptor list and the workspaces kept for it) without launching."""
        import ctypes
        L = _lib.lib()
        if int(L.ts_conv_wgrad_pending()) != 0:
            scratch = (ctypes.c_char * self.cap)()
            n, blocks = ctypes.c_int(0), ctypes.c_int(0)
            L.ts_conv_wgrad_take(ctypes.cast(scratch, ctypes.c_void_p), self.cap, ctypes.byref(n), ctypes.byref(blocks))
        self.keep = []

    def used(self, weight):
        k = weight.data_ptr()
        self.uses[k] = self.uses.get(k, 0) + 1

    def single_use(self, weight):
        # a deferred dw is handed to autograd UNWRITTEN and filled by flush(): only sound when AccumulateGrad takes the tensor itself
        # (no gradient accumulated yet, no hook that would read or clone it first)
        # (a non-leaf -- a slice made inside the step -- hands its gradient to another backward node, which reads it at once)
        return (self.uses.get(weight.data_ptr(), 0) == 1 and weight.is_leaf and weight.grad is None
                and not getattr(weight, "_backward_hooks", None) and not getattr(weight, "_post_accumulate_grad_hooks", None))

    def flush(self):
        import ctypes
        L = _lib.lib()
        if int(L.ts_conv_wgrad_pending()) == 0:
            self.keep = []
            return
        dev = self.keep[0].device
        if self._dev is None:
            self._dev = torch.empty(self.cap, dtype=torch.uint8, device=dev)
            self._host = [torch.empty(self.cap, dtype=torch.uint8).pin_memory() for _ in range(2)]
        capturing = torch.cuda.is_current_stream_capturing()
        turn = self._turn
        if self._copied[turn] is not None and not capturing:
            self._copied[turn].synchronize()             # the upload that last read this pinned table has completed
        host = self._host[turn]
        n, blocks = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(L.ts_conv_wgrad_take(ctypes.c_void_p(host.data_ptr()), self.cap, ctypes.byref(n), ctypes.byref(blocks)), "ts_conv_wgrad_take")
        self._dev.copy_(host, non_blocking=True)
        if not capturing:
            ev = torch.cuda.Event()
            ev.record()
            self._copied[turn] = ev
            self._turn ^= 1
        _lib.check(L.ts_conv_wgrad_finish_many(_lib.ptr(self._dev), n.value, blocks.value, _stream()), "ts_conv_wgrad_finish_many")
        self.keep = []


_WGRAD_DEFER = None


def _wgrad_workspace(cin, cout, taps, device):
    """Partial-sum buffer of ``ts_conv3d_*_bwd_weight`` (symmetric in the channel counts' roles: the larger of
    the two orders covers the transposed form, which exchanges them)."""
    L = _lib.lib()
    n = max(_q("ts_conv3d_bwd_weight_workspace_bytes", cin, cout, taps), _q("ts_conv3d_bwd_weight_workspace_bytes", cout, cin, taps))
    return torch.empty(n // 4, dtype=torch.float32, device=device), n


def _hw_backward(x, weight, dy, geom, need_x, need_w):
    B, Cin, Cout, D, H, W, stride, dilation, transposed = geom
    dy = dy.contiguous()
    L = _lib.lib()
    dx = dw = None
    if need_x:
        dx = torch.empty_like(x)
        # stride 1: the input gradient is the same convolution of dy with the flipped taps, Cout -> Cin channels
        # (element (ci' = co, co' = ci, tap) of [Cout, Cin, 1, 3, 3]: ci' * Cin * 9 + co' * 9 + (8 - tap))
        if not (stride == 1 and not transposed and _x6_conv(dy, weight, (Cin * 9, 9, 1, 1), dx, B, Cout, Cin, D, H, W, dilation)):
            if transposed:
                w_b = _layout(weight, 1, 0)                                      # [co][t][ci] = W_T[ci][co][t]
            elif stride == 2:
                w_b = _layout(weight, 0, 1)                                      # [co][t][ci], taps as they are
            else:
                w_b = _layout(weight, 0, 1, flip=True)                           # taps flipped
            rc = L.ts_conv3d_hw_bwd_data(_lib.ptr(dy), _lib.ptr(w_b), _lib.ptr(dx), B, Cin, Cout, D, H, W, stride, dilation,
                                         int(transposed), dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1), _stream())
            _lib.check(rc, "ts_conv3d_hw_bwd_data")
    if need_w:
        dw = torch.empty_like(weight)
        ws, nws = _wgrad_workspace(Cin, Cout, 9, x.device)
        defer = _WGRAD_DEFER if (_WGRAD_DEFER is not None and _WGRAD_DEFER.single_use(weight)) else None
        was = L.ts_conv_wgrad_defer(1) if defer is not None else 0      # (this thread: backward runs on autograd's own)
        try:
            if transposed:     # roles exchanged: a stride-2 convolution maps dy (2H x 2W) to x
                rc = L.ts_conv3d_hw_bwd_weight(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), B, Cout, Cin, D, 2 * H, 2 * W, 2, 1,
                                               dy.stride(0), dy.stride(1), x.stride(0), x.stride(1), _lib.ptr(ws), nws, _stream())
            else:
                rc = L.ts_conv3d_hw_bwd_weight(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), B, Cin, Cout, D, H, W, stride, dilation,
                                               x.stride(0), x.stride(1), dy.stride(0), dy.stride(1), _lib.ptr(ws), nws, _stream())
        finally:
            if defer is not None:
                L.ts_conv_wgrad_defer(was)
                defer.keep.append(ws)
        _lib.check(rc, "ts_conv3d_hw_bwd_weight")
    return dx, dw


class _Conv3dHW(torch.autograd.Function):
    """Raw Conv3d (1,3,3) [padding == dilation] / ConvTranspose3d (1,3,3) stride 2, padding 1, output_padding 1.
    forward ts_conv3d_hw_fwd (no scale / shift / activation), backward ts_conv3d_hw_bwd_{data,weight}.
    Replaces F.conv3d / F.conv_transpose3d inside layers.Conv3d / ConvTranspose3d
    (reference: layers/basic_layers.py:194-235,340-388 reach cuDNN through the same two functionals)."""

    @staticmethod
    def forward(ctx, x, weight, stride, dilation, transposed):
        x, y, ctx.geom = _hw_forward(x, weight, stride, dilation, transposed)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw = _hw_backward(x, weight, dy, ctx.geom, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, None, None, None


def _d_forward(x, weight, stride, dilation, padding, transposed, bias=None, fold=None, act=0):
    """Raw Conv3d (k,1,1) / ConvTranspose3d (3,1,1) stride 2, padding 1, output_padding 1 along D (+ bias); `fold`: see _hw_forward."""
    _require_gpu(x, weight)
    x = x.contiguous()
    B, Cin, Din, H, W = x.shape
    k = weight.shape[2]
    if transposed:
        Cout = weight.shape[1]
        w_t = _layout(weight, 0, 1)
        Dout = 2 * Din
    else:
        Cout = weight.shape[0]
        w_t = _layout(weight, 1, 0)
        Dout = (Din + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    y = torch.empty((B, Cout, Dout, H, W), device=x.device, dtype=torch.float32)
    sc, sh = fold if fold is not None else (None, _shift_vec(bias, _cpad(Cout)))
    rc = _lib.lib().ts_conv3d_d_fwd(_lib.ptr(x), _lib.ptr(w_t), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(y), B, Cin, Cout, Din, H, W, k, stride,
                                    dilation, padding, int(transposed), int(act), 0.0, x.stride(0), x.stride(1), y.stride(0),
                                    y.stride(1), _stream())
    _lib.check(rc, "ts_conv3d_d_fwd")
    return x, y, (B, Cin, Cout, Din, H, W, k, stride, dilation, padding, transposed)


def _d_backward(x, weight, dy, geom, need_x, need_w):
    B, Cin, Cout, Din, H, W, k, stride, dilation, padding, transposed = geom
    dy = dy.contiguous()
    L = _lib.lib()
    dx = dw = None
    if need_x:
        if transposed:
            w_b = _layout(weight, 1, 0)
        elif stride == 2:
            w_b = _layout(weight, 0, 1)
        else:
            w_b = _layout(weight, 0, 1, flip=True)
        dx = torch.empty_like(x)
        rc = L.ts_conv3d_d_bwd_data(_lib.ptr(dy), _lib.ptr(w_b), _lib.ptr(dx), B, Cin, Cout, Din, H, W, k, stride, dilation,
                                    padding, int(transposed), dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1), _stream())
        _lib.check(rc, "ts_conv3d_d_bwd_data")
    if need_w:
        dw = torch.empty_like(weight)
        ws, nws = _wgrad_workspace(Cin, Cout, k, x.device)
        defer = _WGRAD_DEFER if (_WGRAD_DEFER is not None and _WGRAD_DEFER.single_use(weight)) else None
        was = L.ts_conv_wgrad_defer(1) if defer is not None else 0
        try:
            if transposed:
                rc = L.ts_conv3d_d_bwd_weight(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), B, Cout, Cin, 2 * Din, H, W, 3, 2, 1, 1,
                                              dy.stride(0), dy.stride(1), x.stride(0), x.stride(1), _lib.ptr(ws), nws, _stream())
            else:
                rc = L.ts_conv3d_d_bwd_weight(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), B, Cin, Cout, Din, H, W, k, stride, dilation,
                                              padding, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1), _lib.ptr(ws), nws, _stream())
        finally:
            if defer is not None:
                L.ts_conv_wgrad_defer(was)
                defer.keep.append(ws)
        _lib.check(rc, "ts_conv3d_d_bwd_weight")
    return dx, dw


class _Conv3dD(torch.autograd.Function):
    """Raw Conv3d (k,1,1) / ConvTranspose3d (3,1,1) stride 2, padding 1, output_padding 1 along D (+ bias in the kernel's epilogue)."""

    @staticmethod
    def forward(ctx, x, weight, stride, dilation, padding, transposed, bias=None):
        x, y, ctx.geom = _d_forward(x, weight, stride, dilation, padding, transposed, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw = _d_backward(x, weight, dy, ctx.geom, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        gb = _channel_sum(dy) if (ctx.has_bias and ctx.needs_input_grad[6]) else None
        return dx, dw, None, None, None, None, gb


_ONES = {}


def _ones(n, device):
    key = (n, device)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return _ONES[key]


def _zeros_const(n, device, _cache={}):
    key = (n, device)
    if key not in _cache:
        _cache[key] = torch.zeros(n, device=device, dtype=torch.float32)
    return _cache[key]


def _dc_conv3_weight(weight, out=None):
    """[Cin][Cout][4][4] -> [(4*Cout)][9][pad(Cin)]: weight of the 3x3 convolution that is d/dx of the transposed convolution."""
    Cin, Cout = weight.shape[0], weight.shape[1]
    cpad = _cpad(Cin)
    if out is None:
        out = torch.empty((4 * Cout, 9, cpad), device=weight.device, dtype=torch.float32)
    w = weight.detach()
    w = w if w.is_contiguous() else w.contiguous()
    _lib.check(_lib.lib().ts_deconv2d_k4s2_weight_to_conv3(_lib.ptr(w), _lib.ptr(out), Cin, Cout, cpad, _stream()),
               "ts_deconv2d_k4s2_weight_to_conv3")
    return out


def _dc_forward(x, weight, bias=None, fold=None, act=0):
    """Raw ConvTranspose2d(kernel 4, stride 2, padding 1) (+ bias) of UNet.deconv4 / deconv2 (module.py:453-457): ts_deconv2d_k4s2_fwd;
    `fold`: see _hw_forward."""
    _require_gpu(x, weight)
    x = x.contiguous()
    B, Cin, H, W = x.shape
    Cout = weight.shape[1]
    w_t = _layout(weight, 0, 1)                                                   # [ci][16 taps][pad(co)]
    pad = _cpad(Cout)
    y = torch.empty((B, Cout, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
    if fold is not None:
        sc, sh = fold
    else:
        sc, sh = _ones(pad, x.device), (_shift_vec(bias, pad) if bias is not None else _zeros_const(pad, x.device))
    rc = _lib.lib().ts_deconv2d_k4s2_fwd(_lib.ptr(x), _lib.ptr(w_t), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(y),
                                         B, Cin, Cout, H, W, int(act), y.stride(0), _stream())
    _lib.check(rc, "ts_deconv2d_k4s2_fwd")
    return x, y, (B, Cin, Cout, H, W)


def _dc_backward(x, weight, dy, geom, need_x, need_w):
    """d/dx = conv3x3(space_to_depth2(dy)) with the weight's 3x3 form (four parity classes, two taps per axis each: 16 of the 36
    taps are non-zero), d/dW = the weight gradient of that same convolution gathered back to [Cin][Cout][4][4]: both on the MFMA
    convolution kernels of the (1,3,3) family (a plane = depth 1)."""
    B, Cin, Cout, H, W = geom
    dy = dy.contiguous()
    L = _lib.lib()
    z = torch.empty((B, 4 * Cout, H, W), device=dy.device, dtype=torch.float32)
    _lib.check(L.ts_space_to_depth2_fwd(_lib.ptr(dy), _lib.ptr(z), B, Cout, H, W, _stream()), "ts_space_to_depth2_fwd")
    dx = dw = None
    zc = 4 * Cout
    if need_x:
        if _LAYOUTS is not None and weight.is_contiguous():
            w3 = _LAYOUTS.get_custom(weight, "dc3", _dc_conv3_weight)
        else:
            w3 = _dc_conv3_weight(weight)
        dx = torch.empty_like(x)
        wsb = _q("ts_conv3d_hw_workspace_bytes", B, zc, Cin, 1, H, W, 1, 0)
        ws = torch.empty(wsb, device=x.device, dtype=torch.uint8) if wsb else None
        rc = L.ts_conv3d_hw_fwd(_lib.ptr(z), _lib.ptr(w3), None, None, _lib.ptr(dx), B, zc, Cin, 1, H, W, 1, 1, 0, 0, 0.0,
                                z.stride(0), z.stride(1), dx.stride(0), dx.stride(1), None, 0, _lib.ptr(ws), wsb, _stream())
        _lib.check(rc, "ts_conv3d_hw_fwd")
    if need_w:
        dw3 = torch.empty((Cin, zc, 9), device=x.device, dtype=torch.float32)
        ws, nws = _wgrad_workspace(zc, Cin, 9, x.device)
        rc = L.ts_conv3d_hw_bwd_weight(_lib.ptr(z), _lib.ptr(x), _lib.ptr(dw3), B, zc, Cin, 1, H, W, 1, 1,
                                       z.stride(0), z.stride(1), x.stride(0), x.stride(1), _lib.ptr(ws), nws, _stream())
        _lib.check(rc, "ts_conv3d_hw_bwd_weight")
        dw = torch.empty_like(weight)
        _lib.check(L.ts_deconv2d_k4s2_wgrad_from_conv3(_lib.ptr(dw3), _lib.ptr(dw), Cin, Cout, _stream()), "ts_deconv2d_k4s2_wgrad_from_conv3")
    return dx, dw


def deconv2d_k4s2_supported(weight_shape, stride, padding, output_padding, dilation, groups):
    """True when a ConvTranspose2d runs on the HIP kernels: kernel 4, stride 2, padding 1, <= 32 output and <= 64 input channels."""
    return (len(weight_shape) == 4 and tuple(weight_shape[2:]) == (4, 4) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1) and
            tuple(output_padding) == (0, 0) and tuple(dilation) == (1, 1) and groups == 1 and weight_shape[1] <= 32 and weight_shape[0] <= 64)


class _Deconv2dK4S2(torch.autograd.Function):
    """Raw ConvTranspose2d(4, stride 2, padding 1) + bias (UNet.deconv2, module.py:457): ts_deconv2d_k4s2_fwd forward, backward on the
    (1,3,3) convolution kernels (see _dc_backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x, y, ctx.geom = _dc_forward(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw = _dc_backward(x, weight, dy, ctx.geom, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        gb = _channel_sum(dy) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, gb


def conv_transpose2d_k4s2(x, weight, bias=None):
    """F.conv_transpose2d(x, weight, bias, stride 2, padding 1) for 4x4 kernels on the HIP kernels (forward and backward)."""
    return _Deconv2dK4S2.apply(x, weight, bias)


class BNFolds:
    """Eval-mode BatchNorm of a training step folded into the convolution epilogue.  The frames a training step runs in eval() /
    no_grad (every previous frame, projects/TemporalStereo/TemporalStereo.py:268-274) need no batch statistics and no backward: their
    conv -> BatchNorm -> activation is ONE convolution launch with a per-channel scale / shift, as in the inference engine -- but the
    running statistics and the affine parameters move every step, so the folds are recomputed per step: `refresh()` does that for every
    registered layer in one launch (ts_bn_fold_many).  Inside `with folds:` an eval-mode, gradient-free wrapper call takes the fused
    form; its first call registers the layer (and folds it on the spot)."""

    def __init__(self):
        self.entries = {}           # key -> [gamma, beta, mean, var, bias, scale, shift, C, pad]
        self._table, self._dirty, self._eps = None, False, None

    def __enter__(self):
        global _FOLDS
        self._outer, _FOLDS = _FOLDS, self
        return self

    def __exit__(self, *exc):
        global _FOLDS
        _FOLDS = self._outer
        return False

    def _upload(self, entries):
        import numpy as np
        rec = np.zeros(len(entries), dtype=np.dtype([("gamma", "<u8"), ("beta", "<u8"), ("mean", "<u8"), ("var", "<u8"), ("bias", "<u8"),
                                                      ("scale", "<u8"), ("shift", "<u8"), ("C", "<i4"), ("pad", "<i4")]))
        for i, e in enumerate(entries):
            rec[i] = tuple(t.data_ptr() if t is not None else 0 for t in e[:7]) + (e[7], e[8])
        return torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(entries[0][5].device)

    def get(self, gamma, beta, mean, var, bias, eps, cout):
        key = (mean.data_ptr(), var.data_ptr(), gamma.data_ptr() if gamma is not None else 0, bias.data_ptr() if bias is not None else 0)
        e = self.entries.get(key)
        if e is None:
            if self._eps is not None and float(eps) != self._eps:
                return None                                     # one eps per table (every BatchNorm of the model has the default)
            self._eps = float(eps)
            pad = _cpad(cout)
            e = [gamma, beta, mean, var, bias, torch.empty(pad, device=mean.device), torch.empty(pad, device=mean.device), int(cout), pad]
            self.entries[key] = e
            self._dirty = True
            _lib.check(_lib.lib().ts_bn_fold_many(_lib.ptr(self._upload([e])), 1, self._eps, _stream()), "ts_bn_fold_many")
        return e[5], e[6]

    def refresh(self):
        if not self.entries:
            return
        if self._dirty:
            self._table = self._upload(list(self.entries.values()))
            self._dirty = False
        _lib.check(_lib.lib().ts_bn_fold_many(_lib.ptr(self._table), len(self.entries), self._eps, _stream()), "ts_bn_fold_many")


_FOLDS = None
BN_ACT = {None: 0, "SiLU": 1, "ReLU": 2}
_COUNTS = {}


def _count_const(n, device):
    key = (float(n), device)
    if key not in _COUNTS:
        _COUNTS[key] = torch.full((1,), float(n), device=device, dtype=torch.float32)
    return _COUNTS[key]


def _all_gather_flat(out, part, group):
    """out [world * n] <- every rank's part [n] (all_gather_into_tensor where the backend has it: RCCL; a list gather otherwise)."""
    import torch.distributed as dist
    try:
        dist.all_gather_into_tensor(out, part, group=group)
    except (RuntimeError, NotImplementedError):
        world = dist.get_world_size(group)
        parts = [out[i * part.numel():(i + 1) * part.numel()] for i in range(world)]
        tmp = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(tmp, part, group=group)
        for d, t in zip(parts, tmp):
            d.copy_(t)


import os as _os
_BN_FUSED = _os.environ.get("TS_BN_FUSED", "1") != "0"
_EXCHANGES = [0]            # SyncBatchNorm exchanges issued by this process (forward all_gathers + backward all_reduces; bench.py reports them per step)


def _peer_group(group, n_floats):
    """The installed peer-mailbox group (peer.install) when it spans `group` and carries `n_floats` per exchange, else None:
    torch.distributed collectives (a SyncBatchNorm over a sub-group, or with C >= 512, never goes through the wrong mailboxes)."""
    from . import peer
    return peer.for_group(group, n_floats)


class _ConvBNAct(torch.autograd.Function):
    """conv -> BatchNorm -> activation of the reference's wrappers (layers/basic_layers.py:194-235: conv, then `self.norm`, then
    `self.activation`) as ONE autograd node on HIP kernels: the convolution of _Conv3dHW / _Conv3dD (bias in its epilogue),
    ts_bn_stats_fwd (train mode: batch statistics, running statistics updated in the same launch), ts_bn_apply_act_fwd; backward
    ts_bn_act_bwd_{reduce,apply} (the pre-activation is recomputed from the raw convolution output: nothing but that output is
    kept), then the convolution's data / weight gradients.  `group`: a torch.distributed process group -- statistics and the two
    backward sums are exchanged across its ranks (SyncBatchNorm of dist.py, one all_gather forward, one all_reduce backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, family, geom, eps, momentum, act, training, group,
                counter=None, may_fold=False, addend=None):
        # addend (family "hw" only): [B, Cout, 1, Ho, Wo], added to every depth plane of the raw convolution (first_layer_split)
        ctx.has_addend = addend is not None
        if addend is not None and family != "hw":
            raise ValueError("conv_bn_act: an addend needs the (1,3,3) family")
        # may_fold is decided by conv_bn_act in the CALLER's grad mode (inside Function.forward grad mode is always off): the folded
        # path keeps nothing for backward, so it is for calls through which no gradient can flow
        if _FOLDS is not None and not training and may_fold and running_mean is not None:
            fold = _FOLDS.get(gamma, beta, running_mean, running_var, bias, eps, weight.shape[1] if (family == "dc" or geom[-1]) else weight.shape[0])
            if fold is not None:        # eval frame of a training step: conv -> BatchNorm -> activation in the convolution's epilogue
                if family == "hw":
                    return _hw_forward(x, weight, geom[0], geom[1], geom[2], None, fold, act, addend=addend)[1]
                if family == "dc":
                    return _dc_forward(x, weight, None, fold, act)[1]
                return _d_forward(x, weight, geom[0], geom[1], geom[2], geom[3], None, fold, act)[1]
        if family == "hw":
            x, y, cg = _hw_forward(x, weight, geom[0], geom[1], geom[2], bias, addend=addend)
        elif family == "dc":
            x, y, cg = _dc_forward(x, weight, bias)
        else:
            x, y, cg = _d_forward(x, weight, geom[0], geom[1], geom[2], geom[3], bias)
        B, C = y.shape[0], y.shape[1]
        N = y.numel() // (B * C)
        L = _lib.lib()
        n_total = float(B * N)
        out = torch.empty_like(y)
        if training and group is None and _BN_FUSED:
            # single rank: statistics and their use in two launches (ts_bn_train_fwd)
            mean = torch.empty(C, device=y.device, dtype=torch.float32)
            var = torch.empty_like(mean)
            ws = torch.empty(_q("ts_bn_workspace_bytes", B, C, N), device=y.device, dtype=torch.uint8)
            _lib.check(L.ts_bn_train_fwd(_lib.ptr(y), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(running_mean), _lib.ptr(running_var),
                                         float(momentum), _lib.ptr(counter), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out), _lib.ptr(ws),
                                         B, C, N, y.stride(0), y.stride(1), out.stride(0), out.stride(1), float(eps), int(act), _stream()),
                       "ts_bn_train_fwd")
            ctx.save_for_backward(x, weight, y, mean, var, gamma, beta)
            ctx.meta = (family, cg, float(eps), int(act), bool(training), group, n_total, bias is not None)
            return out
        if training:
            if group is not None:       # the statistics kernel writes straight into the record that is exchanged: [mean | var | count]
                pack = torch.empty(2 * C + 1, device=y.device, dtype=torch.float32)
                mean, var = pack[:C], pack[C:2 * C]
            else:
                mean = torch.empty(C, device=y.device, dtype=torch.float32)
                var = torch.empty_like(mean)
            ws = torch.empty(_q("ts_bn_workspace_bytes", B, C, N), device=y.device, dtype=torch.uint8)
            local_update = group is None and running_mean is not None
            _lib.check(L.ts_bn_stats_fwd(_lib.ptr(y), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(running_mean if local_update else None),
                                         _lib.ptr(running_var if local_update else None), float(momentum), _lib.ptr(counter),
                                         _lib.ptr(ws), B, C, N, y.stride(0), y.stride(1), _stream()), "ts_bn_stats_fwd")
            if group is not None:
                # one all_gather of [mean | var | count], merged on the device (ts_bn_sync_merge); the total count stays there too
                import torch.distributed as dist
                world = dist.get_world_size(group)
                pack[2 * C:].copy_(_count_const(n_total, y.device))
                allp = torch.empty(world * (2 * C + 1), device=y.device, dtype=torch.float32)
                pg = _peer_group(group, 2 * C + 1)
                if pg is not None:      # one kernel over the peer-mapped mailboxes (csrc/peer.hip): no communicator launch, capturable
                    pg.all_gather(pack, allp)
                else:
                    _all_gather_flat(allp, pack, group)
                _EXCHANGES[0] += 1
                mean, var = torch.empty_like(mean), torch.empty_like(var)
                inv_n = torch.empty(1, device=y.device, dtype=torch.float32)
                _lib.check(L.ts_bn_sync_merge(_lib.ptr(allp), world, C, _lib.ptr(mean), _lib.ptr(var), _lib.ptr(running_mean), _lib.ptr(running_var),
                                              float(momentum), _lib.ptr(inv_n), _stream()), "ts_bn_sync_merge")
                n_total = inv_n                       # a device scalar from here on (backward scales its sums with it)
        else:
            mean, var = running_mean, running_var
        _lib.check(L.ts_bn_apply_act_fwd(_lib.ptr(y), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(out),
                                         B, C, N, y.stride(0), y.stride(1), out.stride(0), out.stride(1), float(eps), int(act),
                                         _stream()), "ts_bn_apply_act_fwd")
        ctx.save_for_backward(x, weight, y, mean, var, gamma, beta)
        ctx.meta = (family, cg, float(eps), int(act), bool(training), group, n_total, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, y, mean, var, gamma, beta = ctx.saved_tensors
        family, cg, eps, act, training, group, n_total, has_bias = ctx.meta
        g = g.contiguous()
        B, C = y.shape[0], y.shape[1]
        N = y.numel() // (B * C)
        L = _lib.lib()
        s1 = torch.empty(C, device=y.device, dtype=torch.float32)
        s2 = torch.empty_like(s1)
        ws = torch.empty(_q("ts_bn_workspace_bytes", B, C, N), device=y.device, dtype=torch.uint8)
        if training and group is None and _BN_FUSED:
            dy = torch.empty_like(y)
            _lib.check(L.ts_bn_train_bwd(_lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(gamma), _lib.ptr(beta),
                                         _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(dy), _lib.ptr(ws), B, C, N, y.stride(0), y.stride(1),
                                         g.stride(0), g.stride(1), eps, act, n_total, _stream()), "ts_bn_train_bwd")
            return _ConvBNAct._conv_backward(ctx, x, weight, dy, family, cg, has_bias, (s2, s1) if gamma is not None else (None, None))
        _lib.check(L.ts_bn_act_bwd_reduce(_lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(gamma), _lib.ptr(beta),
                                          _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(ws), B, C, N, y.stride(0), y.stride(1), g.stride(0),
                                          g.stride(1), eps, act, _stream()), "ts_bn_act_bwd_reduce")
        ggamma, gbeta = (s2, s1) if gamma is not None else (None, None)          # local sums: the gradient exchange averages them
        t1, t2 = s1, s2
        count = n_total
        if training and group is not None:
            import torch.distributed as dist
            pack = torch.cat([s1, s2])
            pg = _peer_group(group, 2 * C)
            if pg is not None:
                pg.all_reduce_sum(pack, scale=n_total)               # summed in rank order and scaled inside the exchange kernel
            else:
                dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
                pack = pack * n_total                                # n_total is 1 / (global count) on the device here: pre-scaled sums,
            t1, t2, count = pack[:C], pack[C:], 1.0                  # the kernel's own division is by 1 (no host read of a count)
            _EXCHANGES[0] += 1
        dy = torch.empty_like(y)
        _lib.check(L.ts_bn_act_bwd_apply(_lib.ptr(y), _lib.ptr(g), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(gamma), _lib.ptr(beta),
                                         _lib.ptr(t1), _lib.ptr(t2), _lib.ptr(dy), B, C, N, y.stride(0), y.stride(1), g.stride(0),
                                         g.stride(1), eps, act, int(training), count, _stream()), "ts_bn_act_bwd_apply")
        return _ConvBNAct._conv_backward(ctx, x, weight, dy, family, cg, has_bias, (ggamma, gbeta))

    @staticmethod
    def _conv_backward(ctx, x, weight, dy, family, cg, has_bias, gaffine):
        if family == "hw":
            dx, dw = _hw_backward(x, weight, dy, cg, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        elif family == "dc":
            dx, dw = _dc_backward(x, weight, dy, cg, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        else:
            dx, dw = _d_backward(x, weight, dy, cg, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        gbias = None
        if has_bias and ctx.needs_input_grad[2]:
            gbias = _channel_sum(dy)                               # == 0 up to rounding in train mode (BatchNorm removes the mean)
        gadd = None
        if ctx.has_addend and ctx.needs_input_grad[16]:
            gadd = dy.sum(dim=2, keepdim=True)                     # the D-invariant term reaches every depth plane
        return dx, dw, gbias, gaffine[0], gaffine[1], None, None, None, None, None, None, None, None, None, None, None, gadd


def conv_bn_act(x, weight, bias, bn, activation, family, geom, transposed=False, addend=None):
    """Fused wrapper forward (see _ConvBNAct).  `bn`: an nn.BatchNorm*d / dist.SyncBatchNorm module; `activation`: None | 'SiLU' |
    'ReLU'; family 'hw' geom (stride, dilation, transposed) / family 'd' geom (stride, dilation, padding, transposed)."""
    # parameters / buffers straight from the module's dictionaries: nn.Module.__getattr__ (the fallback every `bn.weight`,
    # `bn.running_mean` ... goes through) was ~10 lookups x 180 wrappers per frame
    if _WGRAD_DEFER is not None and weight.requires_grad and torch.is_grad_enabled():
        _WGRAD_DEFER.used(weight)          # (counted where a backward will follow: no_grad frames do not count)
    d, P, Bf = bn.__dict__, bn._parameters, bn._buffers
    rmean = Bf.get("running_mean")
    training = d["training"] or rmean is None
    group = None
    if training and "process_group" in d:
        import torch.distributed as dist
        pg = d["process_group"]
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(pg) > 1:
            group = pg if pg is not None else dist.group.WORLD
    momentum = d["momentum"] if d["momentum"] is not None else 0.1
    counter = Bf.get("num_batches_tracked") if (training and d["track_running_stats"]) else None     # incremented by the statistics launch
    gamma, beta = P["weight"], P["bias"]
    may_fold = (not training) and (not torch.is_grad_enabled() or not any(
        t is not None and t.requires_grad for t in (x, weight, bias, gamma, beta, addend)))
    return _ConvBNAct.apply(x, weight, bias, gamma, beta, rmean, Bf.get("running_var"), family, geom, d["eps"], momentum,
                            BN_ACT[activation], training, group, counter, may_fold, addend)


def conv3d_supported(weight_shape, stride, padding, dilation, groups, transposed=False, output_padding=(0, 0, 0)):
    """'hw' | 'd' | None: which HIP family runs a Conv3d / ConvTranspose3d with these hyper-parameters."""
    if groups != 1 or len(weight_shape) != 5:
        return None
    kd, kh, kw = weight_shape[2:]
    cout = weight_shape[1] if transposed else weight_shape[0]
    cin = weight_shape[0] if transposed else weight_shape[1]
    if max(cout, cin if transposed else 0) > 64:
        return None                                     # the weight-gradient kernel holds <= 64 output channels
    if (kd, kh, kw) == (1, 3, 3) and stride[0] == 1 and stride[1] == stride[2] and dilation[1] == dilation[2] and \
            padding[0] == 0 and dilation[0] == 1:
        s, d = stride[1], dilation[1]
        if transposed:
            ok = s == 2 and d == 1 and tuple(padding[1:]) == (1, 1) and tuple(output_padding) == (0, 1, 1)
        else:
            ok = tuple(padding[1:]) == (d, d) and (s, d) in ((1, 1), (1, 2), (2, 1))
        return "hw" if ok else None
    if kh == 1 and kw == 1 and kd in (1, 3, 5) and tuple(stride[1:]) == (1, 1) and tuple(padding[1:]) == (0, 0):
        s, d, pd = stride[0], dilation[0], padding[0]
        if transposed:
            ok = kd == 3 and s == 2 and d == 1 and pd == 1 and tuple(output_padding) == (1, 0, 0)
        else:
            ok = s == 1 or (s == 2 and kd == 3 and d == 1 and pd == 1)
        return "d" if ok else None
    return None


def conv3d(x, weight, bias=None, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1)):
    """F.conv3d on the HIP kernels for the (1,3,3) / (k,1,1) families (fp32, GPU)."""
    kind = conv3d_supported(tuple(weight.shape), stride, padding, dilation, 1)
    if _WGRAD_DEFER is not None and weight.requires_grad and torch.is_grad_enabled():
        _WGRAD_DEFER.used(weight)
    if kind == "hw":
        y = _Conv3dHW.apply(x, weight, stride[1], dilation[1], False)
    elif kind == "d":
        return _Conv3dD.apply(x, weight, stride[0], dilation[0], padding[0], False, bias)       # bias in the epilogue
    else:
        raise NotImplementedError("conv3d: kernel %s stride %s padding %s dilation %s has no HIP kernel"
                                  % (tuple(weight.shape[2:]), stride, padding, dilation))
    return y if bias is None else y + bias.view(1, -1, 1, 1, 1)


def conv_transpose3d(x, weight, bias=None, stride=(1, 2, 2), padding=(0, 1, 1), output_padding=(0, 1, 1)):
    """F.conv_transpose3d on the HIP kernels: (1,3,3) stride (1,2,2) or (3,1,1) stride (2,1,1), padding 1, output_padding 1."""
    kind = conv3d_supported(tuple(weight.shape), stride, padding, (1, 1, 1), 1, True, output_padding)
    if _WGRAD_DEFER is not None and weight.requires_grad and torch.is_grad_enabled():
        _WGRAD_DEFER.used(weight)
    if kind == "hw":
        y = _Conv3dHW.apply(x, weight, 2, 1, True)
    elif kind == "d":
        y = _Conv3dD.apply(x, weight, 2, 1, 1, True, None)
    else:
        raise NotImplementedError("conv_transpose3d: kernel %s stride %s padding %s output_padding %s has no HIP kernel"
                                  % (tuple(weight.shape[2:]), stride, padding, output_padding))
    return y if bias is None else y + bias.view(1, -1, 1, 1, 1)


# --------------------------------------------------------------------------------------- element-wise glue of the levels
class _CandidatesInRange(torch.autograd.Function):
    """fine.py:82-93 / precise.py:73-78: the five candidates |high - low| * {0,3,4,5,8}/8 + min(low, high), behind the (detached)
    local-map candidates of the previous frames when there are any -- one launch each way instead of ~13 framework launches."""

    @staticmethod
    def forward(ctx, low, high, local_map):
        _require_gpu(low, high, local_map)
        low, high = _lib.contiguous(low), _lib.contiguous(high)
        B, _, H, W = low.shape
        nl = 0 if local_map is None else local_map.shape[1]
        out = torch.empty((B, nl + 5, H, W), device=low.device, dtype=torch.float32)
        L = _lib.lib()
        if nl:
            lm = _lib.contiguous(local_map)
            h, w = lm.shape[-2:]
            _lib.check(L.ts_resize_bilinear_fwd(_lib.ptr(lm), _lib.ptr(out), B, nl, h, w, H, W, float(W) / w, out.stride(0), _stream()),
                       "ts_resize_bilinear_fwd")
        _lib.check(L.ts_candidates_in_range_fwd(_lib.ptr(low), _lib.ptr(high), _lib.ptr(out), B, H, W, nl, nl + 5, _stream()),
                   "ts_candidates_in_range_fwd")
        ctx.save_for_backward(low, high)
        ctx.nl = nl
        return out

    @staticmethod
    def backward(ctx, g):
        low, high = ctx.saved_tensors
        B, _, H, W = low.shape
        g = _lib.contiguous(g)
        gl, gh = torch.empty_like(low), torch.empty_like(high)
        _lib.check(_lib.lib().ts_candidates_in_range_bwd(_lib.ptr(low), _lib.ptr(high), _lib.ptr(g), _lib.ptr(gl), _lib.ptr(gh), B, H, W,
                                                         ctx.nl, ctx.nl + 5, _stream()), "ts_candidates_in_range_bwd")
        return gl, gh, None


def candidates_in_range(low, high, local_map=None):
    """[B, n_local + 5, H, W]: `local_map` ([B, n_local, h, w], no gradient: the previous frames' state) resized to (H, W) with the
    disparity rescale of fine.py:89-93 in front of the five range candidates of fine.py:82-87."""
    if local_map is not None and local_map.requires_grad:
        raise NotImplementedError("candidates_in_range: the local map is temporal state and carries no gradient here")
    return _CandidatesInRange.apply(low, high, local_map)


class _OffsetHead(torch.autograd.Function):
    """PredictionHeads.regress_offset (module.py:384-390): tanh(x / 100).clamp(-1, 1) * delta, one launch each way."""

    @staticmethod
    def forward(ctx, x, delta):
        _require_gpu(x)
        x = _lib.contiguous(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ts_offset_head_fwd(_lib.ptr(x), _lib.ptr(y), x.numel(), float(delta), _stream()), "ts_offset_head_fwd")
        ctx.save_for_backward(x)
        ctx.delta = float(delta)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _lib.contiguous(g)
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().ts_offset_head_bwd(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gx), x.numel(), ctx.delta, _stream()), "ts_offset_head_bwd")
        return gx, None


def offset_head(x, delta):
    return _OffsetHead.apply(x, delta)


# --------------------------------------------------------------------------------------- K3 element stages
class _ResizeAddSiLU(torch.autograd.Function):
    """silu(F.interpolate(a, size, 'trilinear', align_corners=True) + add): ResidualBlock3D up-steps
    (module.py:285-295) as one kernel each way (ts_resize3d_add_act_{fwd,bwd})."""

    @staticmethod
    def forward(ctx, a, add):
        _require_gpu(a, add)
        a, add = a.contiguous(), add.contiguous()
        B, C, Da, Ha, Wa = a.shape
        D, H, W = add.shape[2:]
        out = torch.empty_like(add)
        na, n = Da * Ha * Wa, D * H * W
        rc = _lib.lib().ts_resize3d_add_act_fwd(_lib.ptr(a), _lib.ptr(add), _lib.ptr(out), B, C, Da, Ha, Wa, D, H, W, 1,
                                                C * na, na, C * n, n, C * n, n, _stream())
        _lib.check(rc, "ts_resize3d_add_act_fwd")
        ctx.save_for_backward(a, add)
        return out

    @staticmethod
    def backward(ctx, g):
        a, add = ctx.saved_tensors
        B, C, Da, Ha, Wa = a.shape
        D, H, W = add.shape[2:]
        g = g.contiguous()
        ga, gadd = torch.empty_like(a), torch.empty_like(add)
        rc = _lib.lib().ts_resize3d_add_act_bwd(_lib.ptr(a), _lib.ptr(add), _lib.ptr(g), _lib.ptr(ga), _lib.ptr(gadd),
                                                B, C, Da, Ha, Wa, D, H, W, 1, _stream())
        _lib.check(rc, "ts_resize3d_add_act_bwd")
        return ga, gadd


def resize_add_silu(a, add):
    """F.silu(F.interpolate(a, size=add.shape[-3:], mode='trilinear', align_corners=True) + add)."""
    return _ResizeAddSiLU.apply(a, add)


class _Pool5AvgMax(torch.autograd.Function):
    """(avg_pool3d, max_pool3d)(x, kernel 5, stride 1, padding 2) of PyramidFusion (module.py:415-417)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        B, C, D, H, W = x.shape
        if min(D, H, W) < 5:      # what F.avg_pool3d(kernel 5) answers, padding or not (PyramidFusion, module.py:416)
            raise RuntimeError("input image (T: %d H: %d W: %d) smaller than kernel size (kT: 5 kH: 5 kW: 5)" % (D, H, W))
        avg, mx = torch.empty_like(x), torch.empty_like(x)
        n = D * H * W
        rc = _lib.lib().ts_pool3d5_avgmax_fwd(_lib.ptr(x), _lib.ptr(avg), _lib.ptr(mx), B, C, D, H, W,
                                              C * n, n, C * n, n, C * n, n, _stream())
        _lib.check(rc, "ts_pool3d5_avgmax_fwd")
        ctx.save_for_backward(x)
        return avg, mx

    @staticmethod
    def backward(ctx, g_avg, g_max):
        (x,) = ctx.saved_tensors
        B, C, D, H, W = x.shape
        gx = torch.empty_like(x)
        # Named references, not temporaries: under torch.cat the two gradients arrive as channel slices, `.contiguous()` makes copies,
        # and a copy that is only alive inside `_lib.ptr(...)` hands its block back to the allocator before the next argument is
        # evaluated -- the second copy then lands in the SAME block and the kernel reads dMax through both pointers (round 3: the
        # composed backward was 2-6 % off below every PyramidFusion while each per-op test, fed contiguous gradients, was green).
        g_avg, g_max = g_avg.contiguous(), g_max.contiguous()
        rc = _lib.lib().ts_pool3d5_avgmax_bwd(_lib.ptr(x), _lib.ptr(g_avg), _lib.ptr(g_max),
                                              _lib.ptr(gx), B, C, D, H, W, _stream())
        _lib.check(rc, "ts_pool3d5_avgmax_bwd")
        return gx


def pool5_avgmax(x):
    """(F.avg_pool3d(x, 5, 1, 2), F.max_pool3d(x, 5, 1, 2)) in one pass over x."""
    return _Pool5AvgMax.apply(x)


class _SortGather(torch.autograd.Function):
    """disp_sample, order = sort(disp_sample, dim=1, stable); volume = gather(volume, dim=2, order)
    (coarse.py:103-105, fine.py:120-122) as one kernel each way."""

    @staticmethod
    def forward(ctx, volume, sample):
        _require_gpu(volume, sample)
        volume, sample = volume.contiguous(), sample.contiguous()
        B, C, DT, H, W = volume.shape
        out_v, out_s = torch.empty_like(volume), torch.empty_like(sample)
        n = DT * H * W
        rc = _lib.lib().ts_merge_candidates_fwd(_lib.ptr(volume), _lib.ptr(sample), None, None, None, None, None,
                                                _lib.ptr(out_s), _lib.ptr(out_v), B, C, DT, 0, H, W, C * n, n, C * n, n, _stream())
        _lib.check(rc, "ts_merge_candidates_fwd")
        ctx.save_for_backward(sample)
        return out_v, out_s

    @staticmethod
    def backward(ctx, g_v, g_s):
        (sample,) = ctx.saved_tensors
        B, DT, H, W = sample.shape
        g_v = g_v.contiguous()
        g_s = g_s.contiguous() if g_s is not None else None
        C = g_v.shape[1]
        gv, gs = torch.empty_like(g_v), torch.empty_like(sample)
        rc = _lib.lib().ts_merge_candidates_bwd(_lib.ptr(sample), _lib.ptr(g_v), _lib.ptr(g_s),
                                                _lib.ptr(gv), _lib.ptr(gs), B, C, DT, H, W, _stream())
        _lib.check(rc, "ts_merge_candidates_bwd")
        return gv, gs


def sort_gather(volume, sample):
    """-> (volume permuted along D by the stable ascending order of `sample`, sorted sample)."""
    return _SortGather.apply(volume, sample)


def block_cost_warped(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    """Inference form of the sampled block_cost WITHOUT its first C channels (the D-fold repeat of
    reference_fm, block_cost.py:51): [B, C + scales*C/8, D, H, W].  No autograd.  Used with the
    `addend` of ts_conv3d_hw_fwd, see include/ts_hip.h."""
    _require_gpu(reference_fm, target_fm, disp_sample)
    left, right, disp = _lib.contiguous(reference_fm), _lib.contiguous(target_fm), _lib.contiguous(disp_sample)
    B, C, H, W = left.shape
    D = disp.shape[1]
    if C % 8 != 0:
        raise ValueError("channel count must be a multiple of 8 (block_cost.py:9)")
    L = _lib.lib()
    out = torch.empty((B, C + block_cost_scale * (C // 8), D, H, W), device=left.device, dtype=torch.float32)
    ws = torch.empty(max(int(L.ts_block_cost_workspace_bytes(B, C, H, W, D, block_cost_scale)), 256),
                     device=left.device, dtype=torch.uint8)

    def launch():
        return L.ts_block_cost_sampled_warped_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out),
                                                  _lib.ptr(ws), B, C, H, W, D, block_cost_scale, _stream())
    rc = launch() if _k1_probe is None else _k1_probe((B, C, H, W, D, "warped"), launch)
    _lib.check(rc, "ts_block_cost_sampled_warped_fwd")
    return out


class _BlockCostWarped(torch.autograd.Function):
    """The sampled block_cost WITHOUT its first C channels (block_cost_warped) with autograd: ts_block_cost_sampled_warped_{fwd,bwd}.
    The training path's first layer of a sampled level takes the D-invariant left half as a per-pixel term (first_layer_split)."""

    @staticmethod
    def forward(ctx, left, right, disp, scales):
        out = block_cost_warped(left, right, disp, scales)
        ctx.save_for_backward(_lib.contiguous(left), _lib.contiguous(right), _lib.contiguous(disp))
        ctx.scales = int(scales)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        left, right, disp = ctx.saved_tensors
        B, C, H, W = left.shape
        D, scales = disp.shape[1], ctx.scales
        L = _lib.lib()
        grad_out = _lib.contiguous(grad_out)
        gl = torch.empty_like(left) if ctx.needs_input_grad[0] else None
        gr = torch.empty_like(right) if ctx.needs_input_grad[1] else None
        gd = torch.empty_like(disp) if ctx.needs_input_grad[2] else None
        ws = torch.empty(max(int(L.ts_block_cost_bwd_workspace_bytes(B, C, H, W, D, scales)), 256), device=left.device, dtype=torch.uint8)
        _lib.check(L.ts_block_cost_sampled_warped_bwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(grad_out), _lib.ptr(gl),
                                                      _lib.ptr(gr), _lib.ptr(gd), _lib.ptr(ws), B, C, H, W, D, scales, _stream()),
                   "ts_block_cost_sampled_warped_bwd")
        return gl, gr, gd, None


class _SplitWeight(torch.autograd.Function):
    """weight [Cout, Cin, ...] -> (weight[:, :n], weight[:, n:]) as CONTIGUOUS tensors; backward: one concatenation."""

    @staticmethod
    def forward(ctx, weight, n):
        ctx.n = int(n)
        return weight[:, :n].contiguous(), weight[:, n:].contiguous()

    @staticmethod
    def backward(ctx, g1, g2):
        return torch.cat([g1, g2], dim=1), None


def first_layer_split(conv, left, volume):
    """The (1,3,3) convolution + BatchNorm + activation over a sampled level's volume [left x D | warped right | corr]
    (precise.py:88-91, fine.py:96-103 over block_cost.py:47-58) WITHOUT the left half in the volume: its channels are the left features
    repeated over the D candidates (block_cost.py:51), so their share of the layer is a 2-D convolution of `left`, computed once per
    pixel and added to every depth plane before the bias / BatchNorm -- in training as the inference engine has done since round 1.
    `conv`: the layers.Conv3d wrapper (its weight covers all 2C + scales*C/8 input channels); `volume` = block_cost_warped_ag(...):
    [B, C + scales*C/8, D, H, W].  The layer's forward reads 42 % fewer channels, its input gradient and weight gradient likewise, the
    cost volume's backward receives no gradient for the half that is not there; the left term's own backward is two single-plane
    launches on the sum over D of the layer's output gradient."""
    C = left.shape[1]
    w1, w2 = _SplitWeight.apply(conv._parameters["weight"], C)
    dil = conv.dilation[1]
    lt = _Conv3dHW.apply(left.unsqueeze(2), w1, 1, dil, False)                  # [B, Cout, 1, H, W], raw
    act = conv._fusable()
    if act is False or conv.stride[1] != 1:
        raise NotImplementedError("first_layer_split: a stride-1 (1,3,3) convolution + BatchNorm (+ SiLU / ReLU)")
    if _WGRAD_DEFER is not None:
        _WGRAD_DEFER.used(w2)               # its gradient is concatenated with w1's DURING backward: never a deferred finish (counts as shared)
    return conv_bn_act(volume, w2, conv._parameters["bias"], conv._modules["norm"], act, "hw", (1, dil, False), addend=lt)


def block_cost_warped_ag(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    return _BlockCostWarped.apply(reference_fm, target_fm, disp_sample, int(block_cost_scale))


def block_cost_corr(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    """Inference form of the sampled block_cost WITHOUT its 2C main channels: the correlation blocks alone,
    [B, scales*C/8, D, H, W] = block_cost(...)[:, 2C:].  No autograd.  The consumer is ts_conv3d_hw_warp_fwd, which takes the
    warped half of the volume in pre-contracted form (include/ts_hip.h)."""
    _require_gpu(reference_fm, target_fm, disp_sample)
    left, right, disp = _lib.contiguous(reference_fm), _lib.contiguous(target_fm), _lib.contiguous(disp_sample)
    B, C, H, W = left.shape
    D = disp.shape[1]
    if C % 8 != 0:
        raise ValueError("channel count must be a multiple of 8 (block_cost.py:9)")
    L = _lib.lib()
    out = torch.empty((B, block_cost_scale * (C // 8), D, H, W), device=left.device, dtype=torch.float32)
    ws = torch.empty(max(_q("ts_block_cost_workspace_bytes", B, C, H, W, D, block_cost_scale), 256), device=left.device, dtype=torch.uint8)

    def launch():
        return L.ts_block_cost_sampled_corr_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out),
                                                _lib.ptr(ws), B, C, H, W, D, block_cost_scale, _stream())
    rc = launch() if _k1_probe is None else _k1_probe((B, C, H, W, D, "corr"), launch)
    _lib.check(rc, "ts_block_cost_sampled_corr_fwd")
    return out


def block_cost(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    """Cost volume of `block_cost` (block_cost.py:16-83), same arguments and output layout.

    disp_sample: int -> integer candidates 0..D-1 (out [B, C+s*C/8, D, H, W]);
                 Tensor [B,D,H,W] -> per-pixel candidates (out [B, 2C+s*C/8, D, H, W]).
    """
    scales = int(block_cost_scale)
    if isinstance(disp_sample, int):
        return _BlockCost.apply(reference_fm, target_fm, None, disp_sample, scales)
    return _BlockCost.apply(reference_fm, target_fm, disp_sample, 0, scales)


# --------------------------------------------------------------------------------------------- K5 (training form)
class _ConvexUpsample(torch.autograd.Function):
    """ConvexUpsample's combination step (module.py:337-353): softmax over the k*k logits of every output pixel, weighted sum of
    the 3x3 neighbourhood of `disp * scale`.  logits [B, 9*r*r, H, W], disp [B,1,H,W] -> [B,1,H*r,W*r]."""

    @staticmethod
    def forward(ctx, logits, disp, r, scale):
        _require_gpu(logits, disp)
        logits, disp = _lib.contiguous(logits), _lib.contiguous(disp)
        B, _, H, W = disp.shape
        if logits.shape[1] != 9 * r * r or disp.shape[1] != 1:
            raise ValueError("convex upsample: logits must be [B, 9*r*r, H, W] and disp [B,1,H,W]")
        out = torch.empty((B, 1, H * r, W * r), device=disp.device, dtype=torch.float32)
        _lib.check(_lib.lib().ts_convex_upsample_fwd(_lib.ptr(logits), _lib.ptr(disp), _lib.ptr(out), B, H, W, r, float(scale), _stream()),
                   "ts_convex_upsample_fwd")
        ctx.save_for_backward(logits, disp)
        ctx.meta = (r, float(scale))
        return out

    @staticmethod
    def backward(ctx, g):
        logits, disp = ctx.saved_tensors
        r, scale = ctx.meta
        B, _, H, W = disp.shape
        gl = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        gd = torch.empty_like(disp) if ctx.needs_input_grad[1] else None
        g = _lib.contiguous(g)
        _lib.check(_lib.lib().ts_convex_upsample_bwd(_lib.ptr(logits), _lib.ptr(disp), _lib.ptr(g), _lib.ptr(gl), _lib.ptr(gd),
                                                     B, H, W, r, scale, _stream()), "ts_convex_upsample_bwd")
        return gl, gd, None, None


def convex_upsample(logits, disp, upscale_factor, disp_scale):
    return _ConvexUpsample.apply(logits, disp, int(upscale_factor), float(disp_scale))


class _UNetUpsample(torch.autograd.Function):
    """UNet.upsample (module.py:468-482): softmax over the 9 logit planes, bilinear (align_corners) upsampling of the unfolded
    disparity * Wo / w, weighted sum.  logits [B,9,Ho,Wo], disp [B,1,h,w] -> [B,1,Ho,Wo]."""

    @staticmethod
    def forward(ctx, logits, disp):
        _require_gpu(logits, disp)
        logits, disp = _lib.contiguous(logits), _lib.contiguous(disp)
        B, _, Ho, Wo = logits.shape
        h, w = disp.shape[-2:]
        out = torch.empty((B, 1, Ho, Wo), device=disp.device, dtype=torch.float32)
        _lib.check(_lib.lib().ts_unet_upsample_fwd(_lib.ptr(logits), _lib.ptr(disp), _lib.ptr(out), B, h, w, Ho, Wo, _stream()),
                   "ts_unet_upsample_fwd")
        ctx.save_for_backward(logits, disp)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, disp = ctx.saved_tensors
        B, _, Ho, Wo = logits.shape
        h, w = disp.shape[-2:]
        gl = torch.empty_like(logits)
        gd = torch.empty_like(disp) if ctx.needs_input_grad[1] else None
        ws = torch.empty_like(logits)
        g = _lib.contiguous(g)
        _lib.check(_lib.lib().ts_unet_upsample_bwd(_lib.ptr(logits), _lib.ptr(disp), _lib.ptr(g), _lib.ptr(gl), _lib.ptr(gd),
                                                   _lib.ptr(ws), B, h, w, Ho, Wo, _stream()), "ts_unet_upsample_bwd")
        return gl, gd


def unet_upsample(logits, disp):
    return _UNetUpsample.apply(logits, disp)


class _ResizePair(torch.autograd.Function):
    """(F.interpolate(x0 * s0, size), F.interpolate(x1 * s1, size)), bilinear, align_corners=True, in one launch
    (ts_resize_bilinear_pair_fwd): the top-k memory's candidates and costs (precise.py:100-103, coarse.py:91-96).  The memory is
    temporal state -- nothing in a training step differentiates it -- so the backward (the framework's own adjoint, recomputed)
    exists for completeness."""

    @staticmethod
    def forward(ctx, x0, x1, H, W, s0, s1):
        _require_gpu(x0, x1)
        a, b = _lib.contiguous(x0), _lib.contiguous(x1)
        if a.shape != b.shape or a.dim() != 4:
            raise ValueError("resize_bilinear_pair: two [B,C,h,w] maps of one shape")
        B, C, h, w = a.shape
        o0 = torch.empty((B, C, H, W), device=a.device, dtype=torch.float32)
        o1 = torch.empty_like(o0)
        _lib.check(_lib.lib().ts_resize_bilinear_pair_fwd(_lib.ptr(a), _lib.ptr(b), _lib.ptr(o0), _lib.ptr(o1), B, C, h, w, H, W,
                                                          float(s0), float(s1), _stream()), "ts_resize_bilinear_pair_fwd")
        ctx.meta = (tuple(a.shape), H, W, float(s0), float(s1))
        ctx.set_materialize_grads(False)
        return o0, o1

    @staticmethod
    def backward(ctx, g0, g1):
        shape, H, W, s0, s1 = ctx.meta
        outs = []
        for g, sc in ((g0, s0), (g1, s1)):
            if g is None:
                outs.append(None)
                continue
            with torch.enable_grad():
                x = torch.zeros(shape, device=g.device, dtype=g.dtype, requires_grad=True)
                y = torch.nn.functional.interpolate(x * sc, size=(H, W), mode="bilinear", align_corners=True)
            outs.append(torch.autograd.grad(y, x, g)[0])
        return outs[0], outs[1], None, None, None, None


def resize_bilinear_pair(x0, x1, size, scale0=1.0, scale1=1.0):
    return _ResizePair.apply(x0, x1, int(size[0]), int(size[1]), float(scale0), float(scale1))


_CONST = {}


def const_zeros(shape, device):
    """A shared all-zero tensor (READ-ONLY by contract: the absent cost memory of a first frame) instead of a fill launch per call."""
    key = ("z", tuple(shape), device)
    if key not in _CONST:
        if torch.cuda.is_current_stream_capturing():       # a capture's allocations belong to its pool: nothing of them is kept
            return torch.zeros(shape, device=device, dtype=torch.float32)
        _CONST[key] = torch.zeros(shape, device=device, dtype=torch.float32)
    return _CONST[key]


def const_arange(n, device):
    key = ("a", int(n), device)
    if key not in _CONST:
        if torch.cuda.is_current_stream_capturing():
            return torch.arange(n, device=device, dtype=torch.float32)
        _CONST[key] = torch.arange(n, device=device, dtype=torch.float32)
    return _CONST[key]


class _WeightedTotal(torch.autograd.Function):
    """sum_i w_i * term_i of 0-d loss terms: one stack + one dot product forward, one multiplication backward (the framework's
    form -- a multiplication per term each way, a stack and a sum -- was ~25 one-element launches per training step)."""

    @staticmethod
    def forward(ctx, wvec, *terms):
        ctx.save_for_backward(wvec)
        return torch.dot(torch.stack([t.reshape(()) for t in terms]), wvec)

    @staticmethod
    def backward(ctx, g):
        (wvec,) = ctx.saved_tensors
        gv = wvec * g
        return (None,) + tuple(gv.unbind(0))


_WVEC = {}


def weighted_total(terms, weights):
    """sum_i weights[i] * terms[i] (0-d tensors; weights: python floats)."""
    key = (tuple(float(w) for w in weights), terms[0].device)
    if key not in _WVEC:
        _WVEC[key] = torch.tensor(key[0], device=terms[0].device, dtype=torch.float32)
    return _WeightedTotal.apply(_WVEC[key], *terms)


# --------------------------------------------------------------------------------------------- K4
class _TopkSoftArgmax(torch.autograd.Function):
    """K4a.  ts_topk_softargmax_{fwd,bwd}; returns (disp, topk_disp, topk_cost)."""

    @staticmethod
    def forward(ctx, cost, sample, offset, k):
        _require_gpu(cost, sample, offset)
        if cost.dim() != 4 or cost.shape != sample.shape or cost.shape != offset.shape:
            raise ValueError("cost, disp_sample and off must be [B,D,H,W] of equal shape")
        cost, sample, offset = _lib.contiguous(cost), _lib.contiguous(sample), _lib.contiguous(offset)
        B, D, H, W = cost.shape
        disp = torch.empty((B, 1, H, W), device=cost.device, dtype=torch.float32)
        tdisp = torch.empty((B, k, H, W), device=cost.device, dtype=torch.float32)
        tcost = torch.empty_like(tdisp)
        tidx = torch.empty((B, k, H, W), device=cost.device, dtype=torch.int32)
        rc = _lib.lib().ts_topk_softargmax_fwd(_lib.ptr(cost), _lib.ptr(sample), _lib.ptr(offset), _lib.ptr(disp),
                                               _lib.ptr(tdisp), _lib.ptr(tcost), _lib.ptr(tidx), B, D, H, W, int(k),
                                               _stream())
        _lib.check(rc, "ts_topk_softargmax_fwd")
        ctx.save_for_backward(tdisp, tcost, tidx, disp)
        ctx.meta = (B, D, H, W, int(k))
        ctx.set_materialize_grads(False)         # unused outputs (the top-k planes when nothing differentiates the memory) arrive as None,
        return disp, tdisp, tcost                # which the kernel takes as NULL: no zero-fill launches

    @staticmethod
    def backward(ctx, g_disp, g_tdisp, g_tcost):
        tdisp, tcost, tidx, disp = ctx.saved_tensors
        B, D, H, W, k = ctx.meta
        need_cost = ctx.needs_input_grad[0]
        need_samp = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gcost = torch.empty((B, D, H, W), device=disp.device, dtype=torch.float32) if need_cost else None
        gsamp = torch.empty((B, D, H, W), device=disp.device, dtype=torch.float32) if need_samp else None
        c = lambda t: None if t is None else _lib.contiguous(t)
        rc = _lib.lib().ts_topk_softargmax_bwd(_lib.ptr(tdisp), _lib.ptr(tcost), _lib.ptr(tidx), _lib.ptr(disp),
                                               _lib.ptr(c(g_disp)), _lib.ptr(c(g_tdisp)), _lib.ptr(c(g_tcost)),
                                               _lib.ptr(gcost), _lib.ptr(gsamp), B, D, H, W, k, _stream())
        _lib.check(rc, "ts_topk_softargmax_bwd")
        return (gcost, gsamp if ctx.needs_input_grad[1] else None,
                gsamp if ctx.needs_input_grad[2] else None, None)


def topk_softargmax(cost, disp_sample, off, k=2):
    """predict_disp() of the reference levels (coarse.py:69-75): (disp_map, topk_disp, topk_cost)."""
    return _TopkSoftArgmax.apply(cost, disp_sample, off, int(k))


class _SoftArgmin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, sample, temperature, normalize):
        _require_gpu(cost, sample)
        cost, sample = _lib.contiguous(cost), _lib.contiguous(sample)
        B, D, H, W = cost.shape
        disp = torch.empty((B, 1, H, W), device=cost.device, dtype=torch.float32)
        rc = _lib.lib().ts_softargmin_fwd(_lib.ptr(cost), _lib.ptr(sample), _lib.ptr(disp), float(temperature),
                                          int(bool(normalize)), B, D, H, W, _stream())
        _lib.check(rc, "ts_softargmin_fwd")
        ctx.save_for_backward(cost, sample, disp)
        ctx.meta = (B, D, H, W, float(temperature), int(bool(normalize)))
        return disp

    @staticmethod
    def backward(ctx, g):
        cost, sample, disp = ctx.saved_tensors
        B, D, H, W, temperature, normalize = ctx.meta
        gc = torch.empty_like(cost) if ctx.needs_input_grad[0] else None
        gs = torch.empty_like(sample) if ctx.needs_input_grad[1] else None
        g = _lib.contiguous(g)
        rc = _lib.lib().ts_softargmin_bwd(_lib.ptr(cost), _lib.ptr(sample), _lib.ptr(disp), _lib.ptr(g),
                                          _lib.ptr(gc), _lib.ptr(gs), temperature, normalize, B, D, H, W, _stream())
        _lib.check(rc, "ts_softargmin_bwd")
        return gc, gs, None, None


def soft_argmin(cost_volume, disp_sample, temperature=1.0, normalize=True):
    """SOFTARGMIN.forward (prediction/soft_argmin.py:38-59)."""
    if cost_volume.dim() != 4:
        raise ValueError('expected 4D input (got {}D input)'.format(cost_volume.dim()))
    if cost_volume.shape != disp_sample.shape:
        raise ValueError('The shape of disparity samples and cost volume should be consistent!')
    return _SoftArgmin.apply(cost_volume, disp_sample, temperature, normalize)


class _ArgmaxSelect(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, sample):
        _require_gpu(cost, sample)
        cost, sample = _lib.contiguous(cost), _lib.contiguous(sample)
        B, D, H, W = cost.shape
        disp = torch.empty((B, 1, H, W), device=cost.device, dtype=torch.float32)
        idx = torch.empty((B, 1, H, W), device=cost.device, dtype=torch.int32)
        rc = _lib.lib().ts_argmax_select_fwd(_lib.ptr(cost), _lib.ptr(sample), _lib.ptr(disp), _lib.ptr(idx),
                                             B, D, H, W, _stream())
        _lib.check(rc, "ts_argmax_select_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = sample.shape
        return disp

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gs = torch.zeros(ctx.shape, device=g.device, dtype=g.dtype)
        gs.scatter_(1, idx.long(), g)          # index plumbing only; one element per pixel
        return None, gs


def argmin_select(cost_volume, disp_sample):
    """ARGMIN.forward (prediction/argmin.py:35-46): gather the sample of the best (max) cost."""
    if cost_volume.shape != disp_sample.shape:
        raise ValueError("{}, {}".format(cost_volume.shape, disp_sample.shape))
    return _ArgmaxSelect.apply(cost_volume, disp_sample)


# --------------------------------------------------------------------------------------------- K2
class _SplatSum(torch.autograd.Function):
    """_FunctionSoftsplat of the reference (softsplat.py:239-332) on ts_softsplat_sum_*."""

    @staticmethod
    def forward(ctx, inp, flow, deterministic=False):
        _require_gpu(inp, flow)
        if flow.shape[1] != 2 or inp.shape[0] != flow.shape[0] or inp.shape[2:] != flow.shape[2:]:
            raise ValueError("flow must be [B,2,H,W] matching the input")
        inp, flow = _lib.contiguous(inp), _lib.contiguous(flow)
        B, C, H, W = inp.shape
        out = torch.empty_like(inp)
        if deterministic:      # 64-bit fixed-point accumulation: the same bits every run
            ws = torch.empty(B * C * H * W * 8, device=inp.device, dtype=torch.uint8)
            _lib.check(_lib.lib().ts_softsplat_sum_fwd_deterministic(_lib.ptr(inp), _lib.ptr(flow), _lib.ptr(out), _lib.ptr(ws),
                                                                     B, C, H, W, _stream()), "ts_softsplat_sum_fwd_deterministic")
        else:
            _lib.check(_lib.lib().ts_softsplat_sum_fwd(_lib.ptr(inp), _lib.ptr(flow), _lib.ptr(out), B, C, H, W, _stream()),
                       "ts_softsplat_sum_fwd")
        ctx.save_for_backward(inp, flow)
        return out

    @staticmethod
    def backward(ctx, g):
        inp, flow = ctx.saved_tensors
        B, C, H, W = inp.shape
        g = _lib.contiguous(g)
        gi = gf = None
        if ctx.needs_input_grad[0]:
            gi = torch.empty_like(inp)
            _lib.check(_lib.lib().ts_softsplat_sum_bwd_input(_lib.ptr(flow), _lib.ptr(g), _lib.ptr(gi), B, C, H, W,
                                                             _stream()), "ts_softsplat_sum_bwd_input")
        if ctx.needs_input_grad[1]:
            gf = torch.empty_like(flow)
            _lib.check(_lib.lib().ts_softsplat_sum_bwd_flow(_lib.ptr(inp), _lib.ptr(flow), _lib.ptr(g), _lib.ptr(gf),
                                                            B, C, H, W, _stream()), "ts_softsplat_sum_bwd_flow")
        return gi, gf, None


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType, deterministic=False):
    """Forward splatting, same signature as the reference's FunctionSoftsplat (softsplat.py:334-360).
    deterministic=True (an addition): order-independent accumulation, bit-identical from run to run.

    'softmax' on inputs that do not require grad (the only use in update_map) runs the fused kernel
    ts_softsplat_softmax_fwd; every other case composes the summation splat (with HIP backward)."""
    if tenMetric is not None and tenMetric.shape[1] != 1:
        raise ValueError("tenMetric must have one channel")
    if strType not in ('summation', 'average', 'linear', 'softmax'):
        raise ValueError("unknown splatting type %r" % (strType,))
    grad_needed = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (tenInput, tenFlow, tenMetric))
    if strType == 'softmax' and not grad_needed and not deterministic:
        _require_gpu(tenInput, tenFlow, tenMetric)
        inp, flow, met = _lib.contiguous(tenInput), _lib.contiguous(tenFlow), _lib.contiguous(tenMetric)
        B, C, H, W = inp.shape
        L = _lib.lib()
        out = torch.empty_like(inp)
        ws = torch.empty(int(L.ts_softsplat_softmax_workspace_bytes(B, C, H, W)), device=inp.device, dtype=torch.uint8)
        _lib.check(L.ts_softsplat_softmax_fwd(_lib.ptr(inp), _lib.ptr(flow), _lib.ptr(met), _lib.ptr(out), _lib.ptr(ws),
                                              B, C, H, W, _stream()), "ts_softsplat_softmax_fwd")
        return out
    x = tenInput
    if strType == 'average':
        x = torch.cat([x, x.new_ones(x.shape[0], 1, x.shape[2], x.shape[3])], 1)
    elif strType == 'linear':
        x = torch.cat([x * tenMetric, tenMetric], 1)
    elif strType == 'softmax':
        e = tenMetric.exp()
        x = torch.cat([x * e, e], 1)
    out = _SplatSum.apply(x, tenFlow, deterministic)
    if strType != 'summation':
        out = out[:, :-1, :, :] / (out[:, -1:, :, :] + 1e-22)
    return out


def reproject_memory(prev_disp, mem_disp, mem_cost, local_map, n_local_out, K, T_a, T_b, baseline, factor, h, w):
    """The fused temporal state update (ts_reproject_memory_fwd): what update_past_cost and update_local_map of
    the reference (projects/TemporalStereo/TemporalStereo.py:386-426, :340-384) compute together, three launches.

    prev_disp [B,1,H,W]; mem_disp/mem_cost [B,k,h,w] or None; local_map [B,n,h,w] or None; K [B,3|4,3|4] full-
    resolution intrinsics; T = T_a @ T_b (T_b may be None); baseline scalar or tensor with B elements.
    Returns (moved mem_disp, moved mem_cost, moved local map), None where not requested."""
    _require_gpu(prev_disp, K, T_a)
    if prev_disp.dim() != 4 or prev_disp.shape[1] != 1:
        raise RuntimeError("reproject_memory: prev_disp must be [B,1,H,W], got %s" % (tuple(prev_disp.shape),))
    B, _, Hf, Wf = prev_disp.shape
    dev = prev_disp.device
    prev_disp = _lib.contiguous(prev_disp)
    k = 0
    if mem_disp is not None:
        mem_disp, mem_cost = _lib.contiguous(mem_disp), _lib.contiguous(mem_cost)
        k = mem_disp.shape[1]
        if tuple(mem_disp.shape) != (B, k, h, w) or mem_cost.shape != mem_disp.shape:
            raise RuntimeError("reproject_memory: memory shapes %s / %s, expected [%d,k,%d,%d]"
                               % (tuple(mem_disp.shape), tuple(mem_cost.shape), B, h, w))
    n_in = 0
    if local_map is not None:
        local_map = _lib.contiguous(local_map)
        n_in = local_map.shape[1]
        if tuple(local_map.shape) != (B, n_in, h, w):
            raise RuntimeError("reproject_memory: local map %s, expected [%d,n,%d,%d]" % (tuple(local_map.shape), B, h, w))
    n_local_out = min(int(n_local_out), n_in + 1)
    K, T_a = _lib.contiguous(K), _lib.contiguous(T_a)
    T_b = None if T_b is None else _lib.contiguous(T_b)
    base_t, base_s = None, 0.0
    if torch.is_tensor(baseline):
        if baseline.numel() == 1 and baseline.device.type == 'cpu':
            base_s = float(baseline)
        else:
            base_t = _lib.contiguous(baseline.to(device=dev, dtype=torch.float32).reshape(-1).expand(B))
    else:
        base_s = float(baseline)
    f32 = dict(device=dev, dtype=torch.float32)
    out_d = torch.empty((B, k, h, w), **f32) if k else None
    out_c = torch.empty((B, k, h, w), **f32) if k else None
    out_l = torch.empty((B, n_local_out, h, w), **f32) if n_local_out else None
    L = _lib.lib()
    ws = torch.empty(L.ts_reproject_memory_workspace_bytes(B, h, w, k, n_local_out), device=dev, dtype=torch.uint8)
    opt = lambda t: None if t is None else _lib.ptr(t)
    _lib.check(L.ts_reproject_memory_fwd(_lib.ptr(prev_disp), prev_disp.stride(0), Hf, Wf, opt(mem_disp), opt(mem_cost), k,
                                         opt(local_map), n_in, n_local_out, _lib.ptr(K), K.shape[-1], _lib.ptr(T_a), opt(T_b),
                                         opt(base_t), base_s, float(factor), opt(out_d), opt(out_c), opt(out_l),
                                         _lib.ptr(ws), B, h, w, _stream()), "ts_reproject_memory_fwd")
    return out_d, out_c, out_l


def project_to_3d(depth, K, inv_K=None, T_target_to_source=None, eps=1e-7):
    """project_to_3d of the reference (layers/inverse_warp.py:92-178), forward only (the project
    calls it on detached tensors).  Returns triangular_depth / optical_flow / flow_mask;
    homo_points_3d and src_pixel_coord (= optical_flow + the pixel grid) are not produced: no caller
    on the hot path reads them."""
    if T_target_to_source is None:
        raise NotImplementedError("the hot path always passes T_target_to_source")
    _require_gpu(depth, K, T_target_to_source)
    if inv_K is None:
        inv_K = torch.inverse(K[:, :3, :3])
    depth = _lib.contiguous(depth)
    K, inv_K, T = _lib.contiguous(K), _lib.contiguous(inv_K), _lib.contiguous(T_target_to_source)
    B, C, H, W = depth.shape
    tri = torch.empty_like(depth)
    flow = torch.empty((B, 2 * C, H, W), device=depth.device, dtype=torch.float32)
    mask = torch.empty((B, C, H, W), device=depth.device, dtype=torch.uint8)
    _lib.check(_lib.lib().ts_project_to_3d_fwd(_lib.ptr(depth), _lib.ptr(K), _lib.ptr(inv_K), _lib.ptr(T), _lib.ptr(tri),
                                               _lib.ptr(flow), _lib.ptr(mask), B, C, H, W, K.shape[-1], inv_K.shape[-1],
                                               float(eps), _stream()), "ts_project_to_3d_fwd")
    out = {'triangular_depth': tri, 'optical_flow': flow, 'flow_mask': mask.bool()}
    return out
