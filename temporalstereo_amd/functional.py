"""torch.autograd.Function wrappers over the C ABI (include/ts_hip.h).

Function-level drop-ins for the reference's hot-path ops:
  block_cost        architecture/modeling/aggregation/utils/block_cost.py:16
All ops require fp32 CUDA(HIP) tensors on an MI355X and raise otherwise: there is no CPU path.
"""
import torch

from . import _lib


def _stream():
    return _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("temporalstereo_amd ops run on the GPU only (got a %s tensor); "
                               "there is deliberately no CPU fallback" % t.device)
        if t.dtype != torch.float32:
            raise TypeError("temporalstereo_amd ops are fp32 (got %s)" % t.dtype)


class _BlockCost(torch.autograd.Function):
    """K1.  forward: ts_block_cost_{int,sampled}_fwd; backward: ts_block_cost_{int,sampled}_bwd."""

    @staticmethod
    def forward(ctx, left, right, disp, num_disp, scales):
        _require_gpu(left, right, disp)
        if left.dim() != 4 or left.shape != right.shape:
            raise ValueError("reference_fm / target_fm must be [B,C,H,W] of equal shape")
        left = left.contiguous()
        right = right.contiguous()
        B, C, H, W = left.shape
        L = _lib.lib()
        sampled = disp is not None
        if sampled:
            if disp.dim() != 4 or disp.shape[0] != B or disp.shape[2:] != left.shape[2:]:
                raise ValueError("disp_sample must be [B,D,H,W] matching the feature maps")
            disp = disp.contiguous()
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
        if sampled:
            rc = L.ts_block_cost_sampled_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(disp), _lib.ptr(out),
                                             _lib.ptr(ws), B, C, H, W, D, scales, _stream())
        else:
            rc = L.ts_block_cost_int_fwd(_lib.ptr(left), _lib.ptr(right), _lib.ptr(out), _lib.ptr(ws),
                                         B, C, H, W, D, scales, _stream())
        _lib.check(rc, "ts_block_cost_%s_fwd" % ("sampled" if sampled else "int"))
        ctx.save_for_backward(left, right, disp if sampled else None)
        ctx.meta = (B, C, H, W, D, scales, sampled)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        left, right, disp = ctx.saved_tensors
        B, C, H, W, D, scales, sampled = ctx.meta
        L = _lib.lib()
        grad_out = grad_out.contiguous()
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


def block_cost(reference_fm, target_fm, disp_sample, block_cost_scale=3):
    """Cost volume of `block_cost` (block_cost.py:16-83), same arguments and output layout.

    disp_sample: int -> integer candidates 0..D-1 (out [B, C+s*C/8, D, H, W]);
                 Tensor [B,D,H,W] -> per-pixel candidates (out [B, 2C+s*C/8, D, H, W]).
    """
    scales = int(block_cost_scale)
    if isinstance(disp_sample, int):
        return _BlockCost.apply(reference_fm, target_fm, None, disp_sample, scales)
    return _BlockCost.apply(reference_fm, target_fm, disp_sample, 0, scales)
