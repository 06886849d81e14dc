"""ctypes loader for libts_hip.so -- the only way the package reaches its kernels.

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised
(a GPU box that silently ran something else would void every parity / performance claim).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libts_hip.so")

_lock = threading.Lock()
_lib = None

c_f32p = ctypes.c_void_p     # device pointers are passed as integers
c_ptr = ctypes.c_void_p
c_int = ctypes.c_int
c_size = ctypes.c_size_t
c_float = ctypes.c_float

# name -> (restype, argtypes).  Mirrors include/ts_hip.h (tests/test_abi.py checks the two agree).
SIGNATURES = {
    "ts_version": (c_int, []),
    "ts_last_error_string": (ctypes.c_char_p, []),
    "ts_block_cost_workspace_bytes": (c_size, [c_int] * 6),
    "ts_block_cost_int_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_warped_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_corr_fwd": (c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_conv3d_hw_warp_workspace_bytes": (c_size, [c_int] * 5),
    "ts_conv3d_hw_warp_fwd": (c_int, [c_f32p] * 8 + [c_int] * 8 + [c_float] + [ctypes.c_longlong] * 5 + [c_ptr, c_size, c_ptr]),
    "ts_cat_fms_fwd": (c_int, [c_f32p] * 4 + [c_int] * 5 + [c_ptr]),
    "ts_inverse_warp_3d_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [c_ptr]),
    "ts_dif_fms_workspace_bytes": (c_size, []),
    "ts_dif_fms_fwd": (c_int, [c_f32p] * 4 + [c_ptr] + [c_int] * 5 + [c_ptr]),
    "ts_correlation_fwd": (c_int, [c_f32p] * 3 + [c_int] * 7 + [c_ptr]),
    "ts_correlation_bwd": (c_int, [c_f32p] * 6 + [c_int] * 7 + [c_ptr]),
    "ts_block_cost_bwd_workspace_bytes": (c_size, [c_int] * 6),
    "ts_block_cost_int_bwd": (c_int, [c_f32p] * 5 + [c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_bwd": (c_int, [c_f32p] * 7 + [c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_warped_bwd": (c_int, [c_f32p] * 7 + [c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_calib_stream": (c_int, [c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "ts_candidates_in_range_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [c_ptr]),
    "ts_candidates_in_range_bwd": (c_int, [c_f32p] * 5 + [c_int] * 5 + [c_ptr]),
    "ts_offset_head_fwd": (c_int, [c_f32p] * 2 + [ctypes.c_longlong, c_float, c_ptr]),
    "ts_offset_head_bwd": (c_int, [c_f32p] * 3 + [ctypes.c_longlong, c_float, c_ptr]),
    "ts_space_to_depth2_fwd": (c_int, [c_f32p] * 2 + [c_int] * 4 + [c_ptr]),
    "ts_deconv2d_k4s2_weight_to_conv3": (c_int, [c_f32p] * 2 + [c_int] * 3 + [c_ptr]),
    "ts_deconv2d_k4s2_wgrad_from_conv3": (c_int, [c_f32p] * 2 + [c_int] * 2 + [c_ptr]),
    "ts_bn_fold_many": (c_int, [c_ptr, c_int, c_float, c_ptr]),
    "ts_clip_rmsprop_workspace_bytes": (c_size, [c_int]),
    "ts_clip_rmsprop_step": (c_int, [c_ptr, c_int] + [c_float] * 4 + [c_ptr, c_size, c_ptr]),
    "ts_topk_softargmax_fwd": (c_int, [c_f32p] * 7 + [c_int] * 5 + [c_ptr]),
    "ts_topk_softargmax_bwd": (c_int, [c_f32p] * 9 + [c_int] * 5 + [c_ptr]),
    "ts_softargmin_fwd": (c_int, [c_f32p] * 3 + [c_float, c_int] + [c_int] * 4 + [c_ptr]),
    "ts_softargmin_bwd": (c_int, [c_f32p] * 6 + [c_float, c_int] + [c_int] * 4 + [c_ptr]),
    "ts_argmax_select_fwd": (c_int, [c_f32p] * 4 + [c_int] * 4 + [c_ptr]),
    "ts_bn_workspace_bytes": (c_size, [c_int, c_int, ctypes.c_longlong]),
    "ts_channel_splice_fwd": (c_int, [c_f32p] * 3 + [c_int] * 3 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_bn_sync_merge": (c_int, [c_f32p, c_int, c_int] + [c_f32p] * 4 + [c_float, c_f32p, c_ptr]),
    "ts_bn_train_fwd": (c_int, [c_f32p] * 5 + [c_float, c_ptr] + [c_f32p] * 3 + [c_ptr, c_int, c_int] + [ctypes.c_longlong] * 5 +
                        [c_float, c_int, c_ptr]),
    "ts_bn_train_bwd": (c_int, [c_f32p] * 9 + [c_ptr, c_int, c_int] + [ctypes.c_longlong] * 5 + [c_float, c_int, c_float, c_ptr]),
    "ts_bn_stats_fwd": (c_int, [c_f32p] * 5 + [c_float, c_ptr, c_ptr, c_int, c_int] + [ctypes.c_longlong] * 3 + [c_ptr]),
    "ts_bn_apply_act_fwd": (c_int, [c_f32p] * 6 + [c_int, c_int] + [ctypes.c_longlong] * 5 + [c_float, c_int, c_ptr]),
    "ts_bn_act_bwd_reduce": (c_int, [c_f32p] * 8 + [c_ptr, c_int, c_int] + [ctypes.c_longlong] * 5 + [c_float, c_int, c_ptr]),
    "ts_bn_act_bwd_apply": (c_int, [c_f32p] * 9 + [c_int, c_int] + [ctypes.c_longlong] * 5 + [c_float, c_int, c_int, c_float, c_ptr]),
    "ts_wasserstein_loss_workspace_bytes": (c_size, [c_int] * 3),
    "ts_wasserstein_loss_fwd": (c_int, [c_f32p] * 6 + [c_ptr] + [c_int] * 6 + [c_float, c_float, c_int, c_ptr]),
    "ts_wasserstein_loss_bwd": (c_int, [c_f32p] * 7 + [c_int] * 5 + [c_float, c_float, c_ptr]),
    "ts_disp_smooth_l1_workspace_bytes": (c_size, [c_int] * 3),
    "ts_disp_smooth_l1_fwd": (c_int, [c_f32p] * 3 + [c_ptr] + [c_int] * 5 + [c_float, c_float, c_ptr]),
    "ts_disp_smooth_l1_bwd": (c_int, [c_f32p] * 5 + [c_int] * 5 + [c_float, c_float, c_ptr]),
    "ts_softsplat_sum_fwd": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_fwd_deterministic": (c_int, [c_f32p] * 3 + [c_ptr] + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_bwd_input": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_bwd_flow": (c_int, [c_f32p] * 4 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_softmax_workspace_bytes": (c_size, [c_int] * 4),
    "ts_softsplat_softmax_fwd": (c_int, [c_f32p] * 4 + [c_ptr] + [c_int] * 4 + [c_ptr]),
    "ts_conv_cout_pad": (c_int, [c_int]),
    "ts_peer_region_bytes": (c_size, []),
    "ts_peer_max_floats": (c_int, []),
    "ts_peer_max_ranks": (c_int, []),
    "ts_peer_alloc": (c_int, [ctypes.POINTER(c_ptr), c_ptr]),
    "ts_peer_open": (c_int, [c_ptr, ctypes.POINTER(c_ptr)]),
    "ts_peer_close": (c_int, [c_ptr]),
    "ts_peer_free": (c_int, [c_ptr]),
    "ts_peer_status": (c_int, [c_ptr, ctypes.POINTER(c_int), c_ptr]),
    "ts_bn_set_small_elems": (ctypes.c_longlong, [ctypes.c_longlong]),
    "ts_channel_sum_fwd": (c_int, [c_f32p, c_f32p, c_ptr, c_int, c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_ptr]),
    "ts_peer_status_async": (c_int, [c_ptr, c_ptr, c_ptr]),
    "ts_peer_set_timeout_ms": (ctypes.c_longlong, [ctypes.c_longlong]),
    "ts_peer_reset": (c_int, [c_ptr, c_ptr]),
    "ts_peer_all_gather": (c_int, [c_ptr, c_f32p, c_f32p, c_int, c_ptr]),
    "ts_peer_all_reduce_sum": (c_int, [c_ptr, c_f32p, c_int, c_f32p, c_ptr]),
    "ts_conv_weight_layout": (c_int, [c_f32p, c_f32p] + [c_int] * 4 + [ctypes.c_longlong] * 3 + [c_int, c_ptr]),
    "ts_conv_weight_layout_many": (c_int, [c_ptr, c_int, c_int, c_ptr]),
    "ts_conv_weight_layout_many2": (c_int, [c_ptr, c_int, c_int, c_ptr]),
    "ts_conv_wgrad_defer": (c_int, [c_int]),
    "ts_conv_wgrad_pending": (c_int, []),
    "ts_conv_wgrad_take": (c_int, [c_ptr, c_size, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "ts_conv_wgrad_finish_many": (c_int, [c_ptr, c_int, c_int, c_ptr]),
    "ts_conv_set_chunk_cap": (c_int, [c_int]),
    "ts_conv3d_hw_x6_supported": (c_int, [c_int] * 6),
    "ts_conv3d_hw_x6_weight_bytes": (ctypes.c_size_t, [c_int] * 2),
    "ts_conv3d_hw_x6_weight_split": (c_int, [c_f32p, c_ptr, c_int, c_int, c_ptr]),
    "ts_conv3d_hw_x6_weight_split_from": (c_int, [c_f32p, c_ptr, c_int, c_int] + [ctypes.c_longlong] * 3 + [c_int, c_ptr]),
    "ts_conv3d_hw_x6_workspace_bytes": (c_size, [c_int] * 6),
    "ts_conv3d_hw_x6s_supported": (c_int, [c_int] * 5),
    "ts_conv3d_hw_x6s_weight_bytes": (c_size, [c_int] * 3),
    "ts_conv3d_hw_x6s_weight_split": (c_int, [c_f32p, c_ptr, c_int, c_int, c_int, c_int, c_ptr]),
    "ts_conv3d_hw_x6s_fwd": (c_int, [c_f32p, c_ptr, c_f32p, c_f32p, c_f32p] + [c_int] * 8 + [c_float] + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_conv3d_hw_x6_fwd": (c_int, [c_f32p, c_ptr, c_f32p, c_f32p, c_f32p] + [c_int] * 8 + [c_float] + [ctypes.c_longlong] * 4 +
                            [c_f32p, ctypes.c_longlong, c_ptr, c_size, c_ptr]),
    "ts_conv3d_hw_fwd": (c_int, [c_f32p] * 5 + [c_int] * 10 + [c_float] + [ctypes.c_longlong] * 4 +
                         [c_f32p, ctypes.c_longlong, c_ptr, c_size, c_ptr]),
    "ts_conv3d_hw_workspace_bytes": (c_size, [c_int] * 8),
    "ts_conv3d_hw_bwd_data": (c_int, [c_f32p] * 3 + [c_int] * 9 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_conv3d_bwd_weight_workspace_bytes": (ctypes.c_size_t, [c_int] * 3),
    "ts_conv3d_hw_bwd_weight": (c_int, [c_f32p] * 3 + [c_int] * 8 + [ctypes.c_longlong] * 4 + [c_ptr, ctypes.c_size_t, c_ptr]),
    "ts_conv3d_d_bwd_data": (c_int, [c_f32p] * 3 + [c_int] * 11 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_conv3d_d_bwd_weight": (c_int, [c_f32p] * 3 + [c_int] * 10 + [ctypes.c_longlong] * 4 + [c_ptr, ctypes.c_size_t, c_ptr]),
    "ts_conv3d_d_fwd": (c_int, [c_f32p] * 5 + [c_int] * 12 + [c_float] + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_resize3d_add_act_fwd": (c_int, [c_f32p] * 3 + [c_int] * 9 + [ctypes.c_longlong] * 6 + [c_ptr]),
    "ts_pool3d5_avgmax_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [ctypes.c_longlong] * 6 + [c_ptr]),
    "ts_merge_candidates_fwd": (c_int, [c_f32p] * 9 + [c_int] * 6 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_convex_upsample_fwd": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_float, c_ptr]),
    "ts_convex_upsample_candidates_fwd": (c_int, [c_f32p] * 6 + [c_int] * 4 + [c_float, c_float, c_int, c_int, c_ptr]),
    "ts_unet_upsample_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [c_ptr]),
    "ts_convex_upsample_bwd": (c_int, [c_f32p] * 5 + [c_int] * 4 + [c_float, c_ptr]),
    "ts_unet_upsample_bwd": (c_int, [c_f32p] * 5 + [c_ptr] + [c_int] * 5 + [c_ptr]),
    "ts_deconv2d_k4s2_fwd": (c_int, [c_f32p] * 5 + [c_int] * 6 + [ctypes.c_longlong, c_ptr]),
    "ts_resize_bilinear_fwd": (c_int, [c_f32p] * 2 + [c_int] * 6 + [c_float, ctypes.c_longlong, c_ptr]),
    "ts_resize_bilinear_pair_fwd": (c_int, [c_f32p] * 4 + [c_int] * 6 + [c_float, c_float, c_ptr]),
    "ts_range_candidates_fwd": (c_int, [c_f32p] * 4 + [c_int] * 3 + [c_float, c_int, c_int, c_ptr]),
    "ts_project_to_3d_fwd": (c_int, [c_f32p] * 7 + [c_int] * 6 + [c_float, c_ptr]),
    "ts_reproject_memory_workspace_bytes": (c_size, [c_int] * 5),
    "ts_reproject_memory_fwd": (c_int, [c_f32p, ctypes.c_longlong, c_int, c_int, c_f32p, c_f32p, c_int, c_f32p, c_int, c_int,
                                        c_f32p, c_int, c_f32p, c_f32p, c_f32p, c_float, c_float, c_f32p, c_f32p, c_f32p,
                                        c_ptr, c_int, c_int, c_int, c_ptr]),
    "ts_resize3d_add_act_bwd": (c_int, [c_f32p] * 5 + [c_int] * 9 + [c_ptr]),
    "ts_pool3d5_avgmax_bwd": (c_int, [c_f32p] * 4 + [c_int] * 5 + [c_ptr]),
    "ts_merge_candidates_bwd": (c_int, [c_f32p] * 5 + [c_int] * 5 + [c_ptr]),
    "ts_copy_rows_fwd": (c_int, [c_f32p] * 2 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_plan_create": (c_ptr, []),
    "ts_plan_destroy": (None, [c_ptr]),
    "ts_plan_length": (c_int, [c_ptr]),
    "ts_plan_add_call": (c_int, [c_ptr, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong), c_int]),
    "ts_plan_run": (c_int, [c_ptr]),
    "ts_stream_fork": (c_int, [c_ptr, c_ptr]),
    "ts_event_record": (c_int, [c_int, c_ptr]),
    "ts_event_wait": (c_int, [c_int, c_ptr]),
}

# entry points that only answer a question (nothing is enqueued): never part of a recorded plan
_QUERIES = frozenset(n for n in SIGNATURES if n.endswith("_bytes") or n.startswith("ts_plan_") or
                     n in ("ts_version", "ts_last_error_string", "ts_conv_cout_pad", "ts_conv3d_hw_x6_supported", "ts_conv3d_hw_x6s_supported", "ts_peer_max_floats", "ts_peer_max_ranks",
                           "ts_peer_alloc", "ts_peer_open", "ts_peer_close", "ts_peer_free", "ts_peer_status", "ts_peer_status_async",
                           "ts_peer_set_timeout_ms", "ts_peer_reset", "ts_bn_set_small_elems",
                           "ts_conv_wgrad_defer", "ts_conv_wgrad_pending", "ts_conv_wgrad_take"))


def lib():
    """The loaded library (loads on first use).  Raises RuntimeError when it is not built.
    While a Recorder is active on this thread, launching calls are also appended to its plan."""
    rec = getattr(_tls, "recorder", None)
    if rec is not None:
        return rec.proxy
    return _real_lib()


def _real_lib():
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "temporalstereo_amd: %s is missing -- build it with "
                        "`python -m temporalstereo_amd.build` (there is no fallback path)" % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name, None)
                    if fn is None:
                        raise RuntimeError("libts_hip.so does not export %s (stale build?)" % name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


_tls = threading.local()


def _word(argtype, value):
    """One 64-bit plan word for a ctypes argument (include/ts_hip.h: ts_plan_add_call)."""
    if isinstance(value, ctypes._SimpleCData):
        value = value.value
    if argtype is c_float:
        import struct
        return struct.unpack("<I", struct.pack("<f", float(value)))[0]
    if value is None:
        return 0
    return int(value) & 0xFFFFFFFFFFFFFFFF


class _RecordingProxy:
    def __init__(self, recorder):
        self._rec = recorder

    def __getattr__(self, name):
        real = getattr(_real_lib(), name)
        if name in _QUERIES:
            return real
        argtypes = SIGNATURES[name][1]
        rec = self._rec

        def call(*args):
            rc = real(*args)
            if rc == 0:
                words = (ctypes.c_ulonglong * len(args))(*[_word(t, a) for t, a in zip(argtypes, args)])
                rec.log.append((name, int(words[len(args) - 1]) if len(args) else 0))      # (entry point, its last word: the stream)
                rc2 = _real_lib().ts_plan_add_call(rec.plan, name.encode(), words, len(args))
                if rc2 != 0:
                    check(rc2, "ts_plan_add_call(%s)" % name)
            return rc
        return call


class Recorder:
    """Context manager: every launching C-ABI call made on this thread inside the block runs as usual
    AND is appended to a native plan (csrc/plan.hip).  Tensors whose pointers were handed to a call
    are kept alive by the recorder, so the plan's baked-in pointers stay valid as long as it lives."""

    def __init__(self):
        self.plan = _real_lib().ts_plan_create()
        if not self.plan:
            raise RuntimeError("ts_plan_create failed")
        self.keep = []
        self.log = []               # what was recorded, in order: (entry point, stream handle) -- tools/exp/plan_streams.py
        self.proxy = _RecordingProxy(self)

    def __enter__(self):
        if getattr(_tls, "recorder", None) is not None:
            raise RuntimeError("plan recording does not nest")
        _tls.recorder = self
        return self

    def __exit__(self, *exc):
        _tls.recorder = None
        return False

    def __len__(self):
        return int(_real_lib().ts_plan_length(self.plan))

    def run(self):
        rc = _real_lib().ts_plan_run(self.plan)
        if rc != 0:
            check(rc, "ts_plan_run")

    def __del__(self):
        try:
            if self.plan:
                _real_lib().ts_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass


def recording():
    """True while a Recorder is active on this thread (ops must then avoid un-recorded torch kernels)."""
    return getattr(_tls, "recorder", None) is not None


def check(rc, what):
    """Turn a non-zero status of the C ABI into RuntimeError (include/ts_hip.h conventions)."""
    if rc != 0:
        msg = _real_lib().ts_last_error_string().decode("utf-8", "replace")
        kind = "argument error" if rc < 0 else "hipError_t"
        raise RuntimeError("%s failed (%s %d): %s" % (what, kind, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL), as the plain integer ctypes takes for a void* argument."""
    if t is None:
        return None
    rec = getattr(_tls, "recorder", None)
    if rec is not None:
        rec.keep.append(t)
    return t.data_ptr()


def current_stream_handle():
    """hipStream_t of the current device's current stream as an integer.  The raw-handle query of the framework's C layer:
    `torch.cuda.current_stream()` builds a Stream object and walks the device-index helpers on every call (~9 us; an eager
    pass or a training step asks ~2500 times)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def contiguous(t):
    """t.contiguous(), except that a copy (an un-recorded torch kernel) is refused while recording."""
    if t.is_contiguous():
        return t
    if recording():
        raise RuntimeError("plan recording needs contiguous tensors (a .contiguous() copy would not be replayed)")
    return t.contiguous()
