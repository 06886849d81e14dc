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
    "ts_block_cost_bwd_workspace_bytes": (c_size, [c_int] * 6),
    "ts_block_cost_int_bwd": (c_int, [c_f32p] * 5 + [c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_block_cost_sampled_bwd": (c_int, [c_f32p] * 7 + [c_ptr] + [c_int] * 6 + [c_ptr]),
    "ts_calib_stream": (c_int, [c_int, c_ptr, c_ptr, c_size, c_ptr]),
    "ts_topk_softargmax_fwd": (c_int, [c_f32p] * 7 + [c_int] * 5 + [c_ptr]),
    "ts_topk_softargmax_bwd": (c_int, [c_f32p] * 9 + [c_int] * 5 + [c_ptr]),
    "ts_softargmin_fwd": (c_int, [c_f32p] * 3 + [c_float, c_int] + [c_int] * 4 + [c_ptr]),
    "ts_softargmin_bwd": (c_int, [c_f32p] * 6 + [c_float, c_int] + [c_int] * 4 + [c_ptr]),
    "ts_argmax_select_fwd": (c_int, [c_f32p] * 4 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_fwd": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_bwd_input": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_sum_bwd_flow": (c_int, [c_f32p] * 4 + [c_int] * 4 + [c_ptr]),
    "ts_softsplat_softmax_workspace_bytes": (c_size, [c_int] * 4),
    "ts_softsplat_softmax_fwd": (c_int, [c_f32p] * 4 + [c_ptr] + [c_int] * 4 + [c_ptr]),
    "ts_conv_cout_pad": (c_int, [c_int]),
    "ts_conv3d_hw_fwd": (c_int, [c_f32p] * 5 + [c_int] * 10 + [c_float] + [ctypes.c_longlong] * 4 + [c_ptr, c_size, c_ptr]),
    "ts_conv3d_hw_workspace_bytes": (c_size, [c_int] * 8),
    "ts_conv3d_d_fwd": (c_int, [c_f32p] * 5 + [c_int] * 12 + [c_float] + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_resize3d_add_act_fwd": (c_int, [c_f32p] * 3 + [c_int] * 9 + [ctypes.c_longlong] * 6 + [c_ptr]),
    "ts_pool3d5_avgmax_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [ctypes.c_longlong] * 6 + [c_ptr]),
    "ts_merge_candidates_fwd": (c_int, [c_f32p] * 9 + [c_int] * 6 + [ctypes.c_longlong] * 4 + [c_ptr]),
    "ts_convex_upsample_fwd": (c_int, [c_f32p] * 3 + [c_int] * 4 + [c_float, c_ptr]),
    "ts_unet_upsample_fwd": (c_int, [c_f32p] * 3 + [c_int] * 5 + [c_ptr]),
    "ts_deconv2d_k4s2_fwd": (c_int, [c_f32p] * 5 + [c_int] * 6 + [ctypes.c_longlong, c_ptr]),
    "ts_resize_bilinear_fwd": (c_int, [c_f32p] * 2 + [c_int] * 6 + [c_float, ctypes.c_longlong, c_ptr]),
    "ts_range_candidates_fwd": (c_int, [c_f32p] * 4 + [c_int] * 3 + [c_float, c_int, c_int, c_ptr]),
    "ts_project_to_3d_fwd": (c_int, [c_f32p] * 7 + [c_int] * 6 + [c_float, c_ptr]),
}


def lib():
    """The loaded library (loads on first use).  Raises RuntimeError when it is not built."""
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


def check(rc, what):
    """Turn a non-zero status of the C ABI into RuntimeError (include/ts_hip.h conventions)."""
    if rc != 0:
        msg = lib().ts_last_error_string().decode("utf-8", "replace")
        kind = "argument error" if rc < 0 else "hipError_t"
        raise RuntimeError("%s failed (%s %d): %s" % (what, kind, rc, msg))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())
