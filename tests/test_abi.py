"""CPU: libts_hip.so builds, loads, and exports every symbol include/ts_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ts_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built_lib():
    from temporalstereo_amd import build
    return build.build(verbose=False)


def test_header_symbols_exported(built_lib):
    handle = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 8
    for name in syms:
        assert hasattr(handle, name), "libts_hip.so does not export %s" % name


def test_python_signature_table_matches_header(built_lib):
    from temporalstereo_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib().ts_version() >= 1
    assert _lib.lib().ts_last_error_string() is not None


def test_argument_errors_without_gpu(built_lib):
    """Validation happens before any launch, so it is checkable on CPU."""
    from temporalstereo_amd import _lib
    L = _lib.lib()
    assert L.ts_block_cost_workspace_bytes(1, 12, 8, 8, 3, 3) == 0          # C % 8 != 0
    assert L.ts_block_cost_workspace_bytes(1, 16, 8, 8, 3, 3) > 0
    rc = L.ts_block_cost_int_fwd(None, None, None, None, 1, 16, 8, 8, 3, 3, None)
    assert rc == -1 and b"NULL" in L.ts_last_error_string()
    rc = L.ts_block_cost_int_fwd(None, None, None, None, 1, 12, 8, 8, 3, 3, None)
    assert rc == -2
    rc = L.ts_block_cost_sampled_fwd(None, None, None, None, None, 1, 16, 8, 8, 1, 3, None)
    assert rc == -3                                                          # D == 1 rejected
    with pytest.raises(RuntimeError):
        _lib.check(rc, "ts_block_cost_sampled_fwd")


def test_plan_api_without_gpu(built_lib):
    """Recording validates names and arity up front (csrc/plan.hip); nothing is launched here."""
    import ctypes as C
    from temporalstereo_amd import _lib
    L = _lib.lib()
    plan = L.ts_plan_create()
    assert plan
    w = (C.c_ulonglong * 32)()
    assert L.ts_plan_add_call(plan, b"ts_no_such_entry", w, 0) == -3
    assert L.ts_plan_add_call(plan, b"ts_conv_cout_pad", w, 1) == -3            # a query, not a launch
    assert L.ts_plan_add_call(plan, b"ts_copy_rows_fwd", w, 3) == -2            # takes 7 arguments
    assert L.ts_plan_add_call(plan, b"ts_copy_rows_fwd", w, 7) == 0
    assert L.ts_plan_length(plan) == 1
    assert L.ts_plan_run(plan) == -2 and b"copy_rows" in L.ts_last_error_string()   # zero rows: refused, not launched
    assert L.ts_plan_run(None) == -1
    L.ts_plan_destroy(plan)
    # every launching entry point of the header can be recorded: the plan's table is complete
    launching = [n for n, (res, _) in _lib.SIGNATURES.items() if res is _lib.c_int and n not in _lib._QUERIES]
    plan = L.ts_plan_create()
    for n in launching:
        assert L.ts_plan_add_call(plan, n.encode(), w, len(_lib.SIGNATURES[n][1])) == 0, n
    assert L.ts_plan_length(plan) == len(launching)
    L.ts_plan_destroy(plan)


def test_ops_refuse_cpu_tensors():
    import torch
    import temporalstereo_amd as ts
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ts.block_cost(torch.zeros(1, 8, 4, 4), torch.zeros(1, 8, 4, 4), 3)


def test_x6_split_k_choice_without_gpu(built_lib):
    """ts_conv3d_hw_x6_workspace_bytes is host logic: which (1,3,3) layers of config 2 the x6 kernel cuts into slices of its input
    channels (long reductions on small grids only: csrc/conv3d.hip x6_ksplit, measured in tools/exp/x6_splitk_bench.py)."""
    from temporalstereo_amd import _lib
    L = _lib.lib()
    out = lambda B, Co, D, H, W: B * Co * D * H * W * 4
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 352, 32, 12, 34, 60) == 4 * out(1, 32, 12, 34, 60)     # coarse first layer: four slices
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 256, 64, 1, 34, 60) == 8 * out(1, 64, 1, 34, 60)       # 20 workgroups, 16 chunks
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 32, 32, 12, 34, 60) == 0                               # short reductions never
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 176, 16, 5, 68, 120) == 0
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 128, 32, 14, 34, 60) == 0
    assert L.ts_conv3d_hw_x6_workspace_bytes(4, 352, 32, 12, 34, 60) == 0                              # batch 4 fills the chip unsplit
    assert L.ts_conv3d_hw_x6_workspace_bytes(1, 272, 16, 2, 9, 16) % out(1, 16, 2, 9, 16) == 0         # ragged chunk count: whole slices
    assert L.ts_conv3d_hw_x6_workspace_bytes(0, 352, 32, 12, 34, 60) == 0
    # without a workspace the entry point runs unsplit; its argument checks come first either way
    rc = L.ts_conv3d_hw_x6_fwd(None, None, None, None, None, 1, 352, 32, 12, 34, 60, 1, 0, 0.0, 0, 0, 0, 0, None, 0, None, 0, None)
    assert rc == -1 and b"NULL" in L.ts_last_error_string()


def test_every_launching_entry_refuses_empty_arguments(built_lib):
    """Error behaviour of the boundary (include/ts_hip.h: status ints + ts_last_error_string): every launching entry called with null
    pointers and zero sizes returns a non-zero status -- none dereferences, none launches, none answers TS_OK.  Run in a child
    process: a missing check would be a crash."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "abi_null_worker.py")
    p = subprocess.run([sys.executable, worker], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [l.split() for l in p.stdout.strip().splitlines()]
    assert p.returncode == 0 and lines and lines[-1] == ["done"], "worker died after %s\n%s" % (lines[-1] if lines else "-", p.stderr[-2000:])
    accepted = [n for n, rc in lines[:-1] if int(rc) == 0]
    assert not accepted, "entries that answered TS_OK to empty arguments: %s" % accepted
    assert len(lines) > 80
