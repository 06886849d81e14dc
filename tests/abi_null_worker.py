"""Worker of tests/test_abi.py::test_every_launching_entry_refuses_empty_arguments: calls EVERY launching entry of the C ABI with
null pointers and zero sizes, in a process of its own (a missing check would be a segmentation fault, not an assertion).  Prints
`name rc` per entry.  Validation happens before any launch, so no GPU is needed."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temporalstereo_amd import _lib

L = _lib.lib()
for name in sorted(_lib.SIGNATURES):
    res, args = _lib.SIGNATURES[name]
    if name in _lib._QUERIES or res is not ctypes.c_int or name in ("ts_stream_fork", "ts_event_record", "ts_event_wait"):
        continue
    vals = []
    for a in args:
        if a in (ctypes.c_int, ctypes.c_longlong, ctypes.c_size_t, ctypes.c_uint, ctypes.c_ulonglong):
            vals.append(0)
        elif a in (ctypes.c_float, ctypes.c_double):
            vals.append(0.0)
        else:
            vals.append(None)
    rc = getattr(L, name)(*vals)
    print(name, rc, flush=True)
print("done")
