"""Builds libts_hip.so (all HIP kernels + the C ABI of include/ts_hip.h) in-tree for gfx950.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the working tree.
Usage:  python -m temporalstereo_amd.build [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libts_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=True):
    """Compile every csrc/*.hip into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", "-ffp-contract=on", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
        cmd += os.environ.get("TS_HIPCC_FLAGS", "").split()     # e.g. TS_HIPCC_FLAGS=-DTS_EXACT_SILU python -m temporalstereo_amd.build --force
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
