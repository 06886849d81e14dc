"""K1 ablation: builds variants of csrc/block_cost.hip (text substitutions, nothing committed) and times the
precise-level launch of each.  usage: python tools/exp/k1_ablate.py build | run"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "temporalstereo_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "exp")
SRC = open(os.path.join(CSRC, "block_cost.hip")).read()

WARP_STORE = "              bst4<VEC>(orsrc, loff, plane, x4, W, pack(tv));                                     // warped half only"
REF_STORES = ("              bst4<VEC>(orsrc, loff, plane, x4, W, lv4);                                           // reference half\n"
              "              bst4<VEC>(orsrc, loff, plane + static_cast<unsigned>(C) * dHW, x4, W, pack(tv));    // warped half")
TAPS = ("          ta[k] = rrow[op[k] & 0xffffu];\n"
        "          if constexpr (SAMPLED) tb[k] = rrow[op[k] >> 16];")
LEFT = "          const float4 lv4 = *reinterpret_cast<const float4*>(ldsL + (c * 2 + (r & 1)) * Wl + x4);"
CORR = ("      bst4<VEC>(orsrc, loff, static_cast<unsigned>(s.mainC + g) * dHW, x4, W,\n"
        "                make_float4(-g0[0], -g0[1], -g0[2], -g0[3]));\n    }\n    if (r & 1) {   // a row pair is complete")
for pat in (WARP_STORE, REF_STORES, TAPS, LEFT, CORR):
    assert SRC.count(pat) >= 1, pat

def variant(name):
    s = SRC
    if "S" in name:   # no main-channel stores
        s = s.replace(WARP_STORE, "              g0[0] += tv[0] + tv[1] + tv[2] + tv[3];")
        s = s.replace(REF_STORES, "              g0[0] += tv[0] + tv[1] + tv[2] + tv[3] + lv[0];")
    if "T" in name:   # no tap reads from LDS
        s = s.replace(TAPS, "          ta[k] = make_float4(fr[k], dv[k], fr[k], dv[k]); tb[k] = ta[k];")
    if "L" in name:   # no left reads from LDS
        s = s.replace(LEFT, "          const float4 lv4 = make_float4(fr[0], fr[1], dv[2], dv[3] + cc);")
    if "C" in name:   # no correlation-row store (keeps a dependency through the pooled sums)
        s = s.replace(CORR, "      s1[0][0] += g0[0] + g0[1] + g0[2] + g0[3];\n    }\n    if (r & 1) {   // a row pair is complete")
    return s

NAMES = ["base", "S", "T", "L", "TL", "ST", "STL", "SC", "STLC"]

def build():
    procs = []
    for n in NAMES:
        src = os.path.join(OUT, "k1_%s.hip" % n)
        open(src, "w").write(variant(n).replace('#include "ts_common.hpp"', '#include "%s/ts_common.hpp"' % CSRC))
        so = os.path.join(OUT, "libk1_%s.so" % n)
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src,
               os.path.join(CSRC, "ts_common.hip"), "-o", so, "-ffp-contract=on", "-munsafe-fp-atomics", "-fno-slp-vectorize",
               "-Wno-unused-function"]
        procs.append((n, src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for n, src, p in procs:
        out, _ = p.communicate()
        print(n, "rc", p.returncode, out.decode()[-300:] if p.returncode else "")
        os.remove(src)

def run():
    import torch
    dev = torch.device("cuda:0")
    B, C, H, W, D = 1, 128, 136, 240, 5
    g = torch.Generator(device="cpu").manual_seed(0)
    l = torch.randn(B, C, H, W, generator=g).to(dev); r = torch.randn(B, C, H, W, generator=g).to(dev)
    d = (torch.rand(B, D, H, W, generator=g) * 40).to(dev)
    out = torch.empty(B, 2 * C + 3 * C // 8, D, H, W, device=dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for n in NAMES:
        L = ctypes.CDLL(os.path.join(OUT, "libk1_%s.so" % n))
        for fn in ("ts_block_cost_sampled_fwd", "ts_block_cost_sampled_warped_fwd"):
            f = getattr(L, fn); f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
            call = lambda: f(P(l), P(r), P(d), P(out), P(ws), B, C, H, W, D, 1, st)     # scales=1: main kernel only
            for _ in range(10): assert call() == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): call()
            e1.record(); torch.cuda.synchronize()
            print("%-6s %-34s %7.1f us" % (n, fn, e0.elapsed_time(e1) * 10), flush=True)

if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
