"""End-to-end parity table (GPU box): for every BASELINE configuration at its stated batch and several seeds, the
full-resolution disparity of every frame of the sequence -- product path vs the fp32 CPU oracle, next to the oracle's
OWN fp32 rounding noise (fp32 oracle vs the same oracle in fp64, i.e. what ANY independent fp32 implementation of the
reference -- the reference on another BLAS included -- is away from the exact result on these random-weight networks).

    python tools/parity_report.py [--seeds 8 4 2 4] [--no-f64] > profiles/rNN_parity_end_to_end.txt
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import parity_tools as PT  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=4, default=[8, 4, 2, 4])
    ap.add_argument("--no-f64", action="store_true")
    ap.add_argument("--only", type=int, default=-1)
    a = ap.parse_args()
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    print("%-46s %9s %2s | %-32s | %-32s | %-32s" % ("configuration", "seed", "f", "ours vs oracle32: dEPE mean|d| >0.01", "oracle32 vs oracle64", "ours vs oracle64"))
    for ci, (name, n) in enumerate(zip(PT.CONFIGS, a.seeds)):
        if a.only >= 0 and ci != a.only:
            continue
        c = PT.CONFIGS[name]
        for k in range(n):
            seed = synth.SEED0 + 100 + 7 * k
            case = PT.Case(c, seed, dev)
            eng = InferenceEngine(case.net, backend="native", replay="plan")
            io32, io64, inat = {}, {}, {}
            for t in range(c["frames"]):
                o32 = case.oracle_frame(t, io32)[0]; io32 = o32[5]
                o64 = None
                if not a.no_f64:
                    o64 = case.oracle_frame(t, io64, torch.float64)[0]; io64 = o64[5]
                if t > 0:
                    inat = case.native_update(t, inat)
                on = eng(*case.frames_gpu[t], dict(inat))
                inat = PT.to_dev(on[5], dev)
                inat = {kk: (vv.clone() if torch.is_tensor(vv) else ({x: y.clone() for x, y in vv.items()} if isinstance(vv, dict) else vv)) for kk, vv in inat.items()}

                def cell(x, y):
                    d, mad = PT.delta_epe(x, y, seed + 10 * t, case.max_disp)
                    far = float(((x.detach().cpu().double() - y.detach().cpu().double()).abs() > 0.01).double().mean())
                    return "%9.2e %9.2e %8.4f%%" % (d, mad, 100 * far)
                print("%-46s %9d %2d | %s | %s | %s" % (name, seed, t, cell(on[0][0], o32[0][0]),
                                                       cell(o32[0][0], o64[0][0]) if o64 else "-", cell(on[0][0], o64[0][0]) if o64 else "-"), flush=True)


if __name__ == "__main__":
    main()
