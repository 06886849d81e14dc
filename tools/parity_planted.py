#!/usr/bin/env python
"""End-to-end parity on planted-disparity scenes with the trained checkpoint (GPU box): per configuration, seed and frame,
EPE against the planted ground truth of the product path (native engine + update_map), of the CPU oracle in fp32 and in fp64,
and the |dEPE| between them.  Usage: python tools/parity_planted.py [--ckpt PATH] [--seeds N] [--configs 1,2,3,4] [--out FILE]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import parity_tools as PT  # noqa: E402
import synth  # noqa: E402


def clone_info(info):
    return {k: (v.clone() if torch.is_tensor(v) else ({a: b.clone() for a, b in v.items()} if isinstance(v, dict) else v)) for k, v in info.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--configs", default="1,2,3,4")
    ap.add_argument("--fp64", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_planted.json"))
    a = ap.parse_args()
    from temporalstereo_amd.aggregation.engine import InferenceEngine
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    names = list(PT.CONFIGS)
    rows = []
    for ci in (int(x) for x in a.configs.split(",")):
        name = names[ci - 1]
        c = PT.CONFIGS[name]
        for k in range(a.seeds):
            seed = synth.SEED0 + 100 + 7 * k
            case = PT.PlantedCase(c, seed, dev, a.ckpt)
            eng = InferenceEngine(case.net, backend="native", replay="plan")
            io32, io64, inat = {}, {}, {}
            for t in range(c["frames"]):
                o32 = case.oracle_frame(t, io32)[0]; io32 = o32[5]
                if t > 0:
                    inat = case.native_update(t, inat)
                on = eng(*case.frames_gpu[t], dict(inat))
                inat = clone_info(on[5])
                e_n, e_32 = PT.epe(on[0][0], case.gt[t], case.max_disp), PT.epe(o32[0][0], case.gt[t], case.max_disp)
                row = dict(config=name, seed=seed, frame=t, epe_native=e_n, epe_oracle_fp32=e_32, delta_epe=abs(e_n - e_32),
                           mean_abs=float((on[0][0].cpu().double() - o32[0][0].double()).abs().mean()),
                           max_abs=float((on[0][0].cpu().double() - o32[0][0].double()).abs().max()))
                if a.fp64:
                    o64 = case.oracle_frame(t, io64, torch.float64)[0]; io64 = o64[5]
                    e_64 = PT.epe(o64[0][0], case.gt[t], case.max_disp)
                    row.update(epe_oracle_fp64=e_64, oracle_fp32_vs_fp64=abs(e_32 - e_64), native_vs_fp64=abs(e_n - e_64))
                rows.append(row)
                print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
