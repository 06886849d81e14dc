#!/usr/bin/env python
"""Text summary of gpurun_out/parity_stagewise.json (written by tests/test_fullsize_gpu.py): worst case per stage over the
configurations and frames, the precise-level K1 against the fp64 oracle, and the near-tie audit per level.
usage: parity_stagewise_summary.py [gpurun_out/parity_stagewise.json] > profiles/rNN_parity_stagewise.txt"""
import json
import sys


def main(path="gpurun_out/parity_stagewise.json"):
    rows = json.load(open(path))
    print("# tests/test_fullsize_gpu.py::test_every_stage_and_level_teacher_forced_at_stated_batch on 1xMI355X (last full run of the GPU suite)")
    print("# per-op teacher forcing: worst case over the four BASELINE configurations (stated batches) and all frames of their sequences")
    print("%-42s %10s %10s %10s   %s" % ("stage (fed the ORACLE's input)", "max|diff|", "mean|diff|", "max|ref|", "bar"))
    order, worst = [], {}
    for r in rows:
        if "max_abs" not in r:
            continue
        w = r["what"]
        if w not in worst:
            order.append(w)
            worst[w] = dict(r)
        else:
            o = worst[w]
            o["max_abs"] = max(o["max_abs"], r["max_abs"]); o["mean_abs"] = max(o["mean_abs"], r["mean_abs"])
            o["ref_max"] = max(o["ref_max"], r["ref_max"])
    for w in order:
        o = worst[w]
        print("%-42s %10.3g %10.3g %10.3g   atol %.0e + rtol %.0e" % (w, o["max_abs"], o["mean_abs"], o["ref_max"], o["atol"], o["rtol"]))
    print()
    for r in rows:
        if "ours_max_abs" in r:
            print("%s  [%s f%d]: ours %.3g from exact, the fp32 oracle %.3g" % (r["what"], r["config"].split()[0], r["frame"], r["ours_max_abs"], r["oracle_fp32_max_abs"]))
    print()
    print("# per-level teacher forcing: pixels whose low-resolution disparity moved by > 1e-3 px, and how many of them are NOT oracle near-ties")
    print("%-44s %-12s %2s %8s %6s %11s %10s %10s %10s" % ("level", "config", "f", "pixels", "moved", "unexplained", "costerrmax", "costerravg", "still_max"))
    for r in rows:
        if "pixels" in r:
            print("%-44s %-12s %2d %8d %6d %11d %10.3g %10.3g %10.3g" % (r["what"], r["config"].split()[0], r["frame"], r["pixels"], r["moved"], r["unexplained"],
                                                                       r["cost_err_max"], r["cost_err_mean"], r["still_max"]))


if __name__ == "__main__":
    main(*sys.argv[1:])
