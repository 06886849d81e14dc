#!/usr/bin/env python
"""Text summary of gpurun_out/parity_end_to_end_planted.json (+ parity_train_grads.json, parity_backward_stagewise.json), written by
tests/test_fullsize_gpu.py / test_train_grads_gpu.py / test_backward_stagewise_gpu.py on the GPU box.
usage: parity_planted_summary.py [dir=gpurun_out] > profiles/rNN_parity_planted.txt"""
import json
import os
import sys


def main(d="gpurun_out"):
    rows = json.load(open(os.path.join(d, "parity_end_to_end_planted.json")))
    # a run of the suite appends; keep the last occurrence of every (what, fixture/config, seed, frame)
    last = {}
    for r in rows:
        last[(r["what"], r.get("fixture"), r.get("config"), r.get("seed"), r.get("frame"))] = r
    rows = list(last.values())
    print("# end to end on planted-disparity scenes with the trained checkpoint (tests/golden/ckpt_planted.npz), 1xMI355X, native engine + update_map")
    print("# EPE = mean |d - planted ground truth| over all valid pixels; bar |dEPE| < 1e-3 px, asserted per fixture / seed / FRAME")
    print("\n## against the imported REFERENCE's fixtures (tests/golden/planted_*.npz: reference aggregator + its own update_map, stated batches)")
    print("%-16s %-46s %5s %10s %10s %10s %10s %10s" % ("fixture", "configuration", "frame", "EPE ours", "EPE ref", "|dEPE|", "mean|d|", "max|d|"))
    for r in rows:
        if r["what"] == "end to end vs reference fixture":
            print("%-16s %-46s %5d %10.6f %10.6f %10.2e %10.2e %10.2e" % (r["fixture"], r["config"][:46], r["frame"], r["epe"], r["epe_reference"],
                                                                          r["delta_epe"], r["mean_abs"], r["max_abs"]))
    print("\n## against the CPU oracle (fp32) carrying its own state, further seeds")
    print("%-46s %10s %5s %10s %10s %10s %10s %10s" % ("configuration", "seed", "frame", "EPE ours", "EPE oracle", "|dEPE|", "mean|d|", "max|d|"))
    worst = 0.0
    for r in rows:
        if r["what"].startswith("end to end vs oracle"):
            worst = max(worst, r["delta_epe"])
            print("%-46s %10d %5d %10.6f %10.6f %10.2e %10.2e %10.2e" % (r["config"][:46], r["seed"], r["frame"], r["epe"], r["epe_oracle"], r["delta_epe"],
                                                                         r["mean_abs"], r["max_abs"]))
    print("worst |dEPE| over all seeds and frames: %.2e px" % worst)
    p = os.path.join(d, "parity_train_grads.json")
    if os.path.exists(p):
        g = json.load(open(p))
        g = g[-(7 + 6 + 41 + 273):]
        print("\n## composed backward of the whole aggregator against the reference's autograd (tests/golden/planted_train_grads.npz)")
        print("loss terms: max relative difference %.2e" % max(r["rel"] for r in g if r["what"] == "loss term"))
        f = [r for r in g if r["what"] == "feature gradient"]
        print("feature gradients (6 maps): max relative L2 %.2e, max element %.2e" % (max(r["rel_l2"] for r in f), max(r["rel_max"] for r in f)))
        w = [r for r in g if r["what"] == "weight gradient" and r["rel_l2"] < 0.5]
        print("named weight gradients (%d with a non-zero exact gradient): max relative L2 %.2e" % (len(w), max(r["rel_l2"] for r in w)))
        a = [r for r in g if r["what"].startswith("every parameter") and r["rel_norm"] < 0.2]
        print("all parameters (%d with a non-zero exact gradient): max |norm - ref| / ref %.2e, max |projection - ref| / norm %.2e"
              % (len(a), max(r["rel_norm"] for r in a), max(r["rel_proj"] for r in a)))
    p = os.path.join(d, "parity_backward_stagewise.json")
    if os.path.exists(p):
        b = json.load(open(p))
        n = len(b)
        # last run only: the file is appended to
        stages = {}
        for r in b:
            if r["what"] != "stage forward" and not r.get("exactly_zero"):
                stages[(r["stage"], r.get("key", r.get("input")))] = r["rel_l2"]
        print("\n## stage-wise teacher-forced backward (oracle in float64 as arbiter): %d gradients, worst relative L2 %.2e" % (len(stages), max(stages.values())))
        by = {}
        for (st, _), v in stages.items():
            by[st] = max(by.get(st, 0.0), v)
        for st, v in by.items():
            print("  %-44s %.2e" % (st, v))


if __name__ == "__main__":
    main(*sys.argv[1:])
