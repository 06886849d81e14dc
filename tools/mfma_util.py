#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 --pmc results .db
(counters SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE [SQ_BUSY_CU_CYCLES]).

  util  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)     busy SIMD-cycles over available SIMD-cycles
          (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs = 1024 MFMA pipes)
  TF/s  = issued MFMA flops / kernel duration; v_mfma_f32_16x16x4_f32 holds a SIMD's pipe for 32 cycles
          (MI355X_MICROARCH.md, per-instruction constants) and does 2*16*16*4 = 2048 flops, so
          flops = BUSY/32 * 2048.  Includes the zero-padded part of a tile (Cout 8 on a 16-wide tile).
          ig_conv_x6_kernel issues v_mfma_f32_16x16x32_bf16 (16 cycles, 16384 flops) and spends six of them, on ten tap slots
          for nine taps, per fp32 product block: its TF/s column is the fp32-EQUIVALENT rate, BUSY/16 * 16384 / 6 * 0.9.
usage: mfma_util.py results.db [name-pattern]
"""
import sqlite3
import sys


def main(path, pattern="%ig_conv%"):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select dispatch_id, kernel_name, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, counter_name, value, duration "
        "from counters_collection where kernel_name like ?", (pattern,)).fetchall()
    disp = {}
    for did, name, gx, gy, gz, wx, cname, val, dur in rows:
        d = disp.setdefault(did, dict(name=name, grid=(gx // max(wx, 1), gy, gz), dur=dur, c={}))
        d["c"][cname] = d["c"].get(cname, 0.0) + val
    groups = {}
    for d in disp.values():
        short = d["name"].replace("void (anonymous namespace)::", "").split("(")[0]
        g = groups.setdefault((short, d["grid"]), [])
        g.append(d)
    out = []
    for (name, grid), ds in groups.items():
        n = len(ds)
        busy = sum(d["c"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in ds) / n
        gui = sum(d["c"].get("GRBM_GUI_ACTIVE", 0.0) for d in ds) / n
        mops = sum(d["c"].get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) for d in ds) / n
        dur = sum(d["dur"] for d in ds) / n
        if gui <= 0:
            continue
        util = busy / (gui / 8.0 * 1024.0)
        flops = busy / 32.0 * 2048.0
        if "x6p" in name:       # v_mfma_f32_32x32x16_bf16, six products per fp32 product, nine taps in nine steps
            flops = busy / 32.0 * 32768.0 / 6.0
        elif "x6" in name:      # v_mfma_f32_16x16x32_bf16, ten tap slots for nine taps
            flops = busy / 16.0 * 16384.0 / 6.0 * 0.9
        out.append((busy * n, name, grid, n, dur / 1e3, util, flops / dur / 1e3, mops))
    out.sort(reverse=True)
    tot_busy = sum(o[0] for o in out)
    # MfmaUtil is a share of the WHOLE chip's matrix pipes.  A persistent kernel (ig_conv_x6p_kernel: one workgroup per CU, at most 208 of
    # them, fewer when the rounds divide evenly) leaves CUs to the pass's other streams on purpose: the last column is the same share over
    # the CUs it occupies (grid <= 256 workgroups of a one-per-CU kernel).
    print("%-34s %-14s %5s %9s %8s %8s %12s" % ("kernel", "grid", "calls", "avg_us", "MfmaUtil", "TF/s", "of its CUs"))
    for _, name, grid, n, us, util, tfs, mops in out[:40]:
        own = ("%10.1f%%" % (100 * util * 256.0 / grid[0])) if ("x6p" in name and grid[0] <= 256 and grid[1] == 1 and grid[2] == 1) else ""
        print("%-34s %-14s %5d %9.1f %7.1f%% %8.1f %12s" % (name[:34], "%dx%dx%d" % grid, n, us, 100 * util, tfs, own))
    wsum = sum(o[5] * o[4] * o[3] for o in out)
    tsum = sum(o[4] * o[3] for o in out)
    print("time-weighted MfmaUtil over %d dispatches of %d kernels: %.1f%%  (f32 MFMA peak 157.3 TF/s)" %
          (sum(o[3] for o in out), len(out), 100 * wsum / max(tsum, 1e-9)))
    # the two arithmetic families apart: kernels on the bf16 pipe (x6 / x6s: six bf16 MFMAs per fp32 product
    # block) and kernels on the f32-input MFMA
    for label, sel in (("bf16-split kernels (ig_conv_x6*)", lambda n: "x6" in n), ("f32-input MFMA kernels", lambda n: "x6" not in n)):
        part = [o for o in out if sel(o[1])]
        w, t = sum(o[5] * o[4] * o[3] for o in part), sum(o[4] * o[3] for o in part)
        print("  %-34s %4d dispatches  %9.1f us  time-weighted MfmaUtil %.1f%%" % (label, sum(o[3] for o in part), t, 100 * w / max(t, 1e-9)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "%ig_conv%")
