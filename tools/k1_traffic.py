#!/usr/bin/env python
"""HBM traffic of the judged K1 launch from two rocprofv3 PMC passes of the BENCH command (FETCH_SIZE and WRITE_SIZE cannot share
a pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --calibrate
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o w -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --calibrate
    python tools/k1_traffic.py <fetch.db> <write.db> > profiles/rNN_k1_hbm_traffic_pmc.json

Corrections as the guide prescribes (HBM section): on gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at
64 bytes -> x2; WRITE_SIZE x1.  Both factors are re-measured in the same passes on bench.py's --calibrate streams (csrc/calib.hip:
a read, a fill and a copy of a known 1 GiB) and reported next to the result."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = [t for t in tabs if t.startswith("counters_collection")][0]
    cols = [d[1] for d in db.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, grid_size_x, grid_size_y, grid_size_z, workgroup_size_x, avg(value), count(*) from %s where counter_name = ? group by 1,2,3,4,5" % (kcol, view)
    out = {}
    for k, gx, gy, gz, wx, v, n in db.execute(q, (counter,)):
        out[(k, gx // max(wx, 1), gy, gz)] = (v, n)
    return out


def main(fetch_db, write_db):
    F = per_kernel(fetch_db, "FETCH_SIZE")
    W = per_kernel(write_db, "WRITE_SIZE")

    def find(tab, sub, grid=None):
        hits = [(k, v) for k, v in tab.items() if sub in k[0] and (grid is None or tuple(k[1:]) == tuple(grid))]
        if not hits:
            raise SystemExit("no dispatch of %s grid %s; have: %s" % (sub, grid, sorted(set(k[0][:60] for k in tab))[:40]))
        return hits[0][1][0]
    kib = 1024.0
    # calibration streams: true bytes known (bench.py --calibrate: 1 GiB each)
    true = float(1 << 30)
    cal = dict(read_fetch_over_true=find(F, "calib_read") * kib / true, fill_write_over_true=find(W, "calib_fill") * kib / true,
               copy_fetch_over_true=find(F, "calib_copy") * kib / true, copy_write_over_true=find(W, "calib_copy") * kib / true)
    fx, wx = 1.0 / cal["read_fetch_over_true"], 1.0 / cal["fill_write_over_true"]
    rows = {}
    for name, sub, grid in (("main (reference half + warped half + scale-0 correlation + pooled maps)", "block_cost_fast<true, true, 3, true, true>", (34, 16, 1)),
                            ("expansion of the pooled maps", "block_cost_upsample_rows<true, 4>", (8, 80, 1)),
                            ("round 1-3 pipeline: main without the reference half", "block_cost_fast<true, true, 3, false, true>", (34, 16, 1)),
                            ("round 4-5 pipeline: the correlation planes alone", "block_cost_corr_rows<2", (34, 16, 1))):
        f, w = find(F, sub, grid) * kib, find(W, sub, grid) * kib
        rows[name] = dict(fetch_size_bytes_raw=f, write_size_bytes_raw=w, hbm_bytes=f * fx + w * wx)
    main_b = rows["main (reference half + warped half + scale-0 correlation + pooled maps)"]["hbm_bytes"]
    up_b = rows["expansion of the pooled maps"]["hbm_bytes"]
    alg = 232527360
    print(json.dumps(dict(workload_key=[1, 128, 136, 240, 5, True],
                          launch="ts_block_cost_sampled_fwd (block_cost_fast + block_cost_upsample_rows), the launches of `python bench.py`",
                          hbm_bytes_per_launch=main_b + up_b, algorithmic_bytes=alg, ratio=(main_b + up_b) / alg,
                          correction="FETCH_SIZE x%.3f, WRITE_SIZE x%.3f (measured on the calibration streams of the same passes)" % (fx, wx),
                          calibration=cal, kernels=rows,
                          warped_variant=dict(launch="ts_block_cost_sampled_warped_fwd (rounds 1-3)",
                                              hbm_bytes_per_launch=rows["round 1-3 pipeline: main without the reference half"]["hbm_bytes"] + up_b,
                                              algorithmic_bytes=148968960),
                          pipeline_variant=dict(launch="ts_block_cost_sampled_corr_fwd (block_cost_corr_rows + expansion: what the pipeline launches)",
                                                hbm_bytes_per_launch=rows["round 4-5 pipeline: the correlation planes alone"]["hbm_bytes"] + up_b,
                                                algorithmic_bytes=65410560)), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
