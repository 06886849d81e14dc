#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel trace) as a per-kernel table.

usage: prof_summary.py results.db [top] [--by-grid | --by-family] [--match SUBSTRING] [--window-ms A B]   (window relative to the LAST kernel end,
       e.g. --window-ms 60 20 keeps kernels that started between 60 and 20 ms before the end)
"""
import sqlite3
import sys


def families(path, window=None):
    """Busy time per kernel family (template arguments and argument lists stripped)."""
    import re
    db = sqlite3.connect(path)
    where = ""
    if window:
        tend = db.execute("select max(end) from kernels").fetchone()[0]
        where = " where start >= %d and start <= %d" % (tend - int(window[0] * 1e6), tend - int(window[1] * 1e6))
    fam = {}
    for name, dur in db.execute("select name, end-start from kernels" + where):
        n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        n = re.sub(r"<.*", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        f = fam.setdefault(n, [0, 0])
        f[0] += 1
        f[1] += dur
    tot = sum(v[1] for v in fam.values()) or 1
    print("%-60s %8s %12s %6s" % ("kernel family", "calls", "busy_ms", "pct"))
    for n, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-60s %8d %12.3f %6.1f" % (n[:60], c, d / 1e6, 100.0 * d / tot))
    print("total busy %.3f ms" % (tot / 1e6))


def main(path, top=25, window=None, by_grid=False, match=None):
    db = sqlite3.connect(path)
    where = ""
    if window:
        tend = db.execute("select max(end) from kernels").fetchone()[0]
        where = " where start >= %d and start <= %d" % (tend - int(window[0] * 1e6), tend - int(window[1] * 1e6))
    rows = list(db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(lds_size), max(workgroup_x), max(grid_x), max(grid_y), max(grid_z) from kernels" + where +
        (" group by name, grid_x, grid_y, grid_z " if by_grid else " group by name ") +
        "order by sum(end-start) desc"))
    tot = sum(r[5] for r in rows) or 1
    if match:           # full names of the kernels containing `match`
        for r in [r for r in rows if match in r[0]][:top]:
            print("%6d calls  %8.1f us avg  %5.1f %%  g=%dx%dx%d  %s" % (r[1], r[2] / 1e3, 100.0 * r[5] / tot, r[9] // max(r[8], 1), r[10], r[11], r[0][:400]))
        return
    span = db.execute("select min(start), max(end), count(*) from kernels" + where).fetchone()
    print("kernels: %d dispatches, %d distinct, busy %.3f ms over a %.3f ms span" % (span[2], len(rows), tot / 1e6, (span[1] - span[0]) / 1e6))
    print("%-72s %6s %9s %9s %9s %6s %5s %7s %5s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds", "wg"))
    for r in rows[:top]:
        name = r[0][:72]
        if by_grid:
            name = (r[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]
                    + " g=%dx%dx%d" % (r[9] // max(r[8], 1), r[10], r[11]))[:72]
        print("%-72s %6d %9.1f %9.1f %9.1f %6.1f %5d %7d %5d" % (name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                             100.0 * r[5] / tot, r[6], r[7], r[8]))


if __name__ == "__main__":
    args = sys.argv[1:]
    window = None
    if "--window-ms" in args:
        i = args.index("--window-ms")
        window = (float(args[i + 1]), float(args[i + 2]))
        args = args[:i] + args[i + 3:]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        args = args[:i] + args[i + 2:]
    by_grid = "--by-grid" in args
    fam = "--by-family" in args
    args = [a for a in args if a not in ("--by-grid", "--by-family")]
    if fam:
        families(args[0], window)
    else:
        main(args[0], int(args[1]) if len(args) > 1 else 25, window, by_grid, match)
