#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel trace) as a per-kernel table (name, calls, avg/min/max us)."""
import sqlite3
import sys


def main(path, top=25):
    db = sqlite3.connect(path)
    rows = list(db.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(lds_size), max(workgroup_x), max(grid_x) from kernels group by name "
        "order by sum(end-start) desc limit %d" % top))
    tot = sum(r[5] for r in rows) or 1
    print("%-72s %6s %9s %9s %9s %6s %5s %7s %5s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "pct", "vgpr", "lds", "wg"))
    for r in rows:
        print("%-72s %6d %9.1f %9.1f %9.1f %6.1f %5d %7d %5d" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                             100.0 * r[5] / tot, r[6], r[7], r[8]))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
