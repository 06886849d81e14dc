#!/usr/bin/env python
"""Summarise rocprofv3 --pmc results (.db): per kernel, mean counter value per dispatch."""
import sqlite3
import sys


def main(path, pattern="%"):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("tables:", tabs)
        return
    cols = [d[1] for d in db.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    gcols = [c for c in ("grid_size_x", "grid_size_y", "grid_size_z", "grid_size") if c in cols]
    if "--columns" in sys.argv:
        print(cols)
    gsel = ", ".join(gcols) if gcols else "0"
    q = ("select %s, counter_name, avg(value), count(*), %s from %s where %s like ? group by %s, counter_name%s "
         "order by %s, counter_name" % (kcol, gsel, view, kcol, kcol, (", " + gsel) if gcols else "", kcol))
    for row in db.execute(q, (pattern,)):
        k, c, v, n = row[:4]
        print("%-60s %-12s %16.1f  (n=%d)  grid=%s" % (k[:60], c, v, n, "x".join(str(g) for g in row[4:])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "%")
