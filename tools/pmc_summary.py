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
    q = ("select %s, counter_name, avg(value), count(*) from %s where %s like ? group by %s, counter_name "
         "order by %s, counter_name" % (kcol, view, kcol, kcol, kcol))
    for k, c, v, n in db.execute(q, (pattern,)):
        print("%-60s %-28s %16.1f  (n=%d)" % (k[:60], c, v, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "%")
