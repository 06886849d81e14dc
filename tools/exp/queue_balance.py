"""Busy time per hardware queue (= per stream of the pipelined engine) from a rocprofv3 kernel-trace .db: is the three-stage
pipeline balanced?  usage: queue_balance.py results.db [window_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = sys.argv[3] if len(sys.argv) > 3 else ("queue_id" if "queue_id" in cols else None)
tend = db.execute("select max(end) from kernels").fetchone()[0]
t0 = tend - int(win * 1e6)
print("columns:", cols)
if qcol:
    rows = list(db.execute("select %s, count(*), sum(end-start), min(start), max(end) from kernels where start >= %d group by %s order by sum(end-start) desc" % (qcol, t0, qcol)))
    for q, n, busy, a, b in rows:
        print("queue %s: %6d kernels, busy %.3f ms of %.3f ms (%.0f %%)" % (q, n, busy / 1e6, (b - a) / 1e6, 100.0 * busy / max(b - a, 1)))
    for q0 in [r[0] for r in rows]:
        print("queue %s, by kernel and grid:" % q0)
        for name, gx, gy, gz, wx, n, tot in db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(end-start) from kernels where start >= %d and %s = ? "
                                                        "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit 16" % (t0, qcol), (q0,)):
            short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
            print("  %6d  %8.3f ms  g=%dx%dx%d  %s" % (n, tot / 1e6, gx // max(wx, 1), gy, gz, short))
