"""Textual timeline of the last full pass in a rocprofv3 kernel-trace db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
back_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
span_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 2.2
tend = db.execute("select max(end) from kernels").fetchone()[0]
a = tend - int(back_ms * 1e6)
rows = list(db.execute("select start, end, queue_id, name, grid_x, grid_y, grid_z, workgroup_x from kernels where start >= %d and start <= %d order by start" % (a, a + int(span_ms * 1e6))))
# start at the first block_cost_fast<false (coarse level) to align on a pass
i0 = next((i for i, r in enumerate(rows) if "copy_rows" in r[3]), 0)
t0 = rows[i0][0]
for s, e, q, n, gx, gy, gz, wx in rows[i0:]:
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    print("%8.1f %7.1f q%d %-40s g=%dx%dx%d" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, gx // max(wx, 1), gy, gz))
