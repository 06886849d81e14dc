"""Union vs sum of kernel intervals in a rocprofv3 kernel-trace db (does multi-stream overlap happen?)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tend = db.execute("select max(end) from kernels").fetchone()[0]
a, b = tend - int(float(sys.argv[2]) * 1e6), tend - int(float(sys.argv[3]) * 1e6)
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute("select start, end, name%s from kernels where start >= %d and start <= %d order by start" % (", " + qcol if qcol else "", a, b)))
tot = sum(r[1] - r[0] for r in rows)
union, cur_s, cur_e = 0, None, None
for r in rows:
    s, e = r[0], r[1]
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("kernels %d  sum %.3f ms  union %.3f ms  span %.3f ms  columns %s" % (len(rows), tot / 1e6, union / 1e6, (rows[-1][1] - rows[0][0]) / 1e6, cols))
if qcol:
    qs = {}
    for r in rows: qs.setdefault(r[3], [0, 0]); qs[r[3]][0] += 1; qs[r[3]][1] += r[1] - r[0]
    print({k: (v[0], round(v[1] / 1e6, 3)) for k, v in qs.items()})
