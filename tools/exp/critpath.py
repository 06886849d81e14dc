"""Critical-path view of the last full pass in a rocprofv3 kernel-trace db: busy vs idle time of the union
of all streams, and per-queue busy time, between two consecutive 'copy_rows' pass starts."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, queue_id, name from kernels order by start"))
starts = [i for i, r in enumerate(rows) if "copy_rows" in r[3] and (i == 0 or "copy_rows" not in rows[i - 1][3])]
# a pass starts with the first copy_rows of a group of 4
groups = []
for i in starts:
    if not groups or rows[i][0] - rows[groups[-1]][0] > 1.0e6:
        groups.append(i)
a, b = groups[-3], groups[-2]
seg = rows[a:b]
t0, t1 = seg[0][0], rows[b][0]
print("pass of %d kernels, %.1f us start-to-start" % (len(seg), (t1 - t0) / 1e3))
ev = sorted([(s, 1) for s, e, q, n in seg] + [(e, -1) for s, e, q, n in seg])
busy, depth, last = 0, 0, t0
for tm, dlt in ev:
    if depth > 0: busy += tm - last
    depth += dlt; last = tm
print("union busy %.1f us, idle %.1f us" % (busy / 1e3, (t1 - t0 - busy) / 1e3))
per = {}
for s, e, q, n in seg: per[q] = per.get(q, 0) + e - s
print({q: round(v / 1e3, 1) for q, v in per.items()})
# gaps: idle intervals longer than 2 us
depth, last, gaps = 0, t0, []
for tm, dlt in ev:
    if depth == 0 and tm - last > 2000: gaps.append(((last - t0) / 1e3, (tm - last) / 1e3))
    depth += dlt; last = tm
print("idle gaps > 2us: %d, total %.1f us" % (len(gaps), sum(g[1] for g in gaps)))
print(" ".join("%.0f:%.1f" % g for g in gaps[:60]))
