"""Timeline of the LAST complete pass in a rocprofv3 kernel-trace db, per queue: offset, duration, gap to the previous
kernel of the same queue.  A pass is delimited by the coarse-level cost volume kernel (block_cost_fast<false,...)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end, queue_id, name, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "block_cost_fast<false" in r[3]]
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if 1.0e6 < rows[marks[i + 1]][0] - rows[marks[i]][0] < 3.0e6]
a, b = pairs[len(pairs) // 2]
t0 = rows[a][0]
# the pass also owns kernels of other queues that started shortly before its marker (wide stream): take by time window
tend = rows[b][0]
seg = [r for r in rows if t0 - 30000 <= r[0] < tend - 30000]
print("pass window %.1f us, %d kernels" % ((tend - t0) / 1e3, len(seg)))
queues = {}
for r in seg: queues.setdefault(r[2], []).append(r)
for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[1] - r[0] for r in rs)
    print("== queue %s: %d kernels, busy %.1f us, span %.1f .. %.1f us" % (q, len(rs), busy / 1e3, (rs[0][0] - t0) / 1e3, (rs[-1][1] - t0) / 1e3))
    if len(sys.argv) > 2 and sys.argv[2] == "full":
        prev = None
        for s, e, _, n, gx, gy, gz, wx in rs:
            nm = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
            print("  %8.1f  dur %6.1f  gap %6.1f  %s g=%dx%dx%d" % ((s - t0) / 1e3, (e - s) / 1e3, ((s - prev) / 1e3) if prev else 0.0, nm, gx // max(wx, 1), gy, gz))
            prev = e
