"""Timeline of the LAST pass in a rocprofv3 kernel-trace .db of `bench.py --frames-in-flight 1`: per hardware queue, every kernel with its start
(relative to the pass), duration and the gap to its predecessor on the same queue.  usage: pass_timeline.py results.db [nth pass]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# a pass ends with the full-resolution upsampler: the window is (end of its (nth-1)-th launch, end of its nth launch]
ends = [r[0] for r in db.execute("select end from kernels where name like '%unet_upsample_kernel%' order by end")]
t0, t1 = ends[nth - 2], ends[nth - 1]
rows = list(db.execute("select queue_id, start, end, name, grid_x, grid_y, grid_z, workgroup_x from kernels where end > ? and end <= ? order by start", (t0, t1)))
base = rows[0][1]
last = {}
busy = {}
for q, s, e, name, gx, gy, gz, wx in rows:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    busy[q] = busy.get(q, 0.0) + (e - s) / 1e3
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:46]
    print("q%-2s t=%8.1f  dur %6.1f  gap %6.1f  g=%dx%dx%d  %s" % (q, (s - base) / 1e3, (e - s) / 1e3, gap, gx // max(wx, 1), gy, gz, short))
print("busy per queue (us):", {k: round(v, 1) for k, v in busy.items()}, " span %.1f us" % ((max(r[2] for r in rows) - base) / 1e3))
