#!/usr/bin/env python
"""Who owns the chip's time when several streams' kernels overlap: a rocprofv3 kernel trace (.db) as CU-weighted shares.

A dispatch's DEMAND is the number of CUs its grid could occupy, min(256, workgroups x workgroup size / 512) (a 512-thread workgroup of
the ping-pong kernel owns a CU; 256-thread workgroups are counted at two per CU: a rough model, stated, not measured).  Over every
interval between two dispatch boundaries the running dispatches share the 256 CUs in proportion to their demand when the demands
exceed the chip, and leave the rest IDLE when they do not.  Output: share of the window's chip-time per kernel (by grid), the idle
share, and the share of the time with nothing running at all.

usage: timeline_share.py results.db [top] [--window-ms A B]     (window relative to the last kernel end, as prof_summary.py)
"""
import sqlite3
import sys

CUS = 256.0


def main(path, top, window):
    db = sqlite3.connect(path)
    where = ""
    if window:
        tend = db.execute("select max(end) from kernels").fetchone()[0]
        where = " where start >= %d and start <= %d" % (tend - int(window[0] * 1e6), tend - int(window[1] * 1e6))
    rows = list(db.execute("select name, start, end, workgroup_x * workgroup_y * workgroup_z, grid_x, grid_y, grid_z, workgroup_x, workgroup_y, workgroup_z from kernels" + where))
    ev = []
    info = []
    for i, (name, s, e, wgsz, gx, gy, gz, wx, wy, wz) in enumerate(rows):
        nwg = (gx // max(wx, 1)) * (gy // max(wy, 1)) * (gz // max(wz, 1))
        demand = min(CUS, nwg * wgsz / 512.0)
        key = (name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44] +
               " g=%dx%dx%d" % (gx // max(wx, 1), gy // max(wy, 1), gz // max(wz, 1)))
        info.append((key, demand, e - s))
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    t0, t1 = ev[0][0], ev[-1][0]
    running = {}
    share, alone, idle, empty = {}, {}, 0.0, 0.0
    prev = t0
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            tot = sum(running.values())
            if not running:
                empty += dt
                idle += dt
            else:
                scale = 1.0 if tot <= CUS else CUS / tot
                for j, d in running.items():
                    share[info[j][0]] = share.get(info[j][0], 0.0) + dt * d * scale / CUS
                idle += dt * max(0.0, 1.0 - tot / CUS)
        prev = t
        if kind:
            running[i] = info[i][1]
        else:
            running.pop(i, None)
    for key, d, dur in info:
        a = alone.setdefault(key, [0, 0.0, d])
        a[0] += 1; a[1] += dur
    span = float(t1 - t0)
    print("window %.3f ms, %d dispatches; chip-time: %.1f %% idle CUs (of which %.1f %% with nothing running at all)" %
          (span / 1e6, len(rows), 100 * idle / span, 100 * empty / span))
    print("%-72s %6s %8s %8s %8s" % ("kernel", "calls", "avg_us", "demand", "share %"))
    for key, v in sorted(share.items(), key=lambda kv: -kv[1])[:top]:
        c, d, dem = alone[key]
        print("%-72s %6d %8.1f %8.0f %8.2f" % (key[:72], c, d / c / 1e3, dem, 100 * v / span))


if __name__ == "__main__":
    args = sys.argv[1:]
    window = None
    if "--window-ms" in args:
        i = args.index("--window-ms")
        window = (float(args[i + 1]), float(args[i + 2]))
        args = args[:i] + args[i + 3:]
    main(args[0], int(args[1]) if len(args) > 1 else 40, window)
