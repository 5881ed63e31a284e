#!/usr/bin/env python
"""CU-time accounting of the steady-state graph replays in a rocprofv3 kernel trace (rocpd database): per kernel the sum of
duration x min(workgroups, CUs) over the window, as a share of CUs x wall time.  usage: cu_time.py <results.db> [n_cu]"""
import sqlite3
import sys

db, ncu = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t = lambda p: [x for x in tabs if x.startswith(p)][0]
disp, sym = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
rows = cur.execute("select s.kernel_name, d.start, d.end, d.grid_size_x * d.grid_size_y * d.grid_size_z, "
                   "d.workgroup_size_x * d.workgroup_size_y * d.workgroup_size_z from %s d join %s s on d.kernel_id = s.id order by d.start"
                   % (disp, sym)).fetchall()
marks = [r[1] for r in rows if "lstm_prep_kernel" in r[0]]
# steady state: forwards 12 .. 28 of the timed replays
lo, hi = marks[12], marks[28]
nf = 16
agg = {}
for name, s, e, grid, wg in rows:
    if s < lo or s >= hi:
        continue
    wgs = max(1, grid // max(wg, 1))
    k = name.split("(")[0]
    k = k[k.find("N_1") + 4:][:40] if "_GLOBAL__N_1" in k else k[:40]
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += (e - s) / 1e3 * min(wgs, ncu)
wall = (hi - lo) / 1e3
print("window %.1f us, %d forwards, %.1f us per forward" % (wall, nf, wall / nf))
tot = sum(a[2] for a in agg.values())
print("sum of kernel CU time: %.1f k CU.us per forward = %.1f %% of %d CUs x wall" % (tot / nf / 1e3, 100 * tot / (ncu * wall), ncu))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print("%-42s %5.1f launches/fwd  %7.1f us/fwd  %7.2f k CU.us/fwd  %5.1f %%" % (k, a[0] / nf, a[1] / nf, a[2] / nf / 1e3, 100 * a[2] / (ncu * wall)))
