#!/usr/bin/env python
"""Per-kernel timeline of ONE forward out of a rocprofv3 --kernel-trace rocpd database: start / duration / queue of every
dispatch between two consecutive lstm_prep_kernel launches (one per forward).

    python tools/trace_timeline.py <results.db> [forward index, default: the median-length one] [--gantt]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    pick = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip("-").isdigit() else None
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select d.start, d.end, d.queue_id, d.grid_size_x, d.workgroup_size_x, s.kernel_name from %s d join %s s "
                       "on d.kernel_id = s.id order by d.start" % (disp, sym)).fetchall()
    marks = [i for i, r in enumerate(rows) if "lstm_prep_kernel" in r[5] or "lstm_pack_kernel(" in r[5]]
    spans = [(rows[marks[i + 1]][0] - rows[marks[i]][0]) / 1e3 for i in range(len(marks) - 1)]
    print("forwards:", len(marks), "spacing us:", " ".join("%.0f" % s for s in spans))
    if pick is None:
        order = sorted(range(len(spans)), key=lambda i: spans[i])
        pick = order[len(order) // 4]           # a typical graph replay (the eager forwards are the long ones)
    # a forward's kernels may start before its lstm_pack (other streams): take everything from the previous forward's
    # last kernel end
    lo, hi = marks[pick], marks[pick + 1]
    t0 = min(r[0] for r in rows[lo:hi])
    # include kernels of this forward that started on other streams before lstm_pack: look back up to 30 dispatches
    back = lo
    prev_end = max(r[1] for r in rows[max(0, lo - 200):lo]) if lo else t0
    print("forward %d: spacing %.0f us" % (pick, spans[pick]))
    print("%8s %8s  q   grid  kernel" % ("start", "dur"))
    for r in rows[max(0, lo - 12):hi]:
        name = r[5].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:58]
        print("%8.1f %8.1f  %-3d %5d  %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2], r[3] // max(r[4], 1), name))


if __name__ == "__main__":
    main()
