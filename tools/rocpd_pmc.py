#!/usr/bin/env python
"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database (one --pmc pass).

    python tools/rocpd_pmc.py gpurun_out/pmc_fetch/fetch_results.db [name-substring]
Prints kernel, dispatches, mean counter value per dispatch (raw units of the counter; FETCH_SIZE/WRITE_SIZE
are in KiB-like units of 1024 B per the rocprofv3 derived-metric definition -- see MI355X_MICROARCH.md HBM).
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    q = ("select s.kernel_name, i.name, count(*), avg(p.value), d.grid_size_x, d.grid_size_y "
         "from %s p join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "join %s i on p.pmc_id = i.id where s.kernel_name like ? group by 1, 2, 5, 6 order by 4 desc"
         % (t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_info_pmc")))
    for name, ctr, n, avg, gx, gy in cur.execute(q, ("%" + sub + "%",)).fetchall():
        print("%-60s grid=(%d,%d) %s n=%d avg=%.1f" % (name[:60], gx, gy, ctr, n, avg))
        if "--split2" in sys.argv:
            # the same kernel launched on two problem sizes with one grid (e.g. the L=196 / L=100 attention launches):
            # per-dispatch sums (a dispatch has one row per XCD/SE instance for SQ counters), split at the median
            q2 = ("select sum(p.value) from %s p join %s d on p.event_id = d.event_id join %s s on d.kernel_id = s.id "
                  "where s.kernel_name = ? and d.grid_size_x = ? group by d.event_id order by 1"
                  % (t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")))
            v = [r[0] for r in cur.execute(q2, (name, gx))]
            h = len(v) // 2
            if h:
                print("    per dispatch: lower half mean %.1f, upper half mean %.1f (n=%d)" % (sum(v[:h]) / h, sum(v[h:]) / (len(v) - h), len(v)))


if __name__ == "__main__":
    main()
