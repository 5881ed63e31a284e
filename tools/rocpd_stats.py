#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(the same columns as `rocprofv3 --stats`): calls, total/avg/min/max duration, % of GPU time.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--md profiles/x.md]
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.kernel_name "
        "order by 3 desc" % (disp, sym)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |"
                     % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    out = "\n".join(lines)
    print(out)
    if md:
        with open(md, "a") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
