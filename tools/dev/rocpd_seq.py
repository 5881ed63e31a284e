#!/usr/bin/env python
"""Durations of one kernel's dispatches in launch order: python tools/dev/rocpd_seq.py <db> <name substring> [n]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute("select d.start, d.end from %s d join %s s on d.kernel_id = s.id where s.kernel_name like ? order by d.start"
                   % (disp, sym), ("%" + sys.argv[2] + "%",)).fetchall()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
prev = None
out = []
for a, b in rows[-n:]:
    out.append("%.1f(+%.1f)" % ((b - a) / 1e3, (a - prev) / 1e3 if prev else 0.0))
    prev = b
print(" ".join(out))
