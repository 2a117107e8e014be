#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average) from a rocprofv3 rocpd sqlite database.
usage: rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else None
if out:
    out.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
for n, k, t, a, mn, mx in rows:
    if out:
        out.write(f'"{n}",{k},{t},{a:.1f},{100.0 * t / tot:.2f},{mn},{mx}\n')
    short = re.sub(r'\(anonymous namespace\)::', '', n)
    short = re.sub(r'\(.*', '', short)[:90]
    print(f'{short:90s} {k:6d} {t / 1e6:9.3f} ms {a / 1e3:9.2f} us {100.0 * t / tot:6.2f}%')
print('total', tot / 1e6, 'ms')
