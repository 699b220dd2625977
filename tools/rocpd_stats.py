#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel stats CSV (the `--stats` view):
name, calls, total ms, avg us, min us, max us, % of GPU kernel time.   usage: rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("""
  select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
  from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
  group by s.kernel_name order by 3 desc""").fetchall()
total = sum(r[2] for r in rows)
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n[:140], c, round(t / 1e6, 3), round(a / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * t / total, 2)])
print(f"{len(rows)} kernels, total {total/1e6:.1f} ms")
