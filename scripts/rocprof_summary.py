#!/usr/bin/env python3
"""Turns a rocprofv3 `--kernel-trace --stats` result database (rocpd .db) into the per-kernel
summary table committed under profiles/.  Usage: rocprof_summary.py <results.db> [min_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                  "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) "
                  "from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>6s} {'total_us':>11s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} "
      f"{'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>6s}")
for r in rows:
    name = r[0].split("(")[0].replace("void ", "")[:64]
    print(f"{name:64s} {r[1]:6d} {r[2]:11.1f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:9.2f} {100 * r[2] / total:6.2f} "
          f"{r[6]:5d} {r[7]:5d} {r[8]:5d} {r[9]:6d}")
