#!/usr/bin/env python3
"""Per-kernel sum / per-dispatch mean of a PMC counter from a rocprofv3 rocpd database.
Usage: tools/rocpd_pmc.py results.db > out.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
print("# columns:", cols)
q = """select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name order by sum(value) desc"""
try:
    rows = db.execute(q).fetchall()
except Exception as e:
    print("query failed:", e)
    rows = []
print(f"{'kernel':70s} {'counter':14s} {'dispatches':>10s} {'sum':>16s} {'mean':>14s}")
for k, c, n, s, a in rows[:60]:
    print(f"{(k or '')[:70]:70s} {c:14s} {n:10d} {s:16.1f} {a:14.1f}")
