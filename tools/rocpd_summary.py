#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace --stats) as a text table:
per kernel: calls, total ms, average us, min/max us, % of GPU kernel time, VGPRs, LDS.
Usage: tools/rocpd_summary.py results.db > profiles/<name>.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("""select name, count(*), sum(duration), avg(duration), min(duration), max(duration),
                            max(vgpr_count), max(lds_size), max(workgroup_x), max(grid_x)
                     from kernels group by name order by sum(duration) desc""").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s} {'vgpr':>5s} {'lds':>6s} {'wg':>4s}")
for n, c, s, a, mn, mx, v, l, wg, g in rows:
    print(f"{n[:70]:70s} {c:7d} {s/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f} {v or 0:5d} {l or 0:6d} {wg or 0:4d}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
