#!/usr/bin/env python3
"""Longest HIP API calls of a rocprofv3 --hip-trace CSV (diagnostic for host-side overheads)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
names = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None
out = []
for r in rows:
    fn = r.get("Function") or r.get("Name")
    if names and fn not in names:
        continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out.append((d, fn, r["Start_Timestamp"]))
out.sort(reverse=True)
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for d, fn, st in out[:40]:
    print(f"{d/1e6:10.2f} ms  {fn:28s} at {(int(st)-t0)/1e9:8.3f} s")
