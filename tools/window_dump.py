#!/usr/bin/env python3
"""Every dispatch longer than `min_ms` inside a window of the LAST bench step of a rocprofv3 --kernel-trace CSV, in start order, with
its queue: the timeline one reads to see which chain an encode lane is waiting for.  Usage: window_dump.py trace.csv [t0_s [len_s [min_ms]]]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:], r.get("Queue_Id", "?")))
rows.sort()
starts = [r[0] for r in rows if "k_kmer_scan" in r[2]]
lo = starts[-1]
for s in reversed(starts):
    if lo - s > 2_000_000_000:
        break
    lo = s
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
ln = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
mn = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
a, b = lo + int(t0 * 1e9), lo + int((t0 + ln) * 1e9)
qs = {}
for s, e, n, q in rows:
    if e < a or s > b or (e - s) < mn * 1e6:
        continue
    qi = qs.setdefault(q, len(qs))
    print(f"{(s - lo) / 1e6:10.2f} {(e - s) / 1e6:8.2f} ms  q{qi:<3d} {n}")
