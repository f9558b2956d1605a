#!/usr/bin/env python3
"""Where an encode lane's main stream stands still: in the LAST bench step of a rocprofv3 --kernel-trace CSV, the queue with the most
k_match time; its idle gaps summed by (kernel before, kernel after), and its busy time by kernel.  Usage: lane_gaps.py trace.csv"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:], r.get("Queue_Id", "?")))
rows.sort()
starts = [r[0] for r in rows if "k_kmer_scan" in r[2]]
lo = starts[-1]
for s in reversed(starts):
    if lo - s > 2_000_000_000:
        break
    lo = s
step = [r for r in rows if r[0] >= lo]
byq = defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))
lane = max(byq, key=lambda q: sum(e - s for s, e, n in byq[q] if n == "k_match"))
ev = byq[lane]
busy = defaultdict(float); cnt = defaultdict(int); gaps = defaultdict(float); gcnt = defaultdict(int)
end = ev[0][1]
prev = ev[0][2]
for s, e, n in ev:
    busy[n] += e - s; cnt[n] += 1
for (s0, e0, n0), (s1, e1, n1) in zip(ev[:-1], ev[1:]):
    if s1 > e0:
        gaps[(n0, n1)] += s1 - e0; gcnt[(n0, n1)] += 1
tb, tg = sum(busy.values()), sum(gaps.values())
print(f"queue {lane}: {len(ev)} dispatches, busy {tb / 1e6:.0f} ms, idle between dispatches {tg / 1e6:.0f} ms, span {(ev[-1][1] - ev[0][0]) / 1e6:.0f} ms")
print("busy by kernel:")
for n, v in sorted(busy.items(), key=lambda kv: -kv[1])[:30]:
    print(f"  {n:46s} {v / 1e6:9.1f} ms  {cnt[n]:6d} x")
print("idle by (before -> after):")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:45]:
    print(f"  {k[0]:>44s} -> {k[1]:44s} {v / 1e6:8.1f} ms  {gcnt[k]:5d} x  ({v / gcnt[k] / 1e3:7.0f} us each)")
