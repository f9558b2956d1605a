#!/usr/bin/env python3
"""Per hardware queue (= the streams of the pipeline: the coders' context, the quality context, the encode lanes and their side
streams) of the LAST bench step of a rocprofv3 --kernel-trace CSV: busy time, first/last kernel, the kernels that fill it.
Shows which chain bounds the step.  Usage: tools/queue_busy.py kernel_trace.csv"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:], r.get("Queue_Id", "?")))
rows.sort()
starts = [r[0] for r in rows if "k_kmer_scan" in r[2]]
# the last step starts at the first k_kmer_scan of the last group of scans (one per chunk, back to back)
lo = starts[-1]
for s in reversed(starts):
    if lo - s > 2_000_000_000:
        break
    lo = s
step = [r for r in rows if r[0] >= lo]
hi = max(r[1] for r in step)
print(f"last step: {(hi - lo) / 1e6:.1f} ms, {len(step)} dispatches")
byq = defaultdict(list)
for s, e, n, q in step:
    byq[q].append((s, e, n))
for q, ev in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in ev)
    per = defaultdict(float); cnt = defaultdict(int)
    for s, e, n in ev:
        per[n] += e - s; cnt[n] += 1
    top = sorted(per.items(), key=lambda kv: -kv[1])[:8]
    print(f"queue {q}: busy {busy / 1e6:9.1f} ms  from {(ev[0][0] - lo) / 1e6:8.1f} to {(max(e for _, e, _ in ev) - lo) / 1e6:8.1f} ms  {len(ev)} dispatches")
    print("     " + "; ".join(f"{n} {v / 1e6:.0f} ms/{cnt[n]}" for n, v in top))
