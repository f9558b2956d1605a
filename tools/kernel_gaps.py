#!/usr/bin/env python3
"""Largest idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (diagnostic for host-side stalls)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows)
t0 = ev[0][0]
starts = [s for s, e, n in ev if "k_kmer_scan" in n]
lo = (starts[-1] - t0) / 1e9                              # the last bench step
gaps = []
busy = 0
for i in range(1, len(ev)):
    if (ev[i][0] - t0) / 1e9 < lo:
        continue
    busy += ev[i][1] - ev[i][0]
    g = ev[i][0] - ev[i - 1][1]
    gaps.append((g, (ev[i - 1][1] - t0) / 1e9, ev[i - 1][2], ev[i][2]))
gaps.sort(reverse=True)
print("kernels busy %.1f ms, span %.1f ms" % (busy / 1e6, (ev[-1][1] - t0) / 1e6 - lo * 1e3))
for g, at, a, b in gaps[:25]:
    print(f"{g/1e6:9.2f} ms idle at {at:8.3f} s  after {a}  before {b}")
