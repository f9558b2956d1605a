#!/usr/bin/env python3
"""Kernel time by name inside the LAST bench step of a rocprofv3 --kernel-trace CSV (window = after the last k_kmer_scan)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [s for s, e, n in ev if "k_kmer_scan" in n]
w0 = starts[-1]
acc = collections.Counter(); cnt = collections.Counter()
for s, e, n in ev:
    if s >= w0:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0][:48]
        acc[n] += e - s; cnt[n] += 1
tot = sum(acc.values())
print("window %.1f ms, kernels busy %.1f ms" % ((ev[-1][1] - w0) / 1e6, tot / 1e6))
for n, v in acc.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    print(f"{v/1e6:9.2f} ms {cnt[n]:5d}x  {n}")
