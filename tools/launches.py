#!/usr/bin/env python3
"""Per-launch durations of one kernel from a rocprofv3 kernel trace CSV: launches.py trace.csv k_align_wave [n_top]"""
import csv, sys
name, top = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20
d = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Start_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if name in r["Kernel_Name"]]
d.sort(key=lambda x: x[1])
t0 = d[0][1] if d else 0
tot = sum(x[0] for x in d)
print(f"{name}: {len(d)} launches, total {tot:.1f} ms, mean {tot / max(len(d), 1):.2f} ms, max {max(x[0] for x in d):.2f} ms")
print("in launch order (ms):", " ".join(f"{x[0]:.0f}" for x in d[:400]))
