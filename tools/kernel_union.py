#!/usr/bin/env python3
"""Union coverage of all kernels of the LAST bench step of a rocprofv3 --kernel-trace CSV: busy time of the device
(any stream), idle time, and the largest idle gaps with the kernels around them."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_kmer_scan" in r[2]]
lo = rows[starts[-1]][0]
step = [r for r in rows if r[0] >= lo]
hi = max(r[1] for r in step)
busy = 0; cur_s, cur_e = step[0][0], step[0][1]; gaps = []
last_name = step[0][2]
for s, e, n in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last_name, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:40]
print(f"window {(hi - lo) / 1e6:.1f} ms, device busy (union of streams) {busy / 1e6:.1f} ms, idle {(hi - lo - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
for g, a, b in sorted(gaps, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"  {g / 1e6:8.2f} ms  after {short(a)}  before {short(b)}")
