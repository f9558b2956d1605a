#!/bin/bash
# Round-6 profile on the GPU box (run from the repo root through gpurun): rocprofv3 kernel trace + stats of one timed pass of the
# default bench (50 Gbases), the per-queue busy times and a one-second window of the timeline.  $1: tag for the output names.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r06}; mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o st -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0 > gpurun_out/$T/bench_under_rocprof.json 2> gpurun_out/$T/bench_under_rocprof.err
KT=$(find /tmp/st -name "*kernel_trace.csv" | head -1); KS=$(find /tmp/st -name "*kernel_stats.csv" | head -1)
cp $KS gpurun_out/$T/kernel_stats.csv
python tools/queue_busy.py $KT > gpurun_out/$T/queue_busy.txt 2>&1
python tools/window_dump.py $KT 8.0 1.2 0.5 > gpurun_out/$T/window.txt 2>&1
python tools/window_dump.py $KT 0.0 1.8 0.3 > gpurun_out/$T/window_start.txt 2>&1   # passes 1 and 2a: what runs before the first chunk's stage A
python tools/lane_gaps.py $KT > gpurun_out/$T/lane_gaps.txt 2>&1
python tools/copy_top.py $KT > gpurun_out/$T/copy_top.txt 2>&1
python - $KT > gpurun_out/$T/kernel_minmax.txt <<'PY'
import csv, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(f"{'kernel':60s} {'calls':>8s} {'total_ms':>10s} {'avg_ms':>9s} {'max_ms':>9s}")
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:70]:
    print(f"{n[-60:]:60s} {len(v):8d} {sum(v):10.1f} {sum(v) / len(v):9.3f} {max(v):9.2f}")
print("dispatches in the run:", sum(len(v) for v in d.values()))
PY
head -30 gpurun_out/$T/kernel_minmax.txt; head -8 gpurun_out/$T/queue_busy.txt
