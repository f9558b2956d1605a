#!/bin/bash
# Clocks and power while the default bench runs: samples rocm-smi twice a second into gpurun_out/clock/samples.txt and prints a summary.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/clock
( while true; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/clock/samples.txt &
W=$!
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0 > gpurun_out/clock/bench.json 2> gpurun_out/clock/bench.err
kill $W
python - <<'PY'
import re
s=[l for l in open("gpurun_out/clock/samples.txt") if "sclk" in l]
print(len(s), "samples; first:", s[0].strip()[:300])
def col(pat):
    v=[]
    for l in s:
        m=re.search(pat,l)
        if m: v.append(float(m.group(1)))
    return v
for name,pat in [("sclk MHz", r"sclk clock level: \d+: \((\d+)Mhz\)"), ("mclk MHz", r"mclk clock level: \d+: \((\d+)Mhz\)"), ("power W", r"Power \(W\): ([\d.]+)"), ("use %", r"GPU use \(%\): (\d+)")]:
    v=col(pat)
    if v:
        v2=sorted(v); print(f"{name}: n {len(v)} min {v2[0]:.0f} p10 {v2[len(v)//10]:.0f} median {v2[len(v)//2]:.0f} p90 {v2[len(v)*9//10]:.0f} max {v2[-1]:.0f}")
PY
