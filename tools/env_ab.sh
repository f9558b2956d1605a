#!/bin/bash
# A/B of environment settings at 50 Gbases: one warm + one timed pass of the default bench per argument ("VAR=value VAR2=value ..."; "" = defaults).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/env_ab
i=0
for M in "$@"; do
  i=$((i+1))
  env $M python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0 $BENCH_EXTRA > gpurun_out/env_ab/run_$i.json 2> gpurun_out/env_ab/run_$i.err
  python - "$M" gpurun_out/env_ab/run_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    top = ", ".join(f"{n.replace('unsigned ', 'u')} {v / 1e3:.2f}" for n, v in list(k.items())[:12])
    print(f"[{sys.argv[1]}]: {d['ms_per_step'] / 1e3:.2f} s/pass; digest stable {d.get('parts_digest_stable')}; kernel s/pass: {top}")
except Exception as e:
    print(f"[{sys.argv[1]}]: failed: {e!r}", open(sys.argv[2].replace('.json', '.err')).read()[-400:])
PY
done
