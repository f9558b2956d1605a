#!/bin/bash
# A/B of the round-6 pipeline switches at 50 Gbases on ONE box (boxes differ by +-0.4 s): each variant = one short bench run.
# usage: tools/r06_ab.sh "NAME:ENV=1 ENV2=2" "NAME2:" ...   -> gpurun_out/ab_<NAME>.json / .log
mkdir -p gpurun_out
for v in "$@"; do
  name="${v%%:*}"; envs="${v#*:}"
  echo "=== $name [$envs]"
  env $envs COLORD_HIP_STREAM_DEBUG=1 python bench.py --steps ${AB_STEPS:-2} --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0 ${AB_ARGS} > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.log
  python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/ab_{n}.json").read().strip().splitlines()[-1])
    r=d.get("roofline",{})
    ks=r.get("kernel_ms_per_step",{})
    top=sorted(ks.items(), key=lambda kv:-kv[1])[:14]
    print(n, "ms_per_step", d.get("ms_per_step"), "value", d.get("value"), "digest_stable", d.get("parts_digest_stable"), "equal_ref", d.get("bench_parts_sha256_equal_ref"))
    print("   ", "; ".join(f"{k.split('(')[0][:38]} {v:.0f}" for k,v in top))
except Exception as e:
    print(n, "FAILED", e)
PY
  grep -h "^\[stream\]" gpurun_out/ab_$name.log | tail -2
done
