#!/bin/bash
# A/B of CU partitioning at 50 Gbases: one warm pass + one timed pass of the default bench per setting of COLORD_HIP_CU_MASK;
# prints the pass time and k_range_code's / k_dna_walk's / k_lis_anchors' time per launch.  Run through gpurun from the repo root.
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/cu_mask
i=0
for M in "" "$@"; do
  i=$((i+1))
  COLORD_HIP_CU_MASK="$M" python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ref-cut --e2e-bases 0 > gpurun_out/cu_mask/run_$i.json 2> gpurun_out/cu_mask/run_$i.err
  python - "$M" gpurun_out/cu_mask/run_$i.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    k = d["roofline"]["kernel_ms_per_step"]
    print(f"mask [{sys.argv[1]}]: {d['ms_per_step'] / 1e3:.2f} s/pass; k_range_code {k.get('k_range_code', 0) / 100:.1f} ms/launch; k_lis_anchors {k.get('k_lis_anchors', 0) / 200:.1f}; k_dna_walk<false> {k.get('k_dna_walk<false>', 0) / 50:.1f}; "
          f"k_align_quad {k.get('k_align_quad', 0) / 200:.1f}; k_sort_scatter<u64,v,8> {k.get('k_sort_scatter<unsigned long, true, 8u>', 0) / 150:.1f}; digest stable {d.get('parts_digest_stable')}")
except Exception as e:
    print(f"mask [{sys.argv[1]}]: failed: {e!r}", open(sys.argv[2].replace('.json', '.err')).read()[-400:])
PY
done
