# UTCL1 (L1 TLB) requests / hits / misses per kernel, serial configuration (one lane, no preparation threads), 3 Gbases.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5n
export COLORD_HIP_ENCODE_LANES=1 COLORD_HIP_NO_DNA_PREP=1 COLORD_HIP_NO_QUAL_PREP=1 COLORD_HIP_EVOLVE_DEPTH=0
B="python bench.py --bases 3e9 --k 25 --a 22 --steps 1 --warmup 0 --no-cpu-baseline --no-ref-cut --e2e-bases 0"
timeout 400 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum -d /tmp/tl1 -o run -- $B > gpurun_out/r5n/b1.json 2> gpurun_out/r5n/b1.err
python - <<'PY' > gpurun_out/r5n/tlb.txt 2>&1
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('/tmp/tl1/**/*.db', recursive=True)[0])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = collections.defaultdict(dict)
for k, c, n, s in rows:
    k = (k or '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    d[k][c] = d[k].get(c, 0) + s; d[k]['n'] = n
names = ['TCP_UTCL1_REQUEST_sum','TCP_UTCL1_TRANSLATION_HIT_sum','TCP_UTCL1_TRANSLATION_MISS_sum','TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum']
print(f"{'kernel':42s} {'n':>6s} " + ' '.join(f"{x[10:-4][:18]:>18s}" for x in names) + "  miss/req")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get(names[2], 0))[:30]:
    r = v.get(names[0], 0) or 1
    print(f"{k:42s} {v['n']:6d} " + ' '.join(f"{v.get(x, 0):18.3e}" for x in names) + f"  {v.get(names[2], 0) / r:8.3f}")
PY
head -34 gpurun_out/r5n/tlb.txt; tail -3 gpurun_out/r5n/b1.err
