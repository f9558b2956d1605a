set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r5d
export COLORD_HIP_ENCODE_LANES=1 COLORD_HIP_NO_DNA_PREP=1 COLORD_HIP_NO_QUAL_PREP=1 COLORD_HIP_EVOLVE_DEPTH=0
B="python bench.py --bases 3e9 --k 25 --a 22 --steps 1 --warmup 0 --no-cpu-baseline --no-ref-cut --e2e-bases 0"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d /tmp/sq1 -o run -- $B > gpurun_out/r5d/b1.json 2> gpurun_out/r5d/b1.err
python - <<'PY' > gpurun_out/r5d/sq.txt 2>&1
import sqlite3, glob, collections
db = sqlite3.connect(glob.glob('/tmp/sq1/**/*.db', recursive=True)[0])
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = collections.defaultdict(dict)
for k, c, n, s in rows:
    k = (k or '').replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40]
    d[k][c] = d[k].get(c, 0) + s; d[k]['n'] = n
names = ['SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_INSTS_VALU']
print(f"{'kernel':42s} {'n':>6s} " + ' '.join(f"{x[3:]:>16s}" for x in names))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:40]:
    print(f"{k:42s} {v['n']:6d} " + ' '.join(f"{v.get(x, 0):16.3e}" for x in names))
PY
head -45 gpurun_out/r5d/sq.txt
