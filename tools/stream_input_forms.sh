# --stream-input against the default on the M. bovis fixture: -G -s, FASTA input, HiFi -p ratio — every stream but `info` must be equal.
set -e
cd $GRAFT_REPO_ROOT; T=$(mktemp -d)
python - $T <<'PY'
import gzip, os, sys
T = sys.argv[1]; R = os.environ["GRAFT_REPO_ROOT"]
open(T + "/m.fastq", "wb").write(gzip.open(R + "/tests/data/M.bovis.fastq.gz", "rb").read())
open(T + "/g.fna", "wb").write(gzip.open(R + "/tests/data/M.bovis-reference.fna.gz", "rb").read())
l = open(T + "/m.fastq").read().split("\n")
with open(T + "/m.fasta", "w") as f:
    for i in range(0, len(l) - 3, 4): f.write(">" + l[i][1:] + "\n" + l[i + 1] + "\n")
PY
C=colord_amd/colord_hip
$C compress-ont --chunk-bases 2e6 -G $T/g.fna -s $T/m.fastq $T/a1.colord 2>/dev/null; $C compress-ont --chunk-bases 2e6 -G $T/g.fna -s --stream-input $T/m.fastq $T/a2.colord 2>/dev/null
$C compress-ont --chunk-bases 2e6 $T/m.fasta $T/b1.colord 2>/dev/null; $C compress-ont --chunk-bases 2e6 --stream-input $T/m.fasta $T/b2.colord 2>/dev/null
$C compress-pbhifi --chunk-bases 2e6 -p ratio $T/m.fastq $T/c1.colord 2>/dev/null; $C compress-pbhifi --chunk-bases 2e6 -p ratio --stream-input $T/m.fastq $T/c2.colord 2>/dev/null
python - $T <<'PY'
import sys, hashlib, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from colord_amd import archive as AR
T = sys.argv[1]
for x in "abc":
    a, b = AR.read_archive(f"{T}/{x}1.colord"), AR.read_archive(f"{T}/{x}2.colord")
    same = set(a) == set(b) and all([(m, hashlib.sha256(p).hexdigest()) for m, p in a[s].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[s].parts] for s in a if s != "info")
    print(x, "streams:", sorted(a), "chunks of dna parts:", len(a["dna"].parts), "same:", same)
PY
colord_amd/colord_hip decompress $T/a2.colord $T/o.fastq && echo "decompressed (qualities are 4-avg: not compared)"
rm -rf $T
