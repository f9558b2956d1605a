#!/usr/bin/env python3
"""One-off check at a size the test suite does not reach: the command-line compressor on a synthetic ONT FASTQ with the
bench's read-length distribution (reads up to 200 kb: hundreds of walk / emission chunks per read), decoded by the
unmodified reference and compared with the reference's own archive stream by stream."""
import hashlib, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd import archive as AR
from colord_amd.fastq import write_fastq
from colord_amd.synth import make_reads
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "colord"); CLI = os.path.join(ROOT, "colord_amd", "colord_hip")
bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "compress-ont"
extra = sys.argv[3:]                                   # e.g. -p ratio
sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
rs = make_reads(seed=77, genome_len=int(bases / 16.7), target_bases=bases)
print("reads", rs.n_reads, "bases", len(rs.bases), "longest", int(max(rs.lengths())) if hasattr(rs, "lengths") else "?", flush=True)
with tempfile.TemporaryDirectory() as tmp:
    fq = os.path.join(tmp, "in.fastq"); write_fastq(fq, rs)
    ref_arc, ref_out, my_arc, my_out = (os.path.join(tmp, x) for x in ("ref.colord", "ref.fastq", "gpu.colord", "gpu.fastq"))
    t = time.time(); subprocess.check_call([REF, mode, "-t", str(os.cpu_count())] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print("reference compress %.1f s" % (time.time() - t), flush=True)
    t = time.time(); subprocess.check_call([CLI, mode] + extra + [fq, my_arc]); print("colord_hip compress %.1f s (whole process: parsing, context set-up, first-call allocations)" % (time.time() - t), flush=True)
    a, b = AR.read_archive(ref_arc), AR.read_archive(my_arc)
    for name in a:
        same = [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts]
        print("stream %-8s parts %5d  %s" % (name, len(a[name].parts), "identical" if same else ("DIFFERENT" if name != "info" else "differs (time stamp)")), flush=True)
    t = time.time(); subprocess.check_call([REF, "decompress", my_arc, my_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL); print("reference decompress of the GPU archive %.1f s" % (time.time() - t), flush=True)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    own_out = os.path.join(tmp, "own.fastq")
    t = time.time(); subprocess.check_call([CLI, "decompress", my_arc, own_out]); print("colord_hip decompress of the GPU archive %.1f s (host decoders)" % (time.time() - t), "; == the reference's output:", sha(own_out) == sha(ref_out), flush=True)
    print("decoded FASTQ == what the reference decodes from its own archive:", sha(my_out) == sha(ref_out), "; == input FASTQ (lossless modes):", sha(my_out) == sha(fq))
