"""End to end through the UNMODIFIED reference decoder (oracle/_ref/colord, built from /root/reference by
oracle/Makefile.ref; it travels with the repo): the reference compresses a synthetic FASTQ, the `dna` and `qual` streams
of its archive are replaced by the ones cl_compress_shard produces on the GPU, and the reference's own `decompress`
must give back exactly what it gives for its own archive.
  * with the reference's part cut the whole ARCHIVE FILE is byte-identical;
  * with much smaller parts (the knob of DESIGN.md section 4: more, shorter range-coder chains) the archive differs but
    decodes to the same FASTQ — any cut at read boundaries is a valid CoLoRd archive."""
import hashlib
import os
import subprocess
import numpy as np
import pytest
import torch
from colord_amd import archive as AR
from colord_amd.fastq import read_fastx, write_fastq
from colord_amd.synth import make_reads
from bench import reference_part_bounds

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "colord")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/colord not built (make -C oracle ref)")
def test_gpu_streams_decoded_by_the_reference(ctx, tmp_path):
    rs = make_reads(seed=11, genome_len=120_000, target_bases=9_000_000, mean_scale=7000.0)      # > 2 reader packs
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    ref_arc, ref_out = str(tmp_path / "ref.colord"), str(tmp_path / "ref.fastq")
    subprocess.check_call([REF, "compress-ont", "-t", "4", fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    arc = AR.read_archive(ref_arc)
    assert {"dna", "qual", "header", "meta"} <= set(arc)
    rs = read_fastx(fq)                                     # as the compressor reads it
    lens = np.diff(rs.offsets).astype(np.uint32)
    k, a = 20, 16                                           # compression.cpp:62-68 for < 1 Gbase
    prm = dict(k=k, f=12, ci=4, cs=80, c=5, anchor_len=a, min_part_alt=64, max_rec=3, min_anchors=1, level=1, source=0, sparse=1,
               sparse_g=1.0, sparse_exponent=1.0, cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)
    packs = reference_part_bounds(lens, 1 << 22)
    assert len(packs) - 1 == len(arc["dna"].parts) >= 2
    quals = torch.from_numpy(rs.quals).to(ctx.device)
    qoff = torch.from_numpy(rs.offsets).to(ctx.device)

    def gpu_archive(part_bounds, path):
        reads = ctx.pack_readset(rs)
        dc = ctx.dna_coder(5, 1, 0)
        qc = ctx.qual_coder(2, 0, 1, (7, 14, 26), ())
        dna, dsz, qual, qsz, info = ctx.compress_shard(reads, prm, part_bounds, packs, dc, qc, quals, qoff)
        draw, qraw = dna.cpu().numpy().tobytes(), qual.cpu().numpy().tobytes()
        d_st, q_st = AR.Stream("dna", arc["dna"].raw_size), AR.Stream("qual", arc["qual"].raw_size)
        o = p = 0
        for i in range(len(part_bounds) - 1):
            d_st.parts.append((int(part_bounds[i + 1] - part_bounds[i]), draw[o:o + int(dsz[i])])); o += int(dsz[i])
            q_st.parts.append((0, qraw[p:p + int(qsz[i])])); p += int(qsz[i])
        streams = [d_st if n == "dna" else q_st if n == "qual" else st for n, st in arc.items()]
        AR.write_archive(path, streams)
        qc.free(); dc.free(); reads.free()
        return d_st, q_st

    # (1) the reference's own part cut: same streams, byte for byte
    d_st, q_st = gpu_archive(packs, str(tmp_path / "gpu.colord"))
    assert [p for _, p in d_st.parts] == [p for _, p in arc["dna"].parts]
    assert [p for _, p in q_st.parts] == [p for _, p in arc["qual"].parts]
    subprocess.check_call([REF, "decompress", str(tmp_path / "gpu.colord"), str(tmp_path / "gpu.fastq")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert sha(str(tmp_path / "gpu.fastq")) == sha(ref_out)
    # (2) parts of 64 Ki symbols (bench.py's default): a different, equally valid archive
    small = reference_part_bounds(lens, 1 << 16)
    assert len(small) > 8 * len(packs)
    gpu_archive(small, str(tmp_path / "gpu_small.colord"))
    subprocess.check_call([REF, "decompress", str(tmp_path / "gpu_small.colord"), str(tmp_path / "gpu_small.fastq")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert sha(str(tmp_path / "gpu_small.fastq")) == sha(ref_out)
