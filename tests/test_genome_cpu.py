"""CPU suite for the host pieces of reference-genome mode (config 4 of BASELINE.json, `compress-ont -G genome [-s]`):
the `ref-genome` stream (CReferenceGenome::Store / its archive constructor, reference_genome.cpp:235-279,325-370: every
sequence a plain read under the DNA coder's level-9 models) and the MD5 that replaces it without -s (reference_genome.cpp:29-67,
205-213).  Judges: archives written by the UNMODIFIED reference (tests/golden/archives/c4_ont_genome_*, make_archives.py)."""
import ctypes as C
import gzip
import os
import numpy as np
from colord_amd import _native as N, archive as AR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARC = os.path.join(ROOT, "tests", "golden", "archives")


def genome():
    """CReferenceGenome's reader (reference_genome.cpp:106-196, reference_genome.h:50-55): multi-FASTA, only ACGT (any case) kept."""
    lut = np.full(256, 255, np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    seqs, cur = [], []
    for line in gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis-reference.fna.gz"), "rb"):
        if line.startswith(b">"):
            if cur:
                seqs.append(np.concatenate(cur))
            cur = []
        else:
            v = lut[np.frombuffer(line.strip(), np.uint8)]
            cur.append(v[v < 4])
    if cur:
        seqs.append(np.concatenate(cur))
    codes = np.concatenate(seqs)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return codes, off


def test_ref_genome_stream_equals_the_reference_and_decodes():
    lib = N.load()
    codes, off = genome()
    n_seqs, payload = AR.read_archive(os.path.join(ARC, "c4_ont_genome_stored.colord"))["ref-genome"].parts[0]
    assert n_seqs == len(off) - 1
    out = np.zeros(len(codes) // 2 + 4096, np.uint8); n = C.c_uint64(0)
    assert lib.cl_genome_encode(codes.ctypes.data, off.ctypes.data, n_seqs, out.ctypes.data, len(out), C.byref(n)) == 0
    assert out[:n.value].tobytes() == payload, "ref-genome stream differs from the reference's"
    # capacity protocol, then the inverse
    small = np.zeros(16, np.uint8)
    assert lib.cl_genome_encode(codes.ctypes.data, off.ctypes.data, n_seqs, small.ctypes.data, 16, C.byref(n)) == N.CL_E_CAPACITY and n.value == len(payload)
    buf = np.frombuffer(payload, np.uint8)
    dec = np.zeros(len(codes), np.uint8); doff = np.zeros(n_seqs + 1, np.uint64); got = C.c_uint64(0)
    assert lib.cl_genome_decode(buf.ctypes.data, len(buf), n_seqs, dec.ctypes.data, len(dec), doff.ctypes.data, C.byref(got)) == 0
    assert got.value == len(codes) and np.array_equal(doff, off) and np.array_equal(dec, codes)


def test_genome_md5_equals_the_references_checksum():
    lib = N.load()
    codes, off = genome()
    meta = AR.read_archive(os.path.join(ARC, "c4_ont_genome_external.colord"))["meta"].parts[0][1]
    md = np.zeros(16, np.uint8)
    assert lib.cl_genome_md5(codes.ctypes.data, off.ctypes.data, len(off) - 1, md.ctypes.data) == 0
    assert md.tobytes() == meta[-16:]
    # RFC 1321 test vector through the same code path is not reachable (packed form); a second genome must differ
    codes2 = codes.copy(); codes2[12345] ^= 1
    md2 = np.zeros(16, np.uint8)
    lib.cl_genome_md5(codes2.ctypes.data, off.ctypes.data, len(off) - 1, md2.ctypes.data)
    assert md2.tobytes() != md.tobytes()


def test_ragged_sequences_round_trip():
    lib = N.load()
    rng = np.random.default_rng(3)
    lens = [0, 1, 2, 3, 4, 5, 255, 256, 511, 513, 70000]
    codes = rng.integers(0, 4, sum(lens), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    out = np.zeros(len(codes) + 4096, np.uint8); n = C.c_uint64(0)
    assert lib.cl_genome_encode(codes.ctypes.data, off.ctypes.data, len(lens), out.ctypes.data, len(out), C.byref(n)) == 0
    dec = np.zeros(len(codes) + 1, np.uint8); doff = np.zeros(len(lens) + 1, np.uint64); got = C.c_uint64(0)
    assert lib.cl_genome_decode(out.ctypes.data, n.value, len(lens), dec.ctypes.data, len(dec), doff.ctypes.data, C.byref(got)) == 0
    assert np.array_equal(doff, off) and np.array_equal(dec[:got.value], codes)


def test_multi_gpu_drivers_genome_reader_and_pseudo_reads(tmp_path):
    """colord_amd.fastq.read_genome / genome_pseudo_reads (what `python -m colord_amd.mgpu -G` feeds the library): the checksum of
    the reference's own archive through them, the state machine's corner cases, and the pseudo-read count the reference wrote."""
    from colord_amd.fastq import read_genome, genome_pseudo_reads
    import struct
    p = tmp_path / "g.fna"
    p.write_bytes(gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis-reference.fna.gz"), "rb").read())
    codes, off = read_genome(str(p))
    c2, o2 = genome()
    assert np.array_equal(codes, c2) and np.array_equal(off, o2)
    meta = AR.read_archive(os.path.join(ARC, "c4_ont_genome_external.colord"))["meta"].parts[0][1]
    read_len, overlap, n_pseudo = struct.unpack("<III", meta[-28:-16])
    pc, po = genome_pseudo_reads(codes, off, read_len, overlap)
    assert len(po) - 1 == n_pseudo and int(po[-1]) == len(pc)
    assert np.array_equal(pc[:read_len], codes[:read_len]) and np.array_equal(pc[read_len:read_len + 10], codes[read_len - overlap:read_len - overlap + 10])
    # CR LF, lower case, other letters dropped, a '>' line right after a header is sequence data, an empty last sequence is a sequence
    q = tmp_path / "x.fa"
    q.write_bytes(b">s1 x\nACGTN\nacgt\r\n>s2\n>ACG\nTT\n\n>s3\n")
    codes, off = read_genome(str(q))
    assert codes.tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 0, 1, 2, 3, 3] and off.tolist() == [0, 8, 13, 13]
    pc, po = genome_pseudo_reads(codes, off, 4, 1)
    assert po.tolist() == [0, 4, 8, 10, 14, 16]
    (tmp_path / "bad.fa").write_bytes(b"ACGT\n")
    import pytest
    with pytest.raises(ValueError):
        read_genome(str(tmp_path / "bad.fa"))
