"""Host form of the synthetic-input generator: pinned by a digest (the recipe is part of what the bench line claims), and its
FASTQ text parses back to the generated arrays."""
import hashlib
import numpy as np
from colord_amd import ontsim
from colord_amd.fastq import read_fastx


def test_host_generator_is_pinned_and_fastq_round_trips(tmp_path):
    t = ontsim.ReadTable(seed=3, genome_len=300_000, target_bases=2_000_000)
    codes, off, quals = ontsim.host_reads(t)
    assert off[-1] == len(codes) == len(quals) and codes.max() <= 3
    frac = np.bincount(quals, minlength=128)[[37, 43, 53, 67]] / len(quals)
    assert np.allclose(frac, [0.1, 0.2, 0.4, 0.3], atol=0.01)
    lens = np.diff(off)
    assert abs(lens.sum() / t.len_src.sum() - 0.9996) < 0.002          # -2 % deletions, +2 % insertions of the kept bases
    fq = str(tmp_path / "s.fastq")
    assert ontsim.write_fastq(t, fq) == len(codes)
    rs = read_fastx(fq)
    assert np.array_equal(rs.bases, codes) and np.array_equal(rs.offsets, off) and np.array_equal(rs.quals, quals)
    assert rs.headers[1] == b"read_1 ch=1 start_time=2020-01-01T00:00:01Z"
    # overlapping reads really share sequence: errors are per read, the genome is one function of the position
    d = hashlib.sha256(codes.tobytes() + quals.tobytes()).hexdigest()
    t2 = ontsim.ReadTable(seed=3, genome_len=300_000, target_bases=2_000_000)
    c2, o2, q2 = ontsim.host_reads(t2, 0, t2.n_reads)
    assert hashlib.sha256(c2.tobytes() + q2.tobytes()).hexdigest() == d
