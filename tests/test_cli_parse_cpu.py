"""The command-line compressor's input reader without a GPU (`colord_hip parse-check`, csrc/cli/compress.cpp): a plain FASTQ is indexed
by several threads over byte ranges of the mapped file and the chunks are filled by parallel copies; what the compressor is handed —
bases, qualities, offsets, reader packs (in_reads.cpp:62-77), coder parts (--part-symbols), ids — must be exactly what the
sequential reader returns, on the inputs the reference's reader accepts (in_reads.cpp:79-92,188-226: CR LF, blank lines, a last line
without end of line, '+' lines that repeat the id) and with quality lines that begin with '@' or '+' (what makes a record start
ambiguous from the middle of a file).  Also here: bench.py's numpy statement of the 4-avg quantisation against the decoder."""
import os
import random
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")


def make_fastq(path, n, seed, crlf=False, blank=False, plus_id=False, trailing=True, max_len=3000):
    rnd = random.Random(seed)
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        for i in range(n):
            L = rnd.randint(1, max_len)
            seq = "".join(rnd.choice("ACGTN" if i % 50 == 0 else "ACGT") for _ in range(L))
            q = "".join(chr(33 + rnd.randint(0, 60)) for _ in range(L))
            if i % 3 == 0:
                q = "@" + q[1:]
            if i % 7 == 0:
                q = "+" + q[1:]
            hid = f"read{i} some/comment"
            f.write("@" + hid + nl + seq + nl + ("+" + hid if plus_id and i % 2 else "+") + nl + q + (nl if (trailing or i < n - 1) else ""))
            if blank and i % 11 == 0:
                f.write(nl)


def parse_check(path, threads, extra=()):
    env = dict(os.environ, COLORD_HIP_INDEX_MIN_BYTES="1000")
    return subprocess.run([CLI, "parse-check", "--parse-threads", str(threads), "--chunk-bases", "2e6"] + list(extra) + [path], capture_output=True, text=True, env=env)


@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(blank=True, plus_id=True), dict(trailing=False), dict(crlf=True, blank=True, trailing=False)],
                         ids=["plain", "crlf", "blank+plusid", "no-final-eol", "crlf+blank+no-final-eol"])
def test_indexed_reader_equals_the_sequential_one(tmp_path, kw):
    fq = str(tmp_path / "in.fastq")
    make_fastq(fq, 3000, seed=5, **kw)
    for part in ("4194304", "65536"):
        ref = parse_check(fq, 1, ["--part-symbols", part])
        assert ref.returncode == 0 and "sequential reader" in ref.stderr, ref.stderr
        for t in (2, 3, 7, 32):
            r = parse_check(fq, t, ["--part-symbols", part])
            assert r.returncode == 0 and "indexed reader" in r.stderr, r.stderr
            assert r.stdout == ref.stdout
    # default parts are the reader packs; 64-Ki parts are many more and end where the chunk ends
    a = parse_check(fq, 1, ["--part-symbols", "4194304"]).stdout.splitlines()[0].split()
    b = parse_check(fq, 1, ["--part-symbols", "65536"]).stdout.splitlines()[0].split()
    assert a[a.index("packs") - 1] == a[a.index("parts") - 1] and int(b[b.index("parts") - 1]) > 8 * int(b[b.index("packs") - 1])


def test_malformed_input_is_reported_as_by_the_sequential_reader(tmp_path):
    fq = str(tmp_path / "bad.fastq")
    make_fastq(fq, 2000, seed=6)
    data = open(fq).read().split("\n")
    data[4 * 1500 + 3] = data[4 * 1500 + 3][:-1]             # one quality line a symbol short, three quarters into the file
    open(fq, "w").write("\n".join(data))
    for t in (1, 8):
        r = parse_check(fq, t)
        assert r.returncode != 0 and "sequence and quality lengths differ" in r.stderr, r.stderr


def test_4avg_quantisation_of_bench_equals_the_decoder(tmp_path):
    """bench.py's `qual_round_trip_checked` compares decoded qualities with quantised_quals_4avg(generator's qualities): pinned here on an
    archive written by the unmodified reference (tests/golden/archives) against the host decoder's output."""
    from bench import quantised_quals_4avg
    from colord_amd.fastq import read_fastx
    out = str(tmp_path / "o.fastq")
    subprocess.check_call([CLI, "decompress", os.path.join(ROOT, "tests", "golden", "archives", "c1_ont_default.colord"), out], stderr=subprocess.DEVNULL)
    dec = read_fastx(out)
    src = read_fastx(os.path.join(ROOT, "tests", "data", "M.bovis.fastq.gz"))
    n = len(dec.offsets) - 1
    want = quantised_quals_4avg(src.quals[:src.offsets[n]], src.offsets[:n + 1].astype(np.int64))
    assert np.array_equal(want, dec.quals)


@pytest.mark.parametrize("form", ["indexed", "sequential", "gz", "fasta"])
def test_a_second_and_third_pass_over_the_input_cut_the_same_chunks(tmp_path, form):
    """`--stream-input` reads its input three times (Reader::rewind: the mapping's cursor, the record index, or gzrewind): every pass must
    return the chunks of the first — bytes, offsets, packs, parts — and the ids and totals must be those of ONE pass."""
    import gzip
    fq = str(tmp_path / "in.fastq")
    make_fastq(fq, 4000, seed=21, crlf=form == "sequential", blank=True)
    path, threads = fq, 4 if form == "indexed" else 1
    if form == "gz":
        path = fq + ".gz"
        with open(fq, "rb") as f, gzip.open(path, "wb") as g:
            g.write(f.read())
    if form == "fasta":
        path = str(tmp_path / "in.fasta")
        lines = open(fq).read().split("\n")
        recs = [l for l in lines if l]                                         # (blank lines dropped: four lines per record)
        with open(path, "w") as f:
            for i in range(0, len(recs) - 3, 4):
                f.write(">" + recs[i][1:] + "\n" + recs[i + 1].replace("N", "A") + "\n")
    one = parse_check(path, threads)
    three = parse_check(path, threads, ["--passes", "3"])
    assert one.returncode == 0 and three.returncode == 0, one.stderr + three.stderr
    a, b = one.stdout.splitlines(), three.stdout.splitlines()
    chunks = [l for l in a if l.startswith("chunk")]
    assert len(chunks) >= 2 and a[-1].startswith("ids ")
    assert b == chunks + ["pass 2"] + chunks + ["pass 3"] + chunks + [a[-1]]
    assert ("indexed" in one.stderr) == (form == "indexed")
