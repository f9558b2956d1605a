"""The command-line compressor (colord_amd/colord_hip: host C++ over the C ABI) against the unmodified reference:
its archive must be decoded by the reference's `colord decompress` to exactly what the reference decodes from its own
archive of the same FASTQ, and every stream except `info` (time stamp, command line) must be byte-identical."""
import hashlib
import os
import subprocess
import pytest
from colord_amd import archive as AR
from colord_amd.fastq import write_fastq
from colord_amd.synth import make_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "colord")
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("mode,prio,seed", [("compress-ont", None, 5), ("compress-ont", "balanced", 6), ("compress-pbhifi", None, 7), ("compress-pbraw", "ratio", 8)])
def test_cli_archive_decoded_by_reference(tmp_path, mode, prio, seed):
    rs = make_reads(seed=seed, genome_len=100_000, target_bases=5_000_000, mean_scale=6000.0)
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    extra = ["-p", prio] if prio else []
    ref_arc, ref_out, my_arc, my_out = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "gpu.colord", "gpu.fastq"))
    subprocess.check_call([REF, mode, "-t", "4"] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, mode] + extra + [fq, my_arc])
    subprocess.check_call([REF, "decompress", my_arc, my_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert sha(my_out) == sha(ref_out)
    # ... and by this build's own decompressor (the inverse path, csrc/decode.hip)
    own_out = str(tmp_path / "own.fastq")
    subprocess.check_call([CLI, "decompress", my_arc, own_out])
    assert sha(own_out) == sha(ref_out)
    a, b = AR.read_archive(ref_arc), AR.read_archive(my_arc)
    assert set(a) == set(b)
    for name in a:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts], name


@pytest.mark.skipif(not os.path.exists(CLI), reason="needs colord_amd/colord_hip")
@pytest.mark.parametrize("cfg", ["c1_ont_default", "c2_hifi_org", "c3_clr_ratio", "c6_ont_org", "c7_hifi_balanced", "s6m_ont", "s3m_ont_n_ratio", "s5m_hifi", "s6m_ont_k25", "s4m_ont_k23_balanced"])
def test_cli_round_trip_on_goldens(tmp_path, cfg):
    """compress on the GPU, decompress with this build's decoders: the output is what the reference returns for the same input
    (`decompressed_sha256` of the golden vectors: the input itself for -q org, the quantised qualities otherwise)."""
    import gzip, json
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", cfg, "streams.json")))
    fq = str(tmp_path / "in.fastq")
    if spec.get("synth"):
        write_fastq(fq, make_reads(**spec["synth"]))
    else:
        open(fq, "wb").write(gzip.open(os.path.join(ROOT, "tests", "data", spec["input"] + ".gz"), "rb").read())
    args = [a if a != "--priority" else "-p" for a in spec["args"]]
    if "-q" in args:
        pytest.skip("the command-line compressor has no -q yet") if not _cli_has("-q") else None
    arc, out = str(tmp_path / "a.colord"), str(tmp_path / "o.fastq")
    subprocess.check_call([CLI] + args + [fq, arc])
    subprocess.check_call([CLI, "decompress", arc, out])
    assert sha(out) == spec["decompressed_sha256"]


def _cli_has(flag):
    r = subprocess.run([CLI], capture_output=True, text=True)
    return flag in (r.stderr + r.stdout)


def _fixture(tmp_path, name="M.bovis.fastq", n_reads=0):
    import gzip
    lines = gzip.open(os.path.join(ROOT, "tests", "data", name + ".gz"), "rb").read().split(b"\n")
    lines = lines[:4 * n_reads] if n_reads else lines[:-1]
    fq = str(tmp_path / name)
    open(fq, "wb").write(b"\n".join(lines) + b"\n")
    return fq, lines


def _same_streams(ref_arc, my_arc):
    a, b = AR.read_archive(ref_arc), AR.read_archive(my_arc)
    assert set(a) == set(b)
    for name in a:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts], name


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("extra", [["-q", "org"], ["-q", "5-avg", "-p", "balanced"], ["-q", "2-avg", "-p", "ratio"], ["-q", "5-fix"], ["-q", "4-fix", "-p", "balanced"], ["-q", "2-fix", "-T", "11", "-D", "2", "20"],
                                   ["-q", "avg"], ["-q", "none", "-D", "5"], ["-q", "4-avg", "-T", "5", "12", "30"], ["-i", "none"], ["-i", "main", "-c", "3"], ["-k", "18", "-a", "15", "-f", "7", "-L", "3", "-H", "60"],
                                   ["-R", "all", "-r", "2", "--min-to-alt", "40"], ["-g", "2.5", "-x", "1.5", "-e", "1.2"]])
def test_cli_options_reproduce_the_reference(tmp_path, extra):
    """Every option of the compress sub-commands that changes the archive: same streams as the reference, and this build's
    decompressor returns what the reference's returns."""
    fq, _ = _fixture(tmp_path)
    ref_arc, ref_out, my_arc, my_out = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "gpu.colord", "gpu.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", "4"] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont"] + extra + [fq, my_arc])
    _same_streams(ref_arc, my_arc)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(ref_out)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("variant", ["fasta", "fasta_multiline_crlf", "fastq_crlf_blank_lines", "fastq_plus_repeats_id", "fastq_gz"])
def test_cli_input_formats(tmp_path, variant):
    """The reader's semantics (in_reads.cpp:62-226): FASTA and multi-line FASTA, CR LF, blank lines, '+' line equal to the id, gzip."""
    import gzip
    fq, lines = _fixture(tmp_path, "A.thaliana.fastq")
    recs = [lines[i:i + 4] for i in range(0, len(lines), 4)]
    if variant == "fasta":
        data = b"".join(b">" + r[0][1:] + b"\n" + r[1] + b"\n" for r in recs)
        src = str(tmp_path / "in.fasta")
    elif variant == "fasta_multiline_crlf":
        data = b"".join(b">" + r[0][1:] + b"\r\n" + b"\r\n".join(r[1][i:i + 70] for i in range(0, len(r[1]), 70)) + b"\r\n\r\n" for r in recs)
        src = str(tmp_path / "in.fasta")
    elif variant == "fastq_crlf_blank_lines":
        data = b"".join(b"\r\n".join(r) + b"\r\n" for r in recs) + b"\r\n\r\n"      # (CR LF + a blank line after EVERY record crashes the reference's k-mer counter)
        src = str(tmp_path / "in.fastq")
    elif variant == "fastq_plus_repeats_id":
        data = b"".join(r[0] + b"\n" + r[1] + b"\n+" + r[0][1:] + b"\n" + r[3] + b"\n" for r in recs)
        src = str(tmp_path / "in.fastq")
    else:
        data = b"\n".join(lines) + b"\n"
        src = str(tmp_path / "in.fastq.gz")
    if src.endswith(".gz"):
        gzip.open(src, "wb").write(data)
    else:
        open(src, "wb").write(data)
    ref_arc, ref_out, my_arc, my_out = (str(tmp_path / x) for x in ("ref.colord", "ref.out", "gpu.colord", "gpu.out"))
    subprocess.check_call([REF, "compress-pbraw", "-t", "4", src, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-pbraw", src, my_arc])
    _same_streams(ref_arc, my_arc)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(ref_out)


@pytest.mark.skipif(not os.path.exists(CLI), reason="needs colord_amd/colord_hip")
def test_cli_rejects_what_the_reference_rejects(tmp_path):
    fq, lines = _fixture(tmp_path, n_reads=8)
    low = lines[:]
    low[1] = low[1][:50] + low[1][50:60].lower() + low[1][60:]                  # lower-case bases: SymbToBinMap has no entry (utils.h:472-475)
    open(str(tmp_path / "low.fastq"), "wb").write(b"\n".join(low) + b"\n")
    r = subprocess.run([CLI, "compress-ont", str(tmp_path / "low.fastq"), str(tmp_path / "o.colord")], capture_output=True, text=True)
    assert r.returncode == 1 and "Only ACGTN symbols supported inside a read" in r.stderr
    badq = lines[:]
    badq[3] = badq[3][:10] + b"\x1f" + badq[3][11:]                             # a quality byte below '!'
    open(str(tmp_path / "badq.fastq"), "wb").write(b"\n".join(badq) + b"\n")
    r = subprocess.run([CLI, "compress-ont", str(tmp_path / "badq.fastq"), str(tmp_path / "o.colord")], capture_output=True, text=True)
    assert r.returncode == 1 and "quality" in r.stderr
    plus = lines[:]
    plus[2] = b"+something_else"
    open(str(tmp_path / "plus.fastq"), "wb").write(b"\n".join(plus) + b"\n")
    r = subprocess.run([CLI, "compress-ont", str(tmp_path / "plus.fastq"), str(tmp_path / "o.colord")], capture_output=True, text=True)
    assert r.returncode == 1 and "quality header not empty but different than read header" in r.stderr
    open(str(tmp_path / "trunc.fastq"), "wb").write(b"\n".join(lines[:-2]) + b"\n")
    r = subprocess.run([CLI, "compress-ont", str(tmp_path / "trunc.fastq"), str(tmp_path / "o.colord")], capture_output=True, text=True)
    assert r.returncode == 1 and "truncated" in r.stderr


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cli_several_chunks_equal_one(tmp_path):
    """--chunk-bases cuts the input into several chunks of the streaming compressor: same archive (and the reference's)."""
    rs = make_reads(seed=21, genome_len=400_000, target_bases=30_000_000, mean_scale=9000.0)
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    ref_arc, one, many = (str(tmp_path / x) for x in ("ref.colord", "one.colord", "many.colord"))
    subprocess.check_call([REF, "compress-ont", "-t", "8", fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", fq, one])
    r = subprocess.run([CLI, "compress-ont", "--chunk-bases", "9e6", fq, many], capture_output=True, text=True)
    assert r.returncode == 0 and "3 chunk(s)" in r.stderr, r.stderr
    _same_streams(ref_arc, one)
    _same_streams(ref_arc, many)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cli_ultra_long_reads(tmp_path):
    """Reads of more than 2^20 bases (ONT ultra-long), as reference read and as encoded read: the match keys take their position
    bits from the longest read / candidate at hand; same archive as the reference."""
    import numpy as np
    from colord_amd.fastq import ReadSet
    rng = np.random.default_rng(5)
    genome = rng.integers(0, 4, 1_600_000, dtype=np.uint8)

    def noisy(a, b, rev=False):
        s = genome[a:b].copy()
        if rev:
            s = (3 - s)[::-1].copy()
        r = rng.random(len(s))
        sub = (r >= 0.02) & (r < 0.05)
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()))) % 4
        return s[r >= 0.02]
    seqs = [noisy(50_000, 1_250_000), noisy(200_000, 1_350_000, rev=True)]          # 1.18 and 1.13 Mbases: reference read, then encoded against it
    for _ in range(40):
        a = int(rng.integers(0, 1_500_000)); ln = int(rng.integers(5_000, 40_000))
        seqs.append(noisy(a, min(a + ln, len(genome)), rev=bool(rng.integers(0, 2))))
    lens = np.array([len(s) for s in seqs]); assert lens[0] > (1 << 20) and lens[1] > (1 << 20)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    quals = np.frombuffer(b"%+5C", np.uint8)[rng.choice(4, int(off[-1]))]
    rs = ReadSet(np.concatenate(seqs), off, quals, [b"r%d" % i for i in range(len(seqs))], [False] * len(seqs), True)
    fq = str(tmp_path / "ul.fastq")
    write_fastq(fq, rs)
    ref_arc, my_arc, my_out = (str(tmp_path / x) for x in ("ref.colord", "gpu.colord", "gpu.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", "8", "-q", "org", fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", "-q", "org", fq, my_arc])
    _same_streams(ref_arc, my_arc)
    dna = sum(len(p) for _, p in AR.read_archive(my_arc)["dna"].parts)
    assert dna < 0.22 * off[-1]                               # the long reads really are coded against each other (2 bits/base plain = 0.25)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(fq)


@pytest.mark.skipif(not os.path.exists(CLI), reason="needs colord_amd/colord_hip")
@pytest.mark.parametrize("stored", [True, False])
def test_cli_reference_genome_mode_equals_the_reference(tmp_path, stored):
    """Config 4 of BASELINE.json through the command line: `compress-ont -G M.bovis-reference.fna [-s] M.bovis.fastq`.  Against the
    archive the UNMODIFIED reference wrote for the same command (tests/golden/archives/c4_ont_genome_*, make_archives.py): every
    stream but `info` byte-identical — `meta` (pseudo-read geometry, checksum), `ref-genome` (-s), `dna`, `qual`, `header` — and this
    build's decompressor returns the reference's output, from the archive alone (-s) or with the genome file (no -s)."""
    import gzip, json
    name = "c4_ont_genome_stored" if stored else "c4_ont_genome_external"
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "archives", "expected.json")))[name]
    fq, gen = str(tmp_path / "M.bovis.fastq"), str(tmp_path / "M.bovis-reference.fna")
    open(fq, "wb").write(gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis.fastq.gz"), "rb").read())
    open(gen, "wb").write(gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis-reference.fna.gz"), "rb").read())
    arc, out = str(tmp_path / "a.colord"), str(tmp_path / "o.fastq")
    subprocess.check_call([CLI, "compress-ont", "-G", gen] + (["-s"] if stored else []) + [fq, arc])
    a, b = AR.read_archive(os.path.join(ROOT, "tests", "golden", "archives", name + ".colord")), AR.read_archive(arc)
    assert set(a) == set(b)
    for s in a:
        if s != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[s].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[s].parts], s
    subprocess.check_call([CLI, "decompress"] + ([] if stored else ["-G", gen]) + [arc, out])
    assert sha(out) == exp["decompressed_sha256"]
    if os.path.exists(REF):                                  # and the reference reads it (in the build container)
        ref_out = str(tmp_path / "r.fastq")
        subprocess.check_call([REF, "decompress"] + ([] if stored else ["-G", gen]) + [arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert sha(ref_out) == exp["decompressed_sha256"]


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("extra", [[], ["-k", "25", "-a", "22"], ["-k", "23", "-a", "21"]], ids=["auto_k20_a16", "k25_a22", "k23_a21"])
def test_cli_250_mbases_of_the_bench_recipe_equal_the_reference(tmp_path, extra):
    """The recipe of config 5 (BASELINE.json: synthetic ONT, N50 ~ 20 kb, reads up to 200 kb, 4-avg qualities) at a size where
    every mechanism of the 50-Gbase run is in play — several chunks, both encode lanes with look-ahead, the walk of the next chunk
    in the coding tail, anchor batches, all four recursion levels, the four-per-wave and the wave-per-gap aligners, long context
    runs — against the unmodified reference on the same FASTQ: `meta`, `dna`, `qual`, `header` byte-identical, and this build's
    decompressor returns what the reference returns.  `extra`: the k-mer / anchor lengths the reference picks by input size
    (compression.cpp:62-93) forced to config 5's own (k = 25, a = 22 at 50 Gbases: 50-bit k-mers, 44-bit m-mers) and to the
    16-Gbase tier's (k = 23, a = 21)."""
    from colord_amd import ontsim
    table = ontsim.ReadTable(seed=29, genome_len=15_000_000, target_bases=250_000_000)
    fq = str(tmp_path / "in.fastq")
    n_bases = ontsim.write_fastq(table, fq)
    assert n_bases >= 240_000_000 and int(table.len_src.max()) >= 150_000          # (errors shorten the source lengths a little)
    ref_arc, my_arc, ref_out, my_out = (str(tmp_path / x) for x in ("ref.colord", "gpu.colord", "ref.fastq", "gpu.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", str(os.cpu_count() or 8)] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "60000000"] + extra + [fq, my_arc])   # five chunks
    a, b = AR.read_archive(ref_arc), AR.read_archive(my_arc)
    assert set(a) == set(b)
    for name in a:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts], name
    assert len(a["dna"].parts) > 50
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(ref_out)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cli_more_candidates_and_levels_than_this_build_holds_degrade(tmp_path):
    """`-c 20 -r 9`: the reference takes any value (arg_parse.cpp:455-640, encoder.cpp:1513-1575); this build's frames hold 16
    candidate views and recursion to depth 8.  The call must not fail: the reads are coded against their 16 best candidates, the
    archive announces c = 20 (the alternative-id model's alphabet) and BOTH decompressors return the input's reads."""
    rs = make_reads(seed=41, genome_len=60_000, target_bases=3_000_000, mean_scale=5000.0)
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    extra = ["-p", "ratio", "-c", "20", "-r", "9", "-q", "org"]
    ref_arc, ref_out, my_arc, my_out, my_out2 = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "gpu.colord", "gpu.fastq", "gpu2.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", "4"] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont"] + extra + [fq, my_arc])
    subprocess.check_call([REF, "decompress", my_arc, my_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "decompress", my_arc, my_out2])
    assert sha(my_out) == sha(fq) and sha(my_out2) == sha(fq)            # -q org: lossless
    assert os.path.getsize(my_arc) <= os.path.getsize(ref_arc) * 1.02   # at most a little larger than with all 20 candidates


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("mode,extra,hifi", [("compress-pbhifi", [], True), ("compress-pbhifi", ["-p", "ratio"], True), ("compress-ont", ["-p", "ratio"], False),
                                             ("compress-ont", ["-p", "balanced"], False), ("compress-pbraw", ["-p", "ratio"], False)],
                         ids=["pbhifi_default", "pbhifi_ratio", "ont_ratio", "ont_balanced", "pbraw_ratio"])
def test_cli_presets_at_100_mbases_equal_the_reference(tmp_path, mode, extra, hifi):
    """The presets other than `compress-ont` default (arg_parse.cpp:89-408) at a size where the machinery of the 50-Gbase run is in play —
    several chunks, both encode lanes, the preparation threads, the work-group / four-per-wave / wave-per-gap aligners, the wave
    emission, 8-byte anchor slots — against the unmodified reference on the same FASTQ: HiFi (level 2, k-mer anchors of shared
    k-mers, 5-avg qualities; reads with 0.3 % errors), `-p ratio` (level 3, c = 10, recursion to depth 6, every read a reference read),
    `-p balanced` (level 2, sparse references), PBRaw (qualities dropped).  Every stream but `info` byte-identical; this build's
    decompressor returns what the reference's returns."""
    from colord_amd import ontsim
    fq = str(tmp_path / "in.fastq")
    if hifi:
        rs = make_reads(seed=31, genome_len=8_000_000, target_bases=110_000_000, mean_scale=15000.0, sigma=0.25, max_len=40000, err=(0.001, 0.001, 0.001))
        write_fastq(fq, rs)
        n_bases = len(rs.bases)
    else:
        table = ontsim.ReadTable(seed=37, genome_len=6_500_000, target_bases=105_000_000)
        n_bases = ontsim.write_fastq(table, fq)
    assert n_bases >= 100_000_000
    ref_arc, my_arc, ref_out, my_out = (str(tmp_path / x) for x in ("ref.colord", "gpu.colord", "ref.fastq", "gpu.fastq"))
    subprocess.check_call([REF, mode, "-t", str(os.cpu_count() or 8)] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, mode, "--chunk-bases", "25000000"] + extra + [fq, my_arc])   # four or five chunks
    _same_streams(ref_arc, my_arc)
    assert len(AR.read_archive(my_arc)["dna"].parts) >= 20
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(ref_out)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cli_part_symbols_archive_is_decoded_by_the_reference(tmp_path):
    """`--part-symbols 65536` (the cut bench.py's headline number is measured with): many more, shorter coder parts — any cut at read
    boundaries is a valid CoLoRd archive (entr_read.h:146-191, entr_qual.h:150-170: the decoders follow the part table).  The
    unmodified reference and this build's decompressor both return what the reference returns for its own archive; the archive is
    at most 0.2 % larger; the default (4194304, defs.h:45) stays the reference's archive byte for byte; the indexed reader of the plain
    FASTQ (several threads) and the sequential one give the same archive."""
    from colord_amd import ontsim
    table = ontsim.ReadTable(seed=43, genome_len=4_000_000, target_bases=60_000_000)
    fq = str(tmp_path / "in.fastq")
    ontsim.write_fastq(table, fq)
    ref_arc, ref_out, a64, out_ref, out_own, dflt, seq = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "p64k.colord", "p64k_ref.fastq", "p64k_own.fastq", "default.colord", "seq.colord"))
    subprocess.check_call([REF, "compress-ont", "-t", str(os.cpu_count() or 8), fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "20000000", "--part-symbols", "65536", fq, a64])
    subprocess.check_call([REF, "decompress", a64, out_ref], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "decompress", a64, out_own])
    assert sha(out_ref) == sha(ref_out) and sha(out_own) == sha(ref_out)
    a, b = AR.read_archive(ref_arc), AR.read_archive(a64)
    assert len(b["dna"].parts) > 16 * len(a["dna"].parts) and len(b["qual"].parts) == len(b["dna"].parts)
    assert os.path.getsize(a64) <= os.path.getsize(ref_arc) * 1.002
    for name in ("header", "meta"):
        assert [p for _, p in a[name].parts] == [p for _, p in b[name].parts]
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "20000000", fq, dflt])
    _same_streams(ref_arc, dflt)
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "20000000", "--part-symbols", "65536", "--parse-threads", "1", fq, seq])
    _same_streams(a64, seq)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("extra", [[], ["-p", "ratio"]], ids=["default", "ratio"])
def test_cli_giant_gaps_equal_the_reference(tmp_path, extra):
    """Giant gaps — tens of thousands of symbols on both sides, inner gaps and flanks, at several recursion levels — go through the
    tile-job aligner (csrc/align_giant.hpp: the tiles of a sweep as a pipeline of waves anywhere on the device, Hirschberg level by
    level over all giants, edlib.cpp:1164-1400).  Random reads yield a few such gaps per Gbase; this input (synth.make_giant_gap_reads)
    is made of them.  Every stream but `info` equals the unmodified reference's, all giants are finished by the tile jobs (none falls
    back to the wave-per-gap kernel), and both decompressors return the reference's output."""
    from colord_amd.synth import make_giant_gap_reads
    rs = make_giant_gap_reads(seed=53)
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    ref_arc, my_arc, ref_out, my_out = (str(tmp_path / x) for x in ("ref.colord", "gpu.colord", "ref.fastq", "gpu.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", str(os.cpu_count() or 8)] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([CLI, "compress-ont", "--chunk-bases", "6000000"] + extra + [fq, my_arc], capture_output=True, text=True, env=dict(os.environ, COLORD_HIP_GAP_DEBUG="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    giants = [l for l in r.stderr.splitlines() if "giant gaps by tile jobs" in l]
    n_giants = sum(int(l.split(":")[1].split()[0]) for l in giants)
    n_back = sum(int(l.split(",")[-1].split()[0]) for l in giants)
    assert n_giants >= 10 and n_back == 0, giants
    _same_streams(ref_arc, my_arc)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "decompress", my_arc, my_out])
    assert sha(my_out) == sha(ref_out)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cli_independent_domains_decode_side_by_side(tmp_path):
    """`--domains K`: K independent model domains (own k-mer statistics, reference reads, index and models each; `hipdomains` with bit 31
    of its count set and a sparse range per domain).  `colord_hip decompress` decodes them with K workers into files of their own and
    joins them; one worker at a time and the sequential record stream behind the C++ API (tests/tools/api_dump) give the same bytes —
    what the reference returns for its own archive of the same FASTQ (4-avg qualities are quantised per read)."""
    from colord_amd import ontsim
    table = ontsim.ReadTable(seed=47, genome_len=4_000_000, target_bases=80_000_000)
    fq = str(tmp_path / "in.fastq")
    ontsim.write_fastq(table, fq)
    ref_arc, ref_out, one, dom, out_par, out_seq = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "one.colord", "dom.colord", "par.fastq", "seq.fastq"))
    subprocess.check_call([REF, "compress-ont", "-t", str(os.cpu_count() or 8), fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", fq, one])
    subprocess.check_call([CLI, "compress-ont", "--domains", "4", "--chunk-bases", "10000000", fq, dom])
    a = AR.read_archive(dom)
    hd = a["hipdomains"].parts[0][1]
    assert int.from_bytes(hd[:4], "little") == (4 | 0x80000000) and len(hd) == 4 + 4 * 16 + 4 * 4
    r = subprocess.run([CLI, "decompress", dom, out_par], capture_output=True, text=True)
    assert r.returncode == 0 and "4 independent domains" in r.stderr, r.stderr
    assert sha(out_par) == sha(ref_out)
    subprocess.check_call([CLI, "decompress", "-t", "1", dom, out_seq])
    assert sha(out_seq) == sha(ref_out)
    # the price: every domain has a quarter of the coverage to find reference reads in (here 5x per domain instead of 20x: `dna` grows by
    # three quarters; `qual`, whose models only start anew, by 0.5 %) — independent domains are for inputs whose coverage can afford them
    b1 = AR.read_archive(one)
    assert sum(len(p) for _, p in a["qual"].parts) <= sum(len(p) for _, p in b1["qual"].parts) * 1.02
    assert os.path.getsize(dom) <= os.path.getsize(one) * 1.5
    # the sequential reader of the public API (include/colord_api.h) walks the domains one after the other
    from tests.test_api_cpu import build
    dump = build(os.path.join(ROOT, "tests", "tools", "api_dump.cpp"), str(tmp_path / "api_dump"))
    r = subprocess.run([dump, dom], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert hashlib.sha256(r.stdout).hexdigest() == sha(ref_out)


@pytest.mark.skipif(not os.path.exists(CLI), reason="needs colord_amd/colord_hip")
def test_cli_stream_input_writes_the_same_archive(tmp_path):
    """`--stream-input`: bounded device memory — the input is read three times (k-mers; reference reads; coding) and a chunk leaves HBM after
    every pass; in the coding pass a loader thread keeps a window of four chunks resident ahead of the coders (the reference reads its
    file twice for the same reason, compression.cpp:432,547-561).  Six / twelve chunks here.  Same archive (every stream but `info`, which
    holds the command line) from the indexed reader of a plain FASTQ, from the sequential one, and from a gzip file re-read through zlib."""
    import gzip, shutil
    from colord_amd import ontsim
    table = ontsim.ReadTable(seed=47, genome_len=4_000_000, target_bases=60_000_000)
    fq = str(tmp_path / "in.fastq")
    ontsim.write_fastq(table, fq)
    base, st_idx, st_seq, st_gz, st_small = (str(tmp_path / x) for x in ("base.colord", "st_idx.colord", "st_seq.colord", "st_gz.colord", "st_small.colord"))
    common = ["compress-ont", "--chunk-bases", "10000000", "--part-symbols", "65536"]
    subprocess.check_call([CLI] + common + [fq, base])
    subprocess.check_call([CLI] + common + ["--stream-input", fq, st_idx])
    _same_streams(base, st_idx)
    subprocess.check_call([CLI] + common + ["--stream-input", "--parse-threads", "1", fq, st_seq])
    _same_streams(base, st_seq)
    gz = fq + ".gz"
    with open(fq, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
        shutil.copyfileobj(f, g)
    subprocess.check_call([CLI] + common + ["--stream-input", gz, st_gz])
    _same_streams(base, st_gz)
    # more chunks than the window several times over, reference cut
    base5, st5 = str(tmp_path / "base5.colord"), str(tmp_path / "st5.colord")
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "5000000", fq, base5])
    subprocess.check_call([CLI, "compress-ont", "--chunk-bases", "5000000", "--stream-input", fq, st5])
    _same_streams(base5, st5)
    out = str(tmp_path / "out.fastq")
    subprocess.check_call([CLI, "decompress", st5, out])
    assert sha(out) == sha(fq)
