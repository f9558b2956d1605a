"""Multi-GPU product path (colord_amd/mgpu.py + cl_exchange callbacks of csrc/stream.hip) on ONE GPU: the ranks share the device
and the collectives run over gloo (same code path as RCCL: TorchExchange only changes how bytes move).
  * one rank: the archive is the reference's, stream by stream;
  * two ranks: the k-mer set, the reference reads and the index are replicated through the exchanges, every rank is one model
    domain; `colord_hip decompress` returns the input, and the archive stays within 1 % of the one-rank archive."""
import hashlib
import json
import os
import subprocess
import sys
import numpy as np
import pytest
from colord_amd import archive as AR, ontsim
from colord_amd.fastq import read_fastx, write_fastq
from colord_amd.synth import make_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def run_ranks(n, args, port):
    env = dict(os.environ, COLORD_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "-m", "colord_amd.mgpu"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


def test_one_rank_archive_is_the_reference_archive(tmp_path):
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "s6m_ont", "streams.json")))
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, make_reads(**spec["synth"]))
    from colord_amd.mgpu import compress_readset
    arc = str(tmp_path / "one.colord")
    res = compress_readset(read_fastx(fq), arc, est_bases=0.49 * os.path.getsize(fq), device=0)
    assert res["dna_parts"] == len(spec["streams"]["dna"]["parts"])
    a = AR.read_archive(arc)
    assert "hipdomains" not in a
    for name in ("meta", "dna", "qual", "header"):
        assert [[m, len(p), hashlib.sha256(p).hexdigest()] for m, p in a[name].parts] == spec["streams"][name]["parts"], name


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_ranks_on_golden_decode_to_the_reference_output(tmp_path, n_ranks):
    """4-avg qualities are quantised per read, so the decoded FASTQ of a multi-domain archive must be exactly what the reference
    returns for its own archive of the same input."""
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "s6m_ont", "streams.json")))
    fq, arc, out = (str(tmp_path / x) for x in ("in.fastq", "r.colord", "r.fastq"))
    write_fastq(fq, make_reads(**spec["synth"]))
    run_ranks(n_ranks, ["compress-ont", "--chunk-bases", "1.5e6", fq, arc], 29610 + n_ranks)
    a = AR.read_archive(arc)
    assert "hipdomains" in a and int.from_bytes(a["hipdomains"].parts[0][1][:4], "little") == n_ranks
    subprocess.check_call([CLI, "decompress", arc, out])
    assert sha(out) == spec["decompressed_sha256"]
    # same reads coded against references, same candidates: the tuple streams do not depend on the sharding; only the coders' model domains do
    ref_total = sum(p[1] for p in spec["streams"]["dna"]["parts"]) + sum(p[1] for p in spec["streams"]["qual"]["parts"])
    got_total = sum(len(p) for _, p in a["dna"].parts) + sum(len(p) for _, p in a["qual"].parts)
    assert got_total < ref_total * 1.08                      # 6 Mbases in 2-3 domains: the loss shrinks with the domain size (next test)


def test_two_ranks_200_mbases_lossless_and_within_one_percent(tmp_path):
    t = ontsim.ReadTable(seed=11, genome_len=12_000_000, target_bases=200_000_000)
    fq, one, two, out = (str(tmp_path / x) for x in ("in.fastq", "one.colord", "two.colord", "two.fastq"))
    ontsim.write_fastq(t, fq)
    subprocess.check_call([CLI, "compress-ont", "-q", "org", fq, one])
    log = run_ranks(2, ["compress-ont", "-q", "org", "--chunk-bases", "4e7", fq, two], 29620)
    subprocess.check_call([CLI, "decompress", two, out])
    assert sha(out) == sha(fq)                               # -q org: bit-exact round trip
    s1, s2 = os.path.getsize(one), os.path.getsize(two)
    print(f"one rank {s1} B, two ranks {s2} B: {100.0 * (s2 / s1 - 1):+.3f} %; {log.strip().splitlines()[-1]}")
    assert s2 <= s1 * 1.01
    a, b = AR.read_archive(one), AR.read_archive(two)
    assert [m for m, _ in a["dna"].parts] != [] and sum(m for m, _ in a["dna"].parts) == sum(m for m, _ in b["dna"].parts) == t.n_reads


def test_rccl_initialises_and_carries_the_exchange_callbacks_on_one_rank(tmp_path):
    """backend "nccl" (= RCCL) with world = 1: the process group comes up on the GPU and the three cl_exchange callbacks of
    colord_amd.parallel.TorchExchange (all_gather_host, all_to_all_v, all_gather_v) move device buffers through it — the same code
    path as on 8 GPUs, minus the peers.  (Multi-rank behaviour is covered with gloo above; RCCL over xGMI needs the 8-GPU node.)"""
    script = tmp_path / "rccl_self_test.py"
    script.write_text('''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["COLORD_ROOT"])
from colord_amd import parallel as par
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
X = par.TorchExchange(torch.device("cuda", 0))
U64P = C.POINTER(C.c_uint64)
vals = np.arange(5, dtype=np.uint64); out = np.zeros(5, np.uint64)
assert X._gather_host(None, vals.ctypes.data_as(U64P), 5, out.ctypes.data_as(U64P)) == 0 and (out == vals).all()
src = torch.arange(1001, dtype=torch.uint8, device="cuda"); dst = torch.zeros(1001, dtype=torch.uint8, device="cuda")
sb = np.array([1001], np.uint64)
assert X._all_to_all_v(None, src.data_ptr(), sb.ctypes.data_as(U64P), dst.data_ptr(), sb.ctypes.data_as(U64P)) == 0, X.err
assert torch.equal(src, dst)
dst.zero_()
assert X._all_gather_v(None, src.data_ptr(), 1001, dst.data_ptr(), sb.ctypes.data_as(U64P)) == 0, X.err
assert torch.equal(src, dst)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
assert par.gather_to_root(src)[0] is src
dist.destroy_process_group()
print("rccl self test ok")
''')
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", HSA_ENABLE_IPC_MODE_LEGACY="0", COLORD_ROOT=ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "rccl self test ok" in r.stdout, r.stderr[-2000:]


def test_bad_options_are_refused_before_the_process_group(tmp_path):
    from colord_amd import mgpu
    fq = tmp_path / "x.fastq"; fq.write_text("@a\nACGT\n+\n!!!!\n")
    assert mgpu.main(["compress-ont", "-p", "fast", str(fq), str(tmp_path / "o")]) == 1
    assert mgpu.main(["compress-ont", "-q", "7-avg", str(fq), str(tmp_path / "o")]) == 1
    assert mgpu.main(["compress-ont", "-k", "40", "-a", "22", str(fq), str(tmp_path / "o")]) == 1
    assert mgpu.main(["compress-ont", str(tmp_path / "missing.fastq"), str(tmp_path / "o")]) == 1


def _m_bovis(tmp_path):
    import gzip
    fq, gen = str(tmp_path / "M.bovis.fastq"), str(tmp_path / "M.bovis-reference.fna")
    open(fq, "wb").write(gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis.fastq.gz"), "rb").read())
    open(gen, "wb").write(gzip.open(os.path.join(ROOT, "tests", "data", "M.bovis-reference.fna.gz"), "rb").read())
    return fq, gen


@pytest.mark.parametrize("stored", [True, False])
def test_reference_genome_mode_one_rank_is_the_reference_archive(tmp_path, stored):
    """`-G genome [-s]` through the multi-GPU driver with one rank: the archive the unmodified reference wrote for config 4
    (tests/golden/archives/c4_ont_genome_*), every stream but `info`."""
    from colord_amd import mgpu
    name = "c4_ont_genome_stored" if stored else "c4_ont_genome_external"
    fq, gen = _m_bovis(tmp_path)
    arc = str(tmp_path / "a.colord")
    assert mgpu.main(["compress-ont", "-G", gen] + (["-s"] if stored else []) + [fq, arc]) == 0
    a, b = AR.read_archive(os.path.join(ROOT, "tests", "golden", "archives", name + ".colord")), AR.read_archive(arc)
    assert set(a) == set(b)
    for s in a:
        if s != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[s].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[s].parts], s


@pytest.mark.parametrize("stored", [True, False])
def test_reference_genome_mode_with_sharded_reads(tmp_path, stored):
    """Two ranks, `-G genome [-s]`: rank 0 counts the genome's k-mers and contributes the pseudo reads to the replicated store, the
    second rank's reads find them (and rank 0's reference reads) as candidates.  The archive decodes to the reference's output
    and — same tuples, two model domains — stays close to the one-rank archive."""
    name = "c4_ont_genome_stored" if stored else "c4_ont_genome_external"
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "archives", "expected.json")))[name]
    fq, gen = _m_bovis(tmp_path)
    arc, out = str(tmp_path / "a.colord"), str(tmp_path / "o.fastq")
    run_ranks(2, ["compress-ont", "-G", gen] + (["-s"] if stored else []) + ["--chunk-bases", "3e6", fq, arc], 29650 + int(stored))
    b = AR.read_archive(arc)
    assert "hipdomains" in b and ("ref-genome" in b) == stored
    subprocess.check_call([CLI, "decompress"] + ([] if stored else ["-G", gen]) + [arc, out])
    assert sha(out) == exp["decompressed_sha256"]
    a = AR.read_archive(os.path.join(ROOT, "tests", "golden", "archives", name + ".colord"))
    assert a["meta"].parts[0][1] == b["meta"].parts[0][1]                # pseudo-read geometry, reference count, checksum
    one = sum(len(p) for _, p in a["dna"].parts)
    assert sum(len(p) for _, p in b["dna"].parts) < one * 1.5           # a genome-less second rank would lose far more than a model restart


# ---- the C++ multi-GPU host: `colord_hip compress-* --gpus N` (csrc/cli/compress.cpp run_compress_multi, cli/transport.hpp) ------------
def test_cpp_rccl_transport_selftest():
    """The three collectives behind cl_exchange over RCCL directly (ncclCommInitAll, grouped send / recv): a communicator of ONE rank
    on this box's one GPU still runs the RCCL code path — uneven and empty shares, every byte checked."""
    r = subprocess.run([CLI, "rccl-selftest"], capture_output=True, text=True, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=600)
    assert r.returncode == 0 and "ok on 1 rank" in r.stdout, r.stderr[-2000:] + r.stdout[-500:]


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_cpp_host_ranks_write_the_archive_of_the_python_driver(tmp_path, n_ranks):
    """`colord_hip --gpus N --gpu-list 0,0[,0] --transport host` (N rank threads sharing this box's GPU, collectives through pinned host
    staging) against `python -m colord_amd.mgpu` with N gloo ranks on the same gzip FASTQ: same shares, same exchanges, same model
    domains -> every stream but `info` byte-identical, `hipdomains` included (each rank pwrites its own parts; the part tables must
    agree all the same); this build's decompressor returns the reference's output."""
    import gzip
    spec = json.load(open(os.path.join(ROOT, "tests", "golden", "s6m_ont", "streams.json")))
    fq, gz, py_arc, cpp_arc, out = (str(tmp_path / x) for x in ("in.fastq", "in.fastq.gz", "py.colord", "cpp.colord", "cpp.fastq"))
    write_fastq(fq, make_reads(**spec["synth"]))
    with open(fq, "rb") as f, gzip.open(gz, "wb", compresslevel=1) as g:
        g.write(f.read())
    run_ranks(n_ranks, ["compress-ont", "--chunk-bases", "1.5e6", gz, py_arc], 29640 + n_ranks)
    r = subprocess.run([CLI, "compress-ont", "--gpus", str(n_ranks), "--gpu-list", ",".join(["0"] * n_ranks), "--transport", "host", "--chunk-bases", "1.5e6", gz, cpp_arc], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = AR.read_archive(py_arc), AR.read_archive(cpp_arc)
    assert set(a) == set(b) and "hipdomains" in b
    for name in a:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts], name
    subprocess.check_call([CLI, "decompress", cpp_arc, out])
    assert sha(out) == spec["decompressed_sha256"]
    # the same reads from the plain file through the indexed (multi-threaded) reader: the same archive
    cpp2 = str(tmp_path / "cpp_plain.colord")
    r = subprocess.run([CLI, "compress-ont", "-k", "20", "-a", "16", "--gpus", str(n_ranks), "--gpu-list", ",".join(["0"] * n_ranks), "--transport", "host", "--chunk-bases", "1.5e6", fq, cpp2],
                       capture_output=True, text=True, timeout=1500, env=dict(os.environ, COLORD_HIP_INDEX_MIN_BYTES="1000"))
    assert r.returncode == 0, r.stderr[-3000:]
    c = AR.read_archive(cpp2)
    for name in b:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in c[name].parts], name


@pytest.mark.parametrize("stored", [True, False])
def test_cpp_host_reference_genome_mode_with_sharded_reads(tmp_path, stored):
    """`colord_hip compress-ont -G genome [-s] --gpus 2` (two rank threads on this box's GPU, host-staged collectives): every rank is handed
    the genome and the pseudo reads, the library lets rank 0 count the genome's k-mers and contribute the pseudo reads — the archive of
    `python -m colord_amd.mgpu` with two gloo ranks, every stream but `info` (`ref-genome` and the genome fields of `meta` included), and
    the reference's output back from the decompressor."""
    name = "c4_ont_genome_stored" if stored else "c4_ont_genome_external"
    exp = json.load(open(os.path.join(ROOT, "tests", "golden", "archives", "expected.json")))[name]
    import gzip
    fq0, gen = _m_bovis(tmp_path)
    fq = fq0 + ".gz"                                                     # (gzip: both drivers parse the whole file and share the reads by bases)
    with open(fq0, "rb") as f, gzip.open(fq, "wb", compresslevel=1) as g:
        g.write(f.read())
    py_arc, cpp_arc, out = str(tmp_path / "py.colord"), str(tmp_path / "cpp.colord"), str(tmp_path / "o.fastq")
    opts = ["compress-ont", "-G", gen] + (["-s"] if stored else []) + ["--chunk-bases", "3e6"]
    run_ranks(2, opts + [fq, py_arc], 29660 + int(stored))
    r = subprocess.run([CLI] + opts + ["--gpus", "2", "--gpu-list", "0,0", "--transport", "host", fq, cpp_arc], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = AR.read_archive(py_arc), AR.read_archive(cpp_arc)
    assert set(a) == set(b) and ("ref-genome" in b) == stored
    for s in a:
        if s != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[s].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[s].parts], s
    subprocess.check_call([CLI, "decompress"] + ([] if stored else ["-G", gen]) + [cpp_arc, out])
    assert sha(out) == exp["decompressed_sha256"]


REF = os.path.join(ROOT, "oracle", "_ref", "colord")


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
def test_cpp_host_eight_ranks_400_mbases(tmp_path):
    """BASELINE.json's configuration 5 is an EIGHT-way split: `colord_hip --gpus 8 --gpu-list 0,0,0,0,0,0,0,0 --transport host` (eight rank
    threads on this box's one GPU, each a compressor with lanes and preparation threads of its own over the one shared pool) on 400 Mbases
    of the bench's recipe.  The archive has eight model domains, decodes to what the unmodified reference returns for its own archive of
    the same file (4-avg qualities are quantised per read), and stays within 1 % of the one-rank archive — the metric's size budget.
    Then the same with --stream-input (bounded device memory under --gpus N: chunks are uploaded again for each pass): the same archive."""
    t = ontsim.ReadTable(seed=13, genome_len=24_000_000, target_bases=400_000_000)
    fq, ref_arc, ref_out, one, eight, out, eight_s = (str(tmp_path / x) for x in ("in.fastq", "ref.colord", "ref.fastq", "one.colord", "eight.colord", "eight.fastq", "eight_stream.colord"))
    ontsim.write_fastq(t, fq)
    subprocess.check_call([REF, "compress-ont", "-t", str(min(64, os.cpu_count() or 8)), fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, "compress-ont", fq, one])
    many = ["--gpus", "8", "--gpu-list", ",".join(["0"] * 8), "--transport", "host"]
    r = subprocess.run([CLI, "compress-ont"] + many + [fq, eight], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    a = AR.read_archive(eight)
    dom = a["hipdomains"].parts[0][1]
    assert int.from_bytes(dom[:4], "little") == 8 and len(dom) == 4 + 8 * 16
    subprocess.check_call([CLI, "decompress", eight, out])
    assert sha(out) == sha(ref_out)
    s1, s8 = os.path.getsize(one), os.path.getsize(eight)
    print(f"one rank {s1} B, eight ranks {s8} B: {100.0 * (s8 / s1 - 1):+.3f} %; {r.stderr.strip().splitlines()[-1]}")
    assert s8 <= s1 * 1.01
    assert sum(m for m, _ in a["dna"].parts) == t.n_reads
    r = subprocess.run([CLI, "compress-ont", "--stream-input"] + many + [fq, eight_s], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    b = AR.read_archive(eight_s)
    assert set(a) == set(b)
    for s in a:
        if s != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[s].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[s].parts], s


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain python command (the shape of the driver's N = 1 command): bench.py starts its ranks itself —
    here two ranks sharing this box's GPU over gloo — and rank 0 prints the one JSON line with n_gpus = 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BENCH_BACKEND="gloo", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--bases", "2e8", "--chunk-bases", "5e7", "--e2e-bases", "1e8"],
                       capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["parts_digest_stable"]
    assert line["config"]["parallelism"].startswith("reads sharded x2")
    # what makes a first run on N GPUs readable (round 6): the RCCL self-test as a preflight, every rank's own step times and sizes, seconds
    # and bytes per exchange, the price of one model domain per rank
    m = line["multi_gpu"]
    assert m["process_group_ranks"] == 2 and m["backend"] == "gloo" and m["ranks_share_gpus"] is True
    assert m["rccl_selftest"]["rc"] == 0, m["rccl_selftest"]
    assert [r["rank"] for r in m["per_rank"]] == [0, 1] and all(len(r["step_s"]) == 1 and len(r["own_step_s"]) == 1 and r["own_step_s"][0] <= r["step_s"][0] + 1e-3 and r["bases"] > 0 and r["dna_bytes"] > 0 for r in m["per_rank"])
    ex = m["exchange_s"]
    assert ex["kmers.all_to_all_v"]["bytes_received_all_ranks_per_step"] > 0 and ex["kmers.all_gather_v"]["calls_per_step"] >= 1
    assert ex["refs.all_gather_v"]["bytes_received_all_ranks_per_step"] > 0 and "parts.gather_to_root" in ex
    assert "domain_loss_vs_one_rank" in m and m["domain_loss_vs_one_rank"]["stream_bytes_per_base"] > 0
