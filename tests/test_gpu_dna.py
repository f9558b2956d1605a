"""GPU parity suite for the DNA coder (a14 + a16): fed with the reference's own tuple streams (golden
es.bin) and reference-read set, the HIP path must reproduce the reference's `dna` stream BYTE FOR BYTE."""
import hashlib
import numpy as np
import pytest
import torch
from oracle import pyoracle as O
from util import PLAIN_CONFIGS, golden
from colord_amd.fastq import ReadSet

pytestmark = pytest.mark.gpu


def ref_subset(rs, accept):
    idx = np.nonzero(accept)[0]
    seqs = [rs.read(i) for i in idx]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    bases = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    return ReadSet(bases, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), None, [], [], False)


def es_arrays(g, device):
    es = g.es
    raw = b"".join(e[2] for e in es)
    off = np.concatenate([[0], np.cumsum([len(e[2]) for e in es])]).astype(np.int64)
    nt = np.array([e[1] for e in es], dtype=np.int32)
    return (torch.from_numpy(np.frombuffer(raw, np.uint8).copy()).to(device), torch.from_numpy(off).to(device), torch.from_numpy(nt).to(device))


def gpu_dna(ctx, g, bounds, split=None):
    rs = g.reads
    accept = g.accept.astype(bool) & ~rs.has_n()
    refs = ctx.pack_readset(ref_subset(rs, accept))
    es, off, nt = es_arrays(g, ctx.device)
    dc = ctx.dna_coder(g.p("c"), g.p("level"), g.p("n_pseudo"))
    parts = []
    calls = [(0, bounds)] if split is None else [(0, bounds[:split + 1]), (bounds[split], bounds[split:])]
    for first, b in calls:
        if len(b) < 2:
            continue
        lo, hi = int(b[0]), int(b[-1])
        sub_off = (off[lo:hi + 1] - off[lo]).contiguous()
        sub_es = es[int(off[lo].item()):int(off[hi].item())].contiguous()
        out, sizes = dc.encode(refs, sub_es, sub_off, nt[lo:hi].contiguous(), np.asarray(b) - lo)
        raw = out.cpu().numpy().tobytes()
        o = 0
        for s in sizes:
            parts.append(raw[o:o + s])
            o += s
    dc.free(); refs.free()
    return parts


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_dna_stream_byte_identical_to_reference(ctx, cfg):
    g = golden(cfg)
    bounds = g.reads.pack_bounds()
    parts = gpu_dna(ctx, g, bounds)
    got = [[int(bounds[i + 1] - bounds[i]), len(p), hashlib.sha256(p).hexdigest()] for i, p in enumerate(parts)]
    assert got == g.spec["streams"]["dna"]["parts"]


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s3m_ont_n_ratio", "s5m_hifi"])
def test_ragged_parts_and_state_across_calls_equal_oracle(ctx, cfg):
    g = golden(cfg)
    rs = g.reads
    n = rs.n_reads
    bounds = np.array([0, 1, 1, n // 3, n // 2, n - 1, n], dtype=np.int64)
    dc = O.DnaCoder(g.p("c"), g.p("level"), g.p("n_pseudo"))
    has_n = rs.has_n()
    for i in range(n):
        if g.accept[i] and not has_n[i]:
            dc.add_ref(rs.read(i))
    exp = []
    for pi in range(len(bounds) - 1):
        for i in range(bounds[pi], bounds[pi + 1]):
            dc.encode(g.es[i][2], g.es[i][1])
        exp.append(dc.finish_part())
    assert gpu_dna(ctx, g, bounds) == exp
    assert gpu_dna(ctx, g, bounds, split=3) == exp


def test_plain_tuple_streams_match_reference(ctx):
    """a12 plain forms: c1 (every read stored plain by the reference) and a set with N reads."""
    for cfg in ("c1_ont_default", "s3m_ont_n_ratio"):
        g = golden(cfg)
        rs = g.reads
        reads = ctx.pack_readset(rs)
        es, off, nt = ctx.encode_plain(reads)
        es, off, nt = es.cpu().numpy().tobytes(), off.cpu().numpy(), nt.cpu().numpy()
        for i in range(rs.n_reads):
            pack, ntup, raw = g.es[i]
            if raw[0] >> 4 in (9, 11):                 # reads the reference stored plain
                assert es[off[i]:off[i + 1]] == raw and nt[i] == ntup, (cfg, i)
        reads.free()


@pytest.mark.parametrize("cfg", ["s6m_ont", "c3_clr_ratio", "s5m_hifi", "s6m_ont_k25", "s4m_ont_k23_balanced"])
@pytest.mark.parametrize("long_run", ["4096", "50000"])
def test_long_context_runs_take_the_parallel_path(ctx, cfg, long_run, monkeypatch):
    """Long runs of one context are evolved by the prefix-count + rescale-chain kernels (k_long_*); with the threshold
    lowered the golden streams run through that path and must still be byte-identical to the reference."""
    monkeypatch.setenv("COLORD_HIP_LONG_RUN", long_run)
    g = golden(cfg)
    bounds = g.reads.pack_bounds()
    parts = gpu_dna(ctx, g, bounds)
    got = [[int(bounds[i + 1] - bounds[i]), len(p), hashlib.sha256(p).hexdigest()] for i, p in enumerate(parts)]
    assert got == g.spec["streams"]["dna"]["parts"]


@pytest.mark.parametrize("cfg", ["s6m_ont", "c3_clr_ratio", "s5m_hifi", "s6m_ont_k25", "s4m_ont_k23_balanced"])
@pytest.mark.parametrize("chunk,warm", [("32", "16"), ("64", "0"), ("256", "4"), ("1000", "128")])
def test_walk_chunks_resume_the_same_walk(ctx, cfg, chunk, warm, monkeypatch):
    """The write pass of the tuple walk resumes from states the count pass saved every `chunk` tuples, and the count pass
    keeps the symbol history only over the `warm` tuples before a saved state (a read is walked again with the history
    throughout when that was not enough: always with warm = 0).  Small chunks put many resumes into the golden streams,
    which must still be byte-identical to the reference."""
    monkeypatch.setenv("COLORD_HIP_WALK_CHUNK", chunk)
    monkeypatch.setenv("COLORD_HIP_WALK_WARM", warm)
    g = golden(cfg)
    bounds = g.reads.pack_bounds()
    parts = gpu_dna(ctx, g, bounds)
    got = [[int(bounds[i + 1] - bounds[i]), len(p), hashlib.sha256(p).hexdigest()] for i, p in enumerate(parts)]
    assert got == g.spec["streams"]["dna"]["parts"]
