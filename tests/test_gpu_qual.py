"""GPU parity suite for the quality coder (a13 + a15 + a16): the HIP path must reproduce the reference's
`qual` stream BYTE FOR BYTE (golden payload hashes of the unmodified reference) and equal the oracle on
every mode; model state must persist across calls exactly like one long-lived CQualityCoder."""
import hashlib
import numpy as np
import pytest
import torch
from oracle import pyoracle as O
from util import ALL_CONFIGS, golden

pytestmark = pytest.mark.gpu


def flags_for(g):
    rs = g.reads
    if g.p("level") <= 1:
        return None
    fl = np.concatenate([O.es_flags(g.es[i][2], len(rs.read(i))) for i in range(rs.n_reads)])
    return fl


def gpu_encode(ctx, rs, mode, source, level, bounds, flags=None, fwd=None, split=None):
    d = O.QUAL_DEFAULTS[mode]
    qc = ctx.qual_coder(mode, source, level, d[0] if fwd is None else fwd, d[1])
    reads = ctx.pack_readset(rs)
    quals = torch.from_numpy(rs.quals).to(ctx.device)
    qoff = torch.from_numpy(rs.offsets).to(ctx.device)
    fl = None if flags is None else torch.from_numpy(flags).to(ctx.device)
    parts = []
    calls = [bounds] if split is None else [bounds[:split + 1], bounds[split:]]
    for b in calls:
        if len(b) < 2:
            continue
        out, sizes = qc.encode(reads, quals, qoff, b, fl)
        raw = out.cpu().numpy().tobytes()
        o = 0
        for s in sizes:
            parts.append(raw[o:o + s])
            o += s
    qc.free(); reads.free()
    return parts


def oracle_encode(rs, mode, source, level, bounds, flags=None, fwd=None):
    qc = O.QualCoder(True, mode, source, level, fwd=fwd)
    parts = []
    for pi in range(len(bounds) - 1):
        for i in range(bounds[pi], bounds[pi + 1]):
            f = None if flags is None else flags[rs.offsets[i]:rs.offsets[i + 1]]
            qc.encode(rs.read(i), rs.qual(i), f)
        parts.append(qc.finish_part())
    return parts


@pytest.mark.parametrize("cfg", ALL_CONFIGS)
def test_qual_stream_byte_identical_to_reference(ctx, cfg):
    g = golden(cfg)
    rs = g.reads
    parts = gpu_encode(ctx, rs, g.p("qual_mode"), g.p("source"), g.p("level"), rs.pack_bounds(), flags_for(g))
    exp = g.spec["streams"]["qual"]["parts"]
    assert [[0, len(p), hashlib.sha256(p).hexdigest()] for p in parts] == exp


@pytest.mark.parametrize("mode", sorted(O.QM.values()))
@pytest.mark.parametrize("source,level", [(0, 1), (1, 3), (2, 2)])
def test_every_mode_equals_oracle(ctx, mode, source, level):
    from colord_amd.synth import make_reads
    rs = make_reads(seed=40 + mode, genome_len=30_000, target_bases=300_000, mean_scale=3000.0)
    rng = np.random.default_rng(mode)
    # wide quality alphabet so that every bin / the 96-symbol model is exercised
    rs.quals = (33 + np.clip(rng.normal(20, 12, len(rs.quals)), 0, 93).astype(np.uint8)).astype(np.uint8)
    flags = rng.choice(np.frombuffer(b"AM P", np.uint8), len(rs.quals)) if level > 1 else None
    n = rs.n_reads
    bounds = np.array([0, n // 4, n // 4, n // 2, n], dtype=np.int64)       # ragged parts incl. an empty one
    got = gpu_encode(ctx, rs, mode, source, level, bounds, flags)
    exp = oracle_encode(rs, mode, source, level, bounds, flags)
    assert [len(p) for p in got] == [len(p) for p in exp]
    assert got == exp


def test_model_state_persists_across_calls(ctx):
    g = golden("s6m_ont")
    rs = g.reads
    n = rs.n_reads
    bounds = np.array([0, n // 5, n // 2, n - 7, n], dtype=np.int64)
    one = gpu_encode(ctx, rs, 2, 0, 1, bounds)
    two = gpu_encode(ctx, rs, 2, 0, 1, bounds, split=2)
    assert one == two == oracle_encode(rs, 2, 0, 1, bounds)


def test_rescale_epochs_hot_context(ctx):
    """Long constant-quality reads drive single contexts through many rescale epochs (rc.h:233-244)."""
    from colord_amd.fastq import ReadSet
    rng = np.random.default_rng(9)
    lens = [150_000, 90_000, 1, 2, 3, 64, 65, 200_000]
    bases = np.concatenate([np.full(l, i % 4, np.uint8) for i, l in enumerate(lens)])
    quals = np.concatenate([np.full(l, 33 + (5, 20, 30, 10)[i % 4], np.uint8) for i, l in enumerate(lens)])
    quals[rng.integers(0, len(quals), 2000)] = 33 + 40
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    rs = ReadSet(bases, off, quals, [b"r%d" % i for i in range(len(lens))], [False] * len(lens), True)
    bounds = np.array([0, 2, len(lens)], dtype=np.int64)
    for mode in (0, 2, 3, 5):
        assert gpu_encode(ctx, rs, mode, 0, 1, bounds) == oracle_encode(rs, mode, 0, 1, bounds), mode


@pytest.mark.parametrize("mode", [0, 2, 5, 7])
def test_quality_bytes_outside_phred33_are_refused(ctx, mode):
    """A quality byte outside '!' .. '!' + 95 would index past the coder's 96-entry maps (the reference reads past its tables): the call
    fails, before any model changes — the same coder then codes the clean input as a fresh one does.  (The check rides along in
    k_qual_symbols; mode `none` codes nothing and ignores the bytes, as the reference does.)"""
    from colord_amd._native import ColordHipError
    g = golden("s6m_ont")
    rs = g.reads
    d = O.QUAL_DEFAULTS[mode]
    bounds = np.array([0, rs.n_reads // 2, rs.n_reads], dtype=np.int64)
    reads = ctx.pack_readset(rs)
    qoff = torch.from_numpy(rs.offsets).to(ctx.device)
    good = torch.from_numpy(rs.quals).to(ctx.device)
    qc = ctx.qual_coder(mode, 0, 1, d[0], d[1])
    for pos, val in ((int(rs.offsets[rs.n_reads // 3]) + 5, 31), (int(rs.offsets[-1]) - 1, 33 + 96), (0, 200)):
        bad = good.clone(); bad[pos] = val
        with pytest.raises(ColordHipError, match="quality byte outside"):
            qc.encode(reads, bad, qoff, bounds)
    out, sizes = qc.encode(reads, good, qoff, bounds)
    raw = out.cpu().numpy().tobytes()
    got, o = [], 0
    for s in sizes:
        got.append(raw[o:o + s]); o += s
    qc.free(); reads.free()
    assert got == gpu_encode(ctx, rs, mode, 0, 1, bounds)
