"""GPU parity suite for the chunked compressor (csrc/stream.hip, cl_compressor_*): an input presented chunk by chunk —
pass 1 over all chunks, reference listing over all chunks, pass 2 chunk by chunk with persistent coders — must give the
`dna` and `qual` parts of ONE cl_compress_shard call over the whole input, byte for byte (a read sees the
index entries of earlier reference reads only, so cutting the input in file order changes nothing).  cl_compress_shard
itself is pinned to the reference's bytes by test_gpu_encode / test_gpu_roundtrip / test_gpu_cli.

Also here: exact counting of k-mers key range by key range (inputs of 2^32 and more k-mers take that path; the
COLORD_HIP_COUNT_LIMIT knob sends small inputs through it)."""
import os
import numpy as np
import pytest
import torch
from util import golden
from bench import reference_part_bounds

pytestmark = pytest.mark.gpu

PRESET_BY_LEVEL = {1: (64, 3), 2: (48, 5), 3: (48, 6)}


def params_of(g):
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    return dict(k=g.p("k"), f=g.p("f"), ci=g.p("ci"), cs=g.p("cs"), c=g.p("c"), anchor_len=g.p("a"), min_part_alt=min_alt, max_rec=max_rec, min_anchors=1,
                level=g.p("level"), source=g.p("source"), sparse=g.p("sparse"), sparse_g=1.0, sparse_exponent=g.p("sparse_exp"),
                cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)


def one_call(ctx, rs, prm, packs, qual_args):
    reads = ctx.pack_readset(rs)
    dc = ctx.dna_coder(prm["c"], prm["level"], 0)
    qc = ctx.qual_coder(*qual_args) if qual_args else None
    quals = torch.from_numpy(rs.quals).to(ctx.device) if qual_args else None
    qoff = torch.from_numpy(rs.offsets).to(ctx.device) if qual_args else None
    dna, dsz, qual, qsz, info = ctx.compress_shard(reads, prm, packs, packs, dc, qc, quals, qoff)
    out = (dna.cpu().numpy().tobytes(), [int(x) for x in dsz], qual.cpu().numpy().tobytes() if qc else b"", [int(x) for x in qsz] if qc else [], info)
    if qc:
        qc.free()
    dc.free(); reads.free()
    return out


def chunked(ctx, rs, prm, packs, cuts, qual_args, announce=None):
    """cuts: indices into `packs` where chunks start/end (first 0, last len(packs)-1).  announce: None = no look-ahead,
    "all" = every chunk announced before the first encode, "next" = chunk i+1 announced just before chunk i is encoded,
    "late" = only the chunks from the second on, announced after the first was encoded without announcement."""
    off = rs.offsets
    chunks = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(packs[a]), int(packs[b])
        codes = torch.from_numpy(rs.bases[off[r0]:off[r1]])
        o = torch.from_numpy((off[r0:r1 + 1] - off[r0]).astype(np.int64))
        arena = ctx.pack_reads(codes, o)
        q = torch.from_numpy(rs.quals[off[r0]:off[r1]]).to(ctx.device) if qual_args else None
        chunks.append((arena, (np.asarray(packs[a:b + 1]) - packs[a]).astype(np.uint32), q, o.to(ctx.device)))
    cmp_ = ctx.compressor(prm, qual_args, None, None, expected_bases=int(off[-1]))
    for arena, *_ in chunks:
        cmp_.count_add(arena)
    st = cmp_.count_finish()
    for arena, *_ in chunks:
        cmp_.refs_add(arena)
    cmp_.refs_finish()
    dna, dsz, qual, qsz, infos = b"", [], b"", [], []
    if announce == "all":
        for arena, pb, q, o in chunks:
            cmp_.prepare(arena, pb)
    elif announce == "next":
        cmp_.prepare(chunks[0][0], chunks[0][1])
    for i, (arena, pb, q, o) in enumerate(chunks):
        if announce == "next" and i + 1 < len(chunks):
            cmp_.prepare(chunks[i + 1][0], chunks[i + 1][1])
        d, ds, qq, qs, info = cmp_.encode(arena, pb, pb, q, o)
        if announce == "late" and i == 0:
            for a2, pb2, _, _ in chunks[1:]:
                cmp_.prepare(a2, pb2)
        dna += d.cpu().numpy().tobytes(); dsz += [int(x) for x in ds]
        if qual_args:
            qual += qq.cpu().numpy().tobytes(); qsz += [int(x) for x in qs]
        infos.append(info)
    inf = cmp_.info()
    cmp_.free()
    for arena, *_ in chunks:
        arena.free()
    return dna, dsz, qual, qsz, infos, st, inf


def even_cuts(n_packs, n_chunks):
    c = sorted(set(int(round(i * n_packs / n_chunks)) for i in range(n_chunks + 1)))
    return c


@pytest.mark.parametrize("cfg,pack_symbols,n_chunks", [("s6m_ont", 1 << 20, 3), ("s5m_hifi", 1 << 20, 2), ("c3_clr_ratio", 1 << 17, 3), ("s3m_ont_n_ratio", 1 << 19, 4)])
def test_chunked_equals_one_call_on_goldens(ctx, cfg, pack_symbols, n_chunks):
    from oracle import pyoracle as O
    g = golden(cfg)
    rs = g.reads
    prm = params_of(g)
    lens = np.diff(rs.offsets).astype(np.uint32)
    packs = reference_part_bounds(lens, pack_symbols)
    assert len(packs) - 1 >= n_chunks
    qm = g.p("qual_mode")
    qual_args = None
    if rs.quals is not None and len(rs.quals) and qm != 8:
        d = O.QUAL_DEFAULTS[qm]
        qual_args = (qm, g.p("source"), g.p("level"), tuple(d[0]), tuple(d[1]))
    ref = one_call(ctx, rs, prm, packs, qual_args)
    got = chunked(ctx, rs, prm, packs, even_cuts(len(packs) - 1, n_chunks), qual_args)
    assert got[1] == ref[1] and got[0] == ref[0], "dna parts differ"
    assert got[3] == ref[3] and got[2] == ref[2], "qual parts differ"
    assert got[5].tot_kmers == ref[4]["tot_kmers"] and got[5].n_unique_counted == ref[4]["n_kept_kmers"]
    assert got[6]["n_refs_total"] == ref[4]["n_refs"] and got[6]["sparse_range"] == ref[4]["sparse_range"]
    assert sum(i["n_anchors"] for i in got[4]) == ref[4]["n_anchors"]


@pytest.mark.parametrize("cfg,pack_symbols,n_chunks,announce,lanes", [("s6m_ont", 1 << 19, 5, "all", 1), ("s6m_ont", 1 << 19, 5, "next", 1), ("s6m_ont", 1 << 19, 5, "late", 2),
                                                                    ("s5m_hifi", 1 << 20, 3, "all", 2), ("c3_clr_ratio", 1 << 17, 4, "all", 3)])
def test_lookahead_lanes_change_no_byte(ctx, monkeypatch, cfg, pack_symbols, n_chunks, announce, lanes):
    """cl_compressor_prepare: chunks whose candidates / anchors / edit scripts are computed ahead on encode lanes (own contexts,
    own threads) give the bytes of the un-announced run — with every chunk announced up front, one chunk ahead, announcements
    that start late, and several lanes (level 2 / 3 quality coders read the lane's tuple streams)."""
    from oracle import pyoracle as O
    g = golden(cfg)
    rs = g.reads
    prm = params_of(g)
    packs = reference_part_bounds(np.diff(rs.offsets).astype(np.uint32), pack_symbols)
    assert len(packs) - 1 >= n_chunks
    qm = g.p("qual_mode")
    qual_args = None
    if rs.quals is not None and len(rs.quals) and qm != 8:
        d = O.QUAL_DEFAULTS[qm]
        qual_args = (qm, g.p("source"), g.p("level"), tuple(d[0]), tuple(d[1]))
    cuts = even_cuts(len(packs) - 1, n_chunks)
    ref = chunked(ctx, rs, prm, packs, cuts, qual_args)
    monkeypatch.setenv("COLORD_HIP_ENCODE_LANES", str(lanes))
    got = chunked(ctx, rs, prm, packs, cuts, qual_args, announce=announce)
    assert got[1] == ref[1] and got[0] == ref[0], "dna parts differ"
    assert got[3] == ref[3] and got[2] == ref[2], "qual parts differ"
    assert [i["n_anchors"] for i in got[4]] == [i["n_anchors"] for i in ref[4]] and [i["tuple_bytes"] for i in got[4]] == [i["tuple_bytes"] for i in ref[4]]


def test_prepare_rejects_wrong_order(ctx):
    from colord_amd import _native as N
    g = golden("s6m_ont")
    rs = g.reads
    prm = params_of(g)
    packs = reference_part_bounds(np.diff(rs.offsets).astype(np.uint32), 1 << 20)
    off = rs.offsets
    cuts = even_cuts(len(packs) - 1, 2)
    chunks = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(packs[a]), int(packs[b])
        o = torch.from_numpy((off[r0:r1 + 1] - off[r0]).astype(np.int64))
        chunks.append((ctx.pack_reads(torch.from_numpy(rs.bases[off[r0]:off[r1]]), o), (np.asarray(packs[a:b + 1]) - packs[a]).astype(np.uint32)))
    cmp_ = ctx.compressor(prm, None, None, None, expected_bases=int(off[-1]))
    with pytest.raises(N.ColordHipError):
        cmp_.prepare(chunks[0][0], chunks[0][1])                # before pass 1 / the reference listing
    for a, _ in chunks:
        cmp_.count_add(a)
    cmp_.count_finish()
    for a, _ in chunks:
        cmp_.refs_add(a)
    cmp_.refs_finish()
    if chunks[0][0].n_reads != chunks[1][0].n_reads:
        with pytest.raises(N.ColordHipError):
            cmp_.prepare(chunks[1][0], chunks[1][1])            # the second chunk in the first position
    cmp_.prepare(chunks[0][0], chunks[0][1])
    with pytest.raises(N.ColordHipError):
        cmp_.encode(chunks[1][0], chunks[1][1], chunks[1][1])   # not the announced chunk
    cmp_.free()
    for a, _ in chunks:
        a.free()


def test_chunked_equals_one_call_200_mbases(ctx):
    """A synthetic ONT set of 200 Mbases (~13 k reads, 48 reader packs of 4 Mi symbols) in 3 chunks.
    (COLORD_TEST_CHUNKED_MBASES: another size — the run under the debugging modes below, where every hand-over waits for the device.)"""
    from colord_amd.synth_device import make_reads_device
    mb = int(os.environ.get("COLORD_TEST_CHUNKED_MBASES", "200"))
    codes, offsets, quals = make_reads_device(ctx.device, seed=77, genome_len=12_000_000 * mb // 200, target_bases=mb * 1_000_000, with_quals=True)
    lens = (offsets[1:] - offsets[:-1]).cpu().numpy().astype(np.uint32)
    packs = reference_part_bounds(lens, 1 << 22)
    prm = dict(k=21, f=12, ci=4, cs=80, c=5, anchor_len=18, min_part_alt=64, max_rec=3, min_anchors=1, level=1, source=0, sparse=1,
               sparse_g=1.0, sparse_exponent=1.0, cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)
    qa = (2, 0, 1, (7, 14, 26), ())
    reads = ctx.pack_reads(codes, offsets)
    dc, qc = ctx.dna_coder(5, 1, 0), ctx.qual_coder(*qa)
    dna, dsz, qual, qsz, info = ctx.compress_shard(reads, prm, packs, packs, dc, qc, quals, offsets)
    ref = (dna.cpu().numpy().tobytes(), [int(x) for x in dsz], qual.cpu().numpy().tobytes(), [int(x) for x in qsz])
    qc.free(); dc.free(); reads.free()
    assert info["n_anchors"] > 100_000 and info["n_refs"] > 100
    cuts = even_cuts(len(packs) - 1, 3)
    off_h = offsets.cpu().numpy()
    cmp_ = ctx.compressor(prm, qa, None, None, expected_bases=int(off_h[-1]))
    chunks = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        r0, r1 = int(packs[a]), int(packs[b])
        o = (offsets[r0:r1 + 1] - offsets[r0]).contiguous()
        chunks.append((ctx.pack_reads(codes[int(off_h[r0]):int(off_h[r1])], o), (packs[a:b + 1] - packs[a]).astype(np.uint32), quals[int(off_h[r0]):int(off_h[r1])].contiguous(), o))
    for ch in chunks:
        cmp_.count_add(ch[0])
    cmp_.count_finish()
    for ch in chunks:
        cmp_.refs_add(ch[0])
    cmp_.refs_finish()
    got_d, got_ds, got_q, got_qs = b"", [], b"", []
    for arena, pb, q, o in chunks:                              # (look-ahead on: the product's way of driving pass 2)
        cmp_.prepare(arena, pb)
    for arena, pb, q, o in chunks:
        d, ds, qq, qs, _ = cmp_.encode(arena, pb, pb, q, o)
        got_d += d.cpu().numpy().tobytes(); got_ds += [int(x) for x in ds]
        got_q += qq.cpu().numpy().tobytes(); got_qs += [int(x) for x in qs]
    cmp_.free()
    for ch in chunks:
        ch[0].free()
    assert got_ds == ref[1] and got_d == ref[0]
    assert got_qs == ref[3] and got_q == ref[2]


def test_count_by_key_ranges_equals_one_sort(ctx, monkeypatch):
    """cl_kmer_count_filter above its one-sort limit: histogram of the top key bits, gather + sort + count per key range."""
    g = golden("s6m_ont")
    reads = ctx.pack_readset(g.reads)
    km = ctx.kmer_scan(reads, g.p("k"), g.p("f"))
    a, sa = ctx.count_filter(km.clone(), g.p("k"), g.p("ci"), g.p("cs"))
    monkeypatch.setenv("COLORD_HIP_COUNT_LIMIT", "60000")          # ~0.5 M k-mers -> about ten key ranges
    b, sb = ctx.count_filter(km.clone(), g.p("k"), g.p("ci"), g.p("cs"))
    monkeypatch.delenv("COLORD_HIP_COUNT_LIMIT")
    assert (sa.tot_kmers, sa.n_unique, sa.n_unique_counted, sa.total_count_filtered) == (sb.tot_kmers, sb.n_unique, sb.n_unique_counted, sb.total_count_filtered)
    assert sb.n_unique_counted == len(g.kept[0])
    assert torch.equal(a.keys(), b.keys()) and torch.equal(a.counts(), b.counts())
    assert np.array_equal(b.keys().cpu().numpy().view(np.uint64), g.kept[0]) and np.array_equal(b.counts().cpu().numpy().view(np.uint32), g.kept[1])
    probe = torch.cat([b.keys()[::7], b.keys()[::5] + 1])
    assert torch.equal(a.check(probe), b.check(probe))
    a.free(); b.free(); reads.free()


@pytest.mark.parametrize("env", [{"COLORD_HIP_POOL_POISON": "1"}, {"COLORD_HIP_SYNC_DEBUG": "1"}])
def test_pipeline_under_pool_poison_and_sync_debug(env):
    """The six-context pipeline under its two debugging modes (both read once per process, hence a process of their own):
    COLORD_HIP_POOL_POISON fills every block with a pattern when it goes back to the shared pool (on the releasing context's main stream)
    and checks the pattern when the block is handed to another context — a stream that still reads a released block reads the pattern and
    the byte comparisons of the two tests fail; a write after release is reported as `POOL POISON`.  COLORD_HIP_SYNC_DEBUG waits after
    every launch (no overlap at all: what differs from the normal run is a race).  The chunked test runs at 80 Mbases here (19 reader packs in
    3 chunks; at 200 Mbases the two modes took 250 s of the suite's 900)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_stream.py"), "-x", "-q", "-k", "chunked_equals_one_call_200_mbases or lookahead_lanes_change_no_byte"],
                       capture_output=True, text=True, env=dict(os.environ, COLORD_TEST_CHUNKED_MBASES="80", **env), cwd=root, timeout=2400)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "POOL POISON" not in r.stderr and "POOL POISON" not in r.stdout
