"""GPU parity suite for the m-mer anchor stage (a8): candidates + anchors as the encoder will use them must equal
the oracle's (which reproduces the reference's tuple streams), fed with the reference's own candidate lists."""
import numpy as np
import pytest
import torch
from oracle import pyoracle as O
from util import golden
from test_gpu_dna import ref_subset

pytestmark = pytest.mark.gpu


def hifi_args(ctx, g, c):
    """(kmer_len, modulo, common_off, common) in the layout of cl_candidates_common, from the reference's own lists."""
    n = g.reads.n_reads
    off = np.zeros(n * c + 1, np.int64)
    vals = []
    for i, e in enumerate(g.cands):
        for j in range(c):
            if j < len(e["refs"]):
                vals.append(np.asarray(e["common"][j], np.uint64))
                off[i * c + j + 1] = len(vals[-1])
    off = np.cumsum(off)
    common = np.concatenate(vals) if vals else np.zeros(0, np.uint64)
    return (g.p("k"), g.p("f"), torch.from_numpy(off).to(ctx.device), torch.from_numpy(common.view(np.int64).copy()).to(ctx.device))


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "s5m_hifi", "c7_hifi_balanced", "c2_hifi_org", "s6m_ont_k25", "s4m_ont_k23_balanced"])
def test_anchor_candidates_equal_oracle(ctx, cfg):
    g = golden(cfg)
    rs = g.reads
    c = g.p("c")
    has_n = rs.has_n()
    accept = g.accept.astype(bool) & ~has_n
    reads = ctx.pack_readset(rs)
    refs = ctx.pack_readset(ref_subset(rs, accept))
    cand = np.full((rs.n_reads, c), 0xffffffff, np.uint32)
    cn = np.zeros(rs.n_reads, np.uint32)
    for i, e in enumerate(g.cands):
        cn[i] = len(e["refs"])
        cand[i, :cn[i]] = e["refs"]
    hifi = g.p("source") == 2
    anc = ctx.anchor_candidates(reads, refs, torch.from_numpy(cand.view(np.int32)).to(ctx.device), torch.from_numpy(cn.view(np.int32)).to(ctx.device), g.p("a"),
                                hifi=hifi_args(ctx, g, c) if hifi else None)
    n_c = anc.n_cands().cpu().numpy()
    tab = anc.cands().cpu().numpy().view(np.uint32)
    off = anc.cand_offsets().cpu().numpy()
    data = anc.data().cpu().numpy().view(np.uint32)
    enc = O.Encoder(g.p("a"), g.p("k"), g.p("f"), g.p("source"))
    for i in range(rs.n_reads):
        if accept[i]:
            enc.add_ref(rs.read(i))
    n_with = 0
    for i in range(rs.n_reads):
        exp = [] if has_n[i] else enc.candidates(rs.read(i), g.cands[i]["refs"], g.cands[i]["common"] if hifi else None)
        assert n_c[i] == len(exp), f"read {i}"
        for j, (rid, rev, tot, anchors) in enumerate(exp):
            assert tuple(tab[i, j]) == (rid, rev, tot, len(anchors)), f"read {i} cand {j}"
            a = off[i * c + j]
            assert [tuple(x) for x in data[a:a + len(anchors)]] == anchors, f"read {i} cand {j}"
        n_with += len(exp) > 0
    assert n_with > (10 if cfg != "c2_hifi_org" else 2)      # the D.melanogaster fixture has 3 coded reads
    anc.free(); refs.free(); reads.free()


def _anchor_arrays(ctx, reads, refs, crefs, cnt, a):
    anc = ctx.anchor_candidates(reads, refs, crefs, cnt, a)
    out = (anc.n_cands().cpu().numpy().copy(), anc.cands().cpu().numpy().copy(), anc.cand_offsets().cpu().numpy().copy(), anc.data().cpu().numpy().copy())
    anc.free()
    return out


@pytest.mark.parametrize("cfg", ["s6m_ont", "s3m_ont_n_ratio", "s5m_hifi", "s6m_ont_k25"])
def test_anchor_candidates_tables_in_lds_equal_oracle(ctx, cfg, monkeypatch):
    """The golden reads are short enough for every m-mer table to live in LDS: with COLORD_HIP_ANCHORS_LDS=24576 all of them go through
    k_match_lds (off by default: DESIGN.md 10) instead of k_table_insert + k_match, and the oracle judges that form too."""
    monkeypatch.setenv("COLORD_HIP_ANCHORS_LDS", "24576")
    test_anchor_candidates_equal_oracle(ctx, cfg)


def test_anchor_tables_in_lds_equal_tables_in_hbm(ctx, monkeypatch):
    """Reads either side of both LDS class bounds (12 288 and 24 576 m-mers) and of the segment rule of k_match, a read whose m-mers
    repeat thousands of times (runs of equal m-mers in the table, more hits in a wave step than the staging holds, the "too many
    matches" veto) and one with an N: candidates and anchors with long reads matched by segments (three segment sizes) and with the
    tables in LDS (both classes; the small class only) == with every table in HBM and one block per read, which the oracle judges on the
    goldens and the reference's bytes on the bench's sample."""
    from colord_amd.fastq import ReadSet
    rng = np.random.default_rng(5)
    a, c = 16, 5
    comp = np.array([3, 2, 1, 0], np.uint8)
    genome = rng.integers(0, 4, 150_000, dtype=np.uint8)
    unit = rng.integers(0, 4, 37, dtype=np.uint8)
    genome[60_000:68_000] = np.resize(unit, 8000)                            # a tandem repeat: 37 distinct m-mers, ~200 times each
    genome[90_000:93_000] = 0                                                # and a homopolymer

    def noisy(s):
        r = rng.random(len(s))
        s = s.copy()
        sub = (r >= 0.02) & (r < 0.05)
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()))) % 4
        s = s[r >= 0.02]
        ins = np.nonzero(rng.random(len(s)) < 0.02)[0]
        return np.insert(s, ins, rng.integers(0, 4, len(ins)).astype(np.uint8))

    lens = [300, 15, 16, 17, 2500, 12_000, 12_287 + a - 1, 12_288 + a - 1, 12_289 + a - 1, 12_600, 20_000, 24_576 + a - 1, 24_577 + a - 1, 30_000, 70_000,
            9000, 23_000, 40_000]
    starts = [int(rng.integers(0, 50_000)) for _ in lens[:-3]] + [58_000, 55_000, 50_000]        # the last three cover the repeat (and the homopolymer)
    seqs = []
    for ln, st in zip(lens, starts):
        s = noisy(genome[st:st + ln + ln // 20 + 8])[:ln]
        assert len(s) == ln
        seqs.append(comp[s[::-1]].copy() if rng.random() < 0.5 else s)
    for ln, st in ((21_000, 40_000), (64_000, 30_000), (8000, 59_000), (26_000, 52_000)):      # a second cover: candidates of every class with every class
        seqs.append(noisy(genome[st:st + ln]))
    seqs.append(seqs[10].copy()); seqs[-1][5000] = 4                          # a read with N: no candidates looked at
    n = len(seqs)
    ln = np.array([len(s) for s in seqs], np.int64)
    rs = ReadSet(np.concatenate(seqs), np.concatenate([[0], np.cumsum(ln)]).astype(np.int64), None, [], [False] * n, False)
    reads = ctx.pack_readset(rs)
    accept = np.ones(n, np.uint8); accept[-1] = 0
    refs = ctx.select_reads(reads, torch.from_numpy(accept).to(ctx.device))
    cand = np.zeros((n, c), np.int32); cn = np.zeros(n, np.int32)
    for i in range(1, n):
        pool = rng.permutation(min(i, n - 1))[:c]
        cn[i] = len(pool); cand[i, :cn[i]] = pool
    crefs, cnt = torch.from_numpy(cand).to(ctx.device), torch.from_numpy(cn).to(ctx.device)
    monkeypatch.delenv("COLORD_HIP_ANCHORS_LDS", raising=False)
    monkeypatch.setenv("COLORD_HIP_MATCH_SEG", "0")                           # one k_match block per read whatever its length: the form of rounds 1-5
    want = _anchor_arrays(ctx, reads, refs, crefs, cnt, a)
    assert want[0].sum() >= 20 and want[3].size > 3000                        # candidates kept, anchors found
    # long reads matched segment by segment (default: 16 384 positions a block; 1000 / 77: many segments for every read but the shortest),
    # and the tables in LDS (with the default segments for the reads above the classes)
    for name_, setting in (("COLORD_HIP_MATCH_SEG", None), ("COLORD_HIP_MATCH_SEG", "1000"), ("COLORD_HIP_MATCH_SEG", "77"), ("COLORD_HIP_ANCHORS_LDS", "24576"), ("COLORD_HIP_ANCHORS_LDS", "12288")):
        monkeypatch.delenv("COLORD_HIP_ANCHORS_LDS", raising=False); monkeypatch.delenv("COLORD_HIP_MATCH_SEG", raising=False)
        if setting is not None:
            monkeypatch.setenv(name_, setting)
        got = _anchor_arrays(ctx, reads, refs, crefs, cnt, a)
        for w, g_, name in zip(want, got, ("n_cands", "cands", "offsets", "anchors")):
            assert np.array_equal(w, g_), f"{name} differ with {name_}={setting}"
    refs.free(); reads.free()
