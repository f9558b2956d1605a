"""GPU parity suite (stages a1-a7): the HIP path, called through the C ABI, against the oracle and the
golden vectors of the unmodified reference.  Bit-exact everywhere (integer/index work)."""
import numpy as np
import pytest
import torch
from oracle import pyoracle as O
from util import PLAIN_CONFIGS, golden

pytestmark = pytest.mark.gpu


def _u64(t):
    return t.cpu().numpy().view(np.uint64)


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def test_radix_sort_matches_torch(ctx):
    gen = torch.Generator().manual_seed(5)
    for n in (0, 1, 63, 4097, 1_000_003):
        keys = torch.randint(0, 1 << 40, (n,), generator=gen, dtype=torch.int64)
        vals = torch.arange(n, dtype=torch.int32)
        dk, dv = keys.to(ctx.device), vals.to(ctx.device)
        ctx.sort_u64(dk, dv, 0, 40)
        ek, order = torch.sort(keys, stable=True)
        assert torch.equal(dk.cpu(), ek)
        assert torch.equal(dv.cpu(), vals[order])          # stability
        dk2 = keys.to(ctx.device)
        ctx.sort_u64(dk2, None, 0, 40)
        assert torch.equal(dk2.cpu(), ek)


@pytest.mark.parametrize("bits", [27, 32, 40])          # digits of 9, 8 and 10 bits
def test_radix_sort_grouped_histogram_tiles(ctx, bits):
    """From 4096 tiles on a histogram block counts 8 or 16 consecutive tiles (sort.hip k_sort_hist<K, DB, G>); the last group is ragged."""
    n = 4096 * 4096 + 5 * 4096 + 77
    gen = torch.Generator().manual_seed(bits)
    keys = torch.randint(0, 1 << bits, (n,), generator=gen, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int32)
    dk, dv = keys.to(ctx.device), vals.to(ctx.device)
    ctx.sort_u64(dk, dv, 0, bits)
    ek, order = torch.sort(keys, stable=True)
    assert torch.equal(dk.cpu(), ek)
    assert torch.equal(dv.cpu(), vals[order])


@pytest.mark.parametrize("cfg", ["c1_ont_default", "s3m_ont_n_ratio"])
def test_arena_layout(ctx, cfg):
    rs = golden(cfg).reads
    r = ctx.pack_readset(rs)
    assert r.n_reads == rs.n_reads and r.total_bases == len(rs.bases)
    lens = np.diff(rs.offsets)
    assert np.array_equal(_u32(r.lengths()), lens.astype(np.uint32))
    woff = r.word_offsets().cpu().numpy()
    assert np.array_equal(np.diff(woff), lens // 32 + 1)
    assert np.array_equal(r.has_n().cpu().numpy().astype(bool), rs.has_n())
    packed, inv = _u64(r.packed()), _u32(r.invalid())
    for i in (0, 1, rs.n_reads // 2, rs.n_reads - 1):
        b = rs.read(i)
        nw = len(b) // 32 + 1
        pad = np.zeros(nw * 32, np.uint8)
        pad[:len(b)] = b
        bad = pad > 3
        bad[len(b):] = True
        pad[bad] = 0
        w = (pad.reshape(nw, 32).astype(np.uint64) << (62 - 2 * np.arange(32, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)
        iv = (bad.reshape(nw, 32).astype(np.uint64) << (31 - np.arange(32, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
        assert np.array_equal(packed[woff[i]:woff[i] + nw], w)
        assert np.array_equal(inv[woff[i]:woff[i] + nw], iv)
        if not rs.has_n()[i]:                              # a7: CReferenceReads byte image
            exp = np.zeros((len(b) + 3) // 4 + 1, np.uint8)
            O.lib().orc_refread_compact(np.ascontiguousarray(b), len(b), exp)
            assert r.compact(i) == exp.tobytes()
    r.free()


def test_arena_ascii_and_errors(ctx):
    seq = b"ACGTNACGTNAC"
    codes = torch.tensor(list(seq), dtype=torch.uint8)
    off = torch.tensor([0, 5, 12], dtype=torch.int64)
    r = ctx.pack_reads(codes, off, ascii=True)
    assert r.has_n().cpu().tolist() == [1, 1]
    r.free()
    from colord_amd._native import ColordHipError
    for bad in (b"ACGX", b"ACgT"):                             # in_reads.cpp:31-35; lower case has no entry in SymbToBinMap (utils.h:472-475)
        with pytest.raises(ColordHipError, match="Only ACGTN"):
            ctx.pack_reads(torch.tensor(list(bad), dtype=torch.uint8), torch.tensor([0, 4], dtype=torch.int64), ascii=True)
    e = ctx.pack_reads(torch.zeros(0, dtype=torch.uint8), torch.zeros(1, dtype=torch.int64))
    assert e.n_reads == 0 and e.total_words == 0
    assert ctx.kmer_scan(e, 20, 12).numel() == 0
    e.free()


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_kmer_scan_count_filter(ctx, cfg):
    g = golden(cfg)
    k, f = g.p("k"), g.p("f")
    reads = ctx.pack_readset(g.reads)
    km = ctx.kmer_scan(reads, k, f)
    exp = O.kmer_scan_reads(g.reads, k, f)
    assert np.array_equal(np.sort(_u64(km)), np.sort(exp))                  # a1: multiset equality
    kset, st = ctx.count_filter(km.clone(), k, g.p("ci"), g.p("cs"))
    ok, oc, ost = O.count_filter(exp, g.p("ci"), g.p("cs"))
    assert (st.tot_kmers, st.n_unique, st.n_unique_counted, st.total_count_filtered) == \
           (ost.tot_kmers, ost.n_unique, ost.n_unique_counted, ost.total_count_filtered)
    assert st.tot_kmers == g.p("tot_kmers") and st.n_unique_counted == g.p("n_unique")
    assert np.array_equal(_u64(kset.keys()), g.kept[0]) and np.array_equal(_u32(kset.counts()), g.kept[1])
    # a3 membership: every kept key is found, perturbed keys are not
    keys = kset.keys()
    assert bool(kset.check(keys).all())
    others = torch.from_numpy(np.setdiff1d(exp, g.kept[0])[:100000].view(np.int64)).to(ctx.device)
    if others.numel():
        assert not bool(kset.check(others).any())
    kset.free(); reads.free()


def test_kmer_scan_small_k_and_f1(ctx):
    # edge cases: k = 1..28 boundaries, f = 1 (every window survives), reads shorter than k
    rng = np.random.default_rng(3)
    from colord_amd.fastq import ReadSet
    lens = [0, 1, 5, 27, 28, 29, 31, 32, 33, 63, 64, 65, 1000]
    seqs = [rng.integers(0, 4, l, dtype=np.uint8) for l in lens]
    seqs[-1][500] = 4
    rs = ReadSet(np.concatenate(seqs), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), None, [], [], False)
    reads = ctx.pack_readset(rs)
    for k, f in ((1, 1), (5, 1), (20, 1), (28, 1), (28, 3), (15, 7), (21, 40)):
        got = np.sort(_u64(ctx.kmer_scan(reads, k, f)))
        assert np.array_equal(got, np.sort(O.kmer_scan_reads(rs, k, f))), (k, f)
    reads.free()


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_accepted_kmers_and_candidates(ctx, cfg):
    g = golden(cfg)
    k, f, c = g.p("k"), g.p("f"), g.p("c")
    rs = g.reads
    reads = ctx.pack_readset(rs)
    kset, _ = ctx.count_filter(ctx.kmer_scan(reads, k, f), k, g.p("ci"), g.p("cs"))
    lists = ctx.accepted_kmers(kset, reads, k, f)
    off = lists.offsets().cpu().numpy()
    kmers = _u64(lists.kmers())
    ids = _u32(lists.ids())
    keys = g.kept[0]
    assert np.array_equal(keys[ids], kmers)
    has_n = rs.has_n()
    for i in range(rs.n_reads):                                              # a4, per read, order included
        exp = O.accepted_kmers(rs.read(i), k, f, keys)
        assert np.array_equal(kmers[off[i]:off[i + 1]], exp), f"read {i}"
    # a6 + a5
    acc = ctx.ref_accept(g.p("n_reads"), g.p("n_pseudo"), g.p("sparse_range"), g.p("sparse_exp")) if g.p("sparse") else np.ones(rs.n_reads, np.uint8)
    assert np.array_equal(acc, g.accept)
    accept = (acc.astype(bool) & ~has_n).astype(np.uint8)
    index = ctx.index_build(kset, lists, torch.from_numpy(accept), 0, g.p("cs"))
    assert index.n_refs == int(accept.sum())
    refs, votes, cnt = ctx.candidates(index, lists, c)
    refs, votes, cnt = _u32(refs), _u32(votes), _u32(cnt)
    graph = O.Graph(c, g.p("cs"))
    for i in range(rs.n_reads):
        erefs, evotes, _ = graph.next_read(kmers[off[i]:off[i + 1]], bool(accept[i]))
        assert cnt[i] == len(erefs), f"read {i}"
        assert list(refs[i, :cnt[i]]) == list(erefs) == g.cands[i]["refs"], f"read {i}"
        assert list(votes[i, :cnt[i]]) == list(evotes)
        assert (refs[i, cnt[i]:] == 0xffffffff).all()
    if g.p("source") == 2:                                                   # HiFi: shared k-mers in read order
        coff, common = ctx.candidates_common(index, lists, c, torch.from_numpy(refs.view(np.int32)).to(ctx.device),
                                             torch.from_numpy(cnt.view(np.int32)).to(ctx.device))
        coff, common = coff.cpu().numpy(), _u64(common)
        for i in range(rs.n_reads):
            for s, exp in enumerate(g.cands[i]["common"]):
                a, b = coff[i * c + s], coff[i * c + s + 1]
                assert np.array_equal(common[a:b], exp), f"read {i} cand {s}"
    index.free(); lists.free(); kset.free(); reads.free()


def test_full_size_properties(ctx):
    """BASELINE-scale shape (hundreds of Mbases) checked through size-independent properties."""
    from colord_amd.synth_device import make_reads_device
    k, f, ci, cs = 21, 12, 4, 80
    codes, offsets = make_reads_device(ctx.device, seed=11, genome_len=8_000_000, target_bases=200_000_000)
    reads = ctx.pack_reads(codes, offsets)
    assert reads.total_bases == codes.numel()
    km = ctx.kmer_scan(reads, k, f)
    n = km.numel()
    # every survivor passes the modulo test and is canonical-size
    assert int((km >> (2 * k)).abs().max().item()) == 0
    frac = n / reads.total_bases
    assert abs(frac - 1.0 / f) < 0.01 / f * 3 + 0.002
    kset, st = ctx.count_filter(km, k, ci, cs)
    assert st.tot_kmers == n
    keys, counts = kset.keys(), kset.counts()
    assert bool((keys[1:] > keys[:-1]).all())                                # strictly ascending = distinct
    assert int(counts.min().item()) >= ci and int(counts.max().item()) <= cs
    assert int(counts.sum().item()) == st.total_count_filtered
    assert bool(kset.check(keys).all())
    lists = ctx.accepted_kmers(kset, reads, k, f)
    off = lists.offsets()
    assert int(off[-1].item()) == lists.total and bool((off[1:] >= off[:-1]).all())
    ids = lists.ids().long()
    assert torch.equal(keys[ids], lists.kmers())
    # per-read distinctness: (read, id) pairs are unique
    read_of = torch.repeat_interleave(torch.arange(lists.n_reads, device=ctx.device), (off[1:] - off[:-1]))
    key = read_of * kset.size + ids
    assert torch.unique(key).numel() == key.numel()
    acc = ctx.ref_accept(reads.n_reads, 0, 50, 1.0)
    index = ctx.index_build(kset, lists, torch.from_numpy(acc), 0, cs)
    refs, votes, cnt = ctx.candidates(index, lists, 5)
    rank = torch.from_numpy(np.concatenate([[0], np.cumsum(acc)]).astype(np.int64)).to(ctx.device)
    valid = torch.arange(5, device=ctx.device)[None, :] < cnt[:, None]
    assert bool((refs.long()[valid] < rank[:-1, None].expand(-1, 5)[valid]).all())   # only earlier reference reads
    v = votes.long()
    assert bool((v[:, :-1] >= v[:, 1:]).all())                                  # votes descending
    assert float((cnt > 0).float().mean().item()) > 0.5                        # 25x coverage: most reads find neighbours
    index.free(); lists.free(); kset.free(); reads.free()
