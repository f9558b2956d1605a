"""GPU parity suite for the edit-script encoder (a10-a12): the tuple streams cl_encode_reads produces from the
reference's candidate lists must equal, byte for byte, the streams tapped from the unmodified reference
(golden es.bin) — gap alignment with edlib's tie-breaking, indel canonicalisation, static and adaptive (per reader
pack) cost decisions, recursion into alternative references, tuple run-length rules."""
import numpy as np
import pytest
import torch
from oracle import pyoracle as O
from util import golden
from test_gpu_dna import ref_subset

pytestmark = pytest.mark.gpu

# presets (arg_parse.cpp:89-408 / SURVEY App. B): min part length to consider an alternative read, max recursion
PRESET_BY_LEVEL = {1: (64, 3), 2: (48, 5), 3: (48, 6)}


def gpu_streams(ctx, g, pack_bounds=None):
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
    from test_gpu_anchors import hifi_args
    anc = ctx.anchor_candidates(reads, refs, torch.from_numpy(cand.view(np.int32)).to(ctx.device), torch.from_numpy(cn.view(np.int32)).to(ctx.device), g.p("a"),
                                hifi=hifi_args(ctx, g, c) if g.p("source") == 2 else None)
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    es, off, nt = ctx.encode_reads(reads, refs, anc, g.p("a"), min_alt, max_rec, 1.0, rs.pack_bounds() if pack_bounds is None else pack_bounds)
    es, off, nt = es.cpu().numpy(), off.cpu().numpy(), nt.cpu().numpy()
    anc.free(); refs.free(); reads.free()
    return es, off, nt


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "c1_ont_default", "c6_ont_org", "s5m_hifi", "c7_hifi_balanced", "c2_hifi_org", "s6m_ont_k25", "s4m_ont_k23_balanced"])
def test_tuple_streams_equal_reference(ctx, cfg):
    g = golden(cfg)
    es, off, nt = gpu_streams(ctx, g)
    n_es = 0
    bad = []
    for i in range(g.reads.n_reads):
        got = es[off[i]:off[i + 1]].tobytes()
        if nt[i] != g.es[i][1] or got != g.es[i][2]:
            bad.append(i)
        n_es += len(got) > 0 and got[0] >> 4 == 10
    assert not bad, f"{len(bad)} of {g.reads.n_reads} reads differ, first {bad[:10]}"
    if cfg in ("c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "s5m_hifi", "c7_hifi_balanced", "s6m_ont_k25", "s4m_ont_k23_balanced"):
        assert n_es > 10                                    # the edit-script path is really exercised


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "c1_ont_default", "s5m_hifi", "c7_hifi_balanced", "c2_hifi_org", "s6m_ont_k25", "s4m_ont_k23_balanced"])
def test_whole_dna_path_byte_identical_to_reference(ctx, cfg):
    """Read bases (and qualities) in, `dna` and `qual` stream parts out, every stage on the GPU (a1-a16): the parts must
    have the sizes and SHA-256 of the parts the unmodified reference wrote for the same file."""
    import hashlib
    g = golden(cfg)
    rs = g.reads
    k, f, c = g.p("k"), g.p("f"), g.p("c")
    reads = ctx.pack_readset(rs)
    kset, st = ctx.count_filter(ctx.kmer_scan(reads, k, f), k, g.p("ci"), g.p("cs"))
    lists = ctx.accepted_kmers(kset, reads, k, f)
    acc = ctx.ref_accept(g.p("n_reads"), g.p("n_pseudo"), g.p("sparse_range"), g.p("sparse_exp")) if g.p("sparse") else np.ones(rs.n_reads, np.uint8)
    accept = torch.from_numpy(acc.copy()).to(ctx.device) & (reads.has_n() == 0).to(torch.uint8)
    index = ctx.index_build(kset, lists, accept, 0, g.p("cs"))
    crefs, votes, cnt = ctx.candidates(index, lists, c)
    refs = ctx.select_reads(reads, accept)
    assert refs.n_reads == int(accept.sum().item())
    hifi = None
    if g.p("source") == 2:                                  # HiFi: the graph also delivers the shared k-mers of every candidate
        coff, common = ctx.candidates_common(index, lists, c, crefs, cnt)
        hifi = (k, f, coff, common)
    anc = ctx.anchor_candidates(reads, refs, crefs, cnt, g.p("a"), hifi=hifi)
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    bounds = rs.pack_bounds()
    es, off, nt = ctx.encode_reads(reads, refs, anc, g.p("a"), min_alt, max_rec, 1.0, bounds)
    dc = ctx.dna_coder(c, g.p("level"), g.p("n_pseudo"))
    out, sizes = dc.encode(refs, es, off, nt, np.asarray(bounds))
    raw = out.cpu().numpy().tobytes()
    got, o = [], 0
    for i, s in enumerate(sizes):
        got.append([int(bounds[i + 1] - bounds[i]), int(s), hashlib.sha256(raw[o:o + s]).hexdigest()])
        o += s
    assert got == g.spec["streams"]["dna"]["parts"]
    # the quality stream from the same tuple streams: at levels 2 and 3 its contexts use the per-base classes of the script
    if rs.quals is not None and len(rs.quals) and g.p("qual_mode") != O.QM.get("none", 8):
        from oracle import pyoracle as O2
        qoff = torch.from_numpy(rs.offsets).to(ctx.device)
        flags = ctx.es_flags(reads, es, off, qoff) if g.p("level") > 1 else None
        if flags is not None:
            exp_fl = np.concatenate([O2.es_flags(g.es[i][2], len(rs.read(i))) for i in range(rs.n_reads)])
            assert np.array_equal(flags.cpu().numpy(), exp_fl)
        d = O2.QUAL_DEFAULTS[g.p("qual_mode")]
        qc = ctx.qual_coder(g.p("qual_mode"), g.p("source"), g.p("level"), d[0], d[1])
        qout, qsizes = qc.encode(reads, torch.from_numpy(rs.quals).to(ctx.device), qoff, np.asarray(bounds), flags)
        qraw = qout.cpu().numpy().tobytes()
        qgot, o = [], 0
        for s_ in qsizes:
            qgot.append([0, int(s_), hashlib.sha256(qraw[o:o + s_]).hexdigest()])
            o += s_
        assert qgot == g.spec["streams"]["qual"]["parts"]
        qc.free()
    dc.free(); anc.free(); refs.free(); index.free(); lists.free(); kset.free(); reads.free()


def test_select_reads_and_arena_concat(ctx):
    g = golden("s3m_ont_n_ratio")
    rs = g.reads
    reads = ctx.pack_readset(rs)
    keep = (np.arange(rs.n_reads) % 3 != 1).astype(np.uint8)
    sub = ctx.select_reads(reads, torch.from_numpy(keep).to(ctx.device))
    exp = ctx.pack_readset(ref_subset(rs, keep.astype(bool)))
    for a, b in ((sub, exp),):
        assert a.n_reads == b.n_reads and a.total_words == b.total_words and a.total_bases == b.total_bases
        assert torch.equal(a.packed(), b.packed()) and torch.equal(a.invalid(), b.invalid())
        assert torch.equal(a.word_offsets(), b.word_offsets()) and torch.equal(a.lengths(), b.lengths()) and torch.equal(a.has_n(), b.has_n())
    # concatenation of two arenas (what the ranks do with their reference reads)
    both = ctx.reads_from_arena(torch.cat([sub.packed()[:sub.total_words], reads.packed()[:reads.total_words]]),
                                torch.cat([sub.invalid()[:sub.total_words], reads.invalid()[:reads.total_words]]),
                                torch.cat([sub.lengths(), reads.lengths()]))
    assert both.n_reads == sub.n_reads + reads.n_reads and both.total_bases == sub.total_bases + reads.total_bases
    assert torch.equal(both.has_n(), torch.cat([sub.has_n(), reads.has_n()]))
    assert torch.equal(both.word_offsets()[sub.n_reads:], reads.word_offsets() + sub.total_words)
    both.free(); exp.free(); sub.free(); reads.free()


def sparse_g(g):
    """The preset's sparse range in genome lengths (-g; arg_parse.cpp presets use 1, 2, 4, ...): the value that yields the
    range the reference stored (compression.cpp:501-503)."""
    for v in (1.0, 2.0, 3.0, 4.0, 0.5, 1.5, 6.0, 8.0):
        if max(1, int(v * g.p("n_unique") * g.p("f") / g.p("mean_read_len"))) == g.p("sparse_range"):
            return v
    return 1.0


@pytest.mark.parametrize("cfg", ["c1_ont_default", "c2_hifi_org", "c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "s5m_hifi", "c7_hifi_balanced", "s6m_ont_k25", "s4m_ont_k23_balanced"])
def test_compress_shard_one_call_equals_reference(ctx, cfg):
    """cl_compress_shard — the C++ wiring of all stages (runCompression's data path) — from read bases and qualities to
    the parts of both streams in one native call: sizes and SHA-256 of the unmodified reference's parts."""
    import hashlib
    g = golden(cfg)
    rs = g.reads
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    prm = dict(k=g.p("k"), f=g.p("f"), ci=g.p("ci"), cs=g.p("cs"), c=g.p("c"), anchor_len=g.p("a"), min_part_alt=min_alt, max_rec=max_rec, min_anchors=1,
               level=g.p("level"), source=g.p("source"), sparse=g.p("sparse"), sparse_g=sparse_g(g),
               sparse_exponent=float(g.p("sparse_exp")), cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)
    reads = ctx.pack_readset(rs)
    bounds = rs.pack_bounds()
    dc = ctx.dna_coder(g.p("c"), g.p("level"), 0)
    with_q = rs.quals is not None and len(rs.quals) > 0 and g.p("qual_mode") != 8
    qc = None
    ctx2 = None
    if with_q:
        d = O.QUAL_DEFAULTS[g.p("qual_mode")]
        if g.p("level") == 1:                              # a coder on a second context: the quality stream runs concurrently
            from colord_amd.device import Context
            ctx2 = Context(0)
        qc = (ctx2 or ctx).qual_coder(g.p("qual_mode"), g.p("source"), g.p("level"), d[0], d[1])
    dna, dsz, qual, qsz, info = ctx.compress_shard(reads, prm, bounds, bounds, dc, qc, torch.from_numpy(rs.quals).to(ctx.device) if with_q else None,
                                                   torch.from_numpy(rs.offsets).to(ctx.device) if with_q else None)
    if g.p("sparse"):
        assert info["sparse_range"] == g.p("sparse_range")
    assert info["n_refs"] == int((g.accept.astype(bool) & ~rs.has_n()).sum())

    def parts(raw, sizes, meta):
        raw, o, out = raw.cpu().numpy().tobytes(), 0, []
        for i, s in enumerate(sizes):
            out.append([meta(i), int(s), hashlib.sha256(raw[o:o + int(s)]).hexdigest()])
            o += int(s)
        return out
    assert parts(dna, dsz, lambda i: int(bounds[i + 1] - bounds[i])) == g.spec["streams"]["dna"]["parts"]
    if with_q:
        assert parts(qual, qsz, lambda i: 0) == g.spec["streams"]["qual"]["parts"]
        qc.free()
    if ctx2 is not None:
        ctx2.close()
    dc.free(); reads.free()


@pytest.mark.parametrize("side,has_symbol", [("left", True), ("left", False), ("right", True), ("right", False)])
def test_one_reference_symbol_flank_closed_form(ctx, side, has_symbol):
    """A flank of the read against ONE reference symbol (GK_FLANK_TINY, use == 1) takes a closed form in the wave aligner instead
    of a 1 x m sweep + Hirschberg (measured: 1 x 121 957 was the slowest gap of its launch, 87 ms).  The reference runs
    find_edit_dist there (edit_script.h:156-239, 272-279, 346-354): the oracle's full DP is the judge, for a symbol that occurs in
    the flank (matched at its first occurrence in alignment order) and for one that does not (substitution), on both flanks."""
    from colord_amd.fastq import ReadSet
    rng = np.random.default_rng(11)
    a, k, f, min_alt, max_rec = 16, 20, 12, 64, 3
    core = rng.integers(0, 4, 6000, dtype=np.uint8)
    x = np.uint8(2)
    flank = rng.integers(0, 4, 40000, dtype=np.uint8)                     # >= 32 752 columns: the wave class, not the four-per-wave one
    if not has_symbol:
        flank[flank == x] = (x + 1) % 4
    if side == "left":
        ref, enc = np.concatenate([[x], core]), np.concatenate([flank, core])
    else:
        ref, enc = np.concatenate([core, [x]]), np.concatenate([core, flank])
    seqs = [ref.astype(np.uint8), enc.astype(np.uint8)]
    lens = np.array([len(s) for s in seqs], np.int64)
    rs = ReadSet(np.concatenate(seqs), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), None, [], [False, False], False)
    reads = ctx.pack_readset(rs)
    accept = torch.tensor([1, 0], dtype=torch.uint8, device=ctx.device)
    refs = ctx.select_reads(reads, accept)
    c = 5
    crefs = torch.zeros((2, c), dtype=torch.int32, device=ctx.device)
    cnt = torch.tensor([0, 1], dtype=torch.int32, device=ctx.device)
    anc = ctx.anchor_candidates(reads, refs, crefs, cnt, a)
    assert int(anc.n_cands().cpu()[1]) == 1
    es, off, nt = ctx.encode_reads(reads, refs, anc, a, min_alt, max_rec, 1.0, np.array([0, 2], np.uint32))
    h_es, h_off = es.cpu().numpy(), off.cpu().numpy()
    orc = O.Encoder(a, k, f, 0, min_part_alt=min_alt, max_rec=max_rec)
    orc.add_ref(seqs[0])
    orc.new_pack()
    for i, cands in ((0, []), (1, [0])):
        t, n_t = orc.encode(seqs[i], False, cands, None)
        assert h_es[h_off[i]:h_off[i + 1]].tobytes() == t, f"tuple stream of read {i} differs"
        assert int(nt.cpu()[i]) == n_t
    anc.free(); refs.free(); reads.free()
