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
