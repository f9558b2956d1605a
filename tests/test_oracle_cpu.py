"""CPU suite: the oracle's C restatement against golden vectors dumped from the unmodified reference
(stages a1-a6).  No GPU needed."""
import numpy as np
import pytest
from oracle import pyoracle as O
from util import PLAIN_CONFIGS, golden


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_count_filter_matches_reference_kmc(cfg):
    g = golden(cfg)
    km = O.kmer_scan_reads(g.reads, g.p("k"), g.p("f"))
    keys, cnt, st = O.count_filter(km, g.p("ci"), g.p("cs"))
    assert st.tot_kmers == g.p("tot_kmers")                 # "#Total no. of k-mers"
    assert st.n_unique_counted == g.p("n_unique")           # "#Unique_counted_k-mers"
    assert st.total_count_filtered == g.p("total_count_filtered")
    gk, gc = g.kept
    assert np.array_equal(keys, gk) and np.array_equal(cnt, gc)
    assert g.reads.n_reads == g.p("n_reads")


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_acceptor_matches_reference_rng(cfg):
    g = golden(cfg)
    acc = O.ref_accept(g.p("n_reads"), g.p("n_pseudo"), g.p("sparse_range"), g.p("sparse_exp"))
    if not g.p("sparse"):
        acc[:] = 1
    assert np.array_equal(acc, g.accept)
    assert int(acc.sum()) == g.p("tot_ref_reads") or not g.p("sparse")


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_candidates_match_reference_graph(cfg):
    g = golden(cfg)
    k, f = g.p("k"), g.p("f")
    keys, _ = g.kept
    graph = O.Graph(g.p("c"), g.p("cs"))
    has_n = g.reads.has_n()
    hifi = g.p("source") == 2
    for i in range(g.reads.n_reads):
        ak = O.accepted_kmers(g.reads.read(i), k, f, keys)
        refs, votes, com = graph.next_read(ak, bool(g.accept[i]) and not has_n[i], hifi)
        ref = g.cands[i]
        assert ref["read_id"] == i and ref["has_n"] == bool(has_n[i]) and ref["len"] == len(g.reads.read(i))
        assert list(refs) == ref["refs"], f"read {i}"
        if hifi:
            for a, b in zip(com, ref["common"]):
                assert np.array_equal(a, b)


def test_hash_and_refread_layout():
    # known-answer: fmix64 of small integers (MurmurHash3 finaliser) and the 2-bit store layout
    assert O.lib().orc_hash_mm(0) == 0
    assert O.lib().orc_hash_mm(1) == 0xb456bcfc34c2cb2c
    b = np.array([0, 1, 2, 3, 3, 2], np.uint8)
    out = np.zeros(3, np.uint8)
    assert O.lib().orc_refread_compact(b, len(b), out) == 3
    assert list(out) == [0b00011011, 0b11100000, 2]
