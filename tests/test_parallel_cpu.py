"""CPU suite, world_size 2 over gloo: the two exchange steps of the multi-GPU path (colord_amd/parallel.py)
produce exactly what one process produces on the unsharded input.  The per-rank compute that the HIP
kernels do on a GPU is done here by the oracle, so only the sharding/exchange logic is under test."""
import os
import socket
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from colord_amd import parallel as par
    from oracle import pyoracle as O
    from util import golden
    g = golden(cfg)
    rs = g.reads
    k, f, ci, cs = g.p("k"), g.p("f"), g.p("ci"), g.p("cs")
    cut = [0, rs.n_reads // 3, rs.n_reads]                 # ragged shards on purpose
    lo, hi = cut[rank], cut[rank + 1]
    # stage a1 on the local shard, exchange 1a, stage a2 on the owned key partition
    km = np.concatenate([O.kmer_scan(rs.read(i), k, f) for i in range(lo, hi)] + [np.empty(0, np.uint64)])
    mine = par.exchange_kmers(torch.from_numpy(km.view(np.int64)))
    keys, cnt, st = O.count_filter(mine.numpy().view(np.uint64), ci, cs)
    tot, uniq, nreads = par.all_reduce_sum_ints(st.tot_kmers, st.n_unique_counted, hi - lo)
    allk = torch.cat(par.all_gather_v(torch.from_numpy(keys.view(np.int64)))).numpy().view(np.uint64)
    allc = torch.cat(par.all_gather_v(torch.from_numpy(cnt.view(np.int32)))).numpy().view(np.uint32)
    o = np.argsort(allk, kind="stable")
    ok1 = (tot == g.p("tot_kmers") and uniq == g.p("n_unique") and nreads == g.p("n_reads")
           and np.array_equal(allk[o], g.kept[0]) and np.array_equal(allc[o], g.kept[1]))
    # stage a4 + acceptor on the local shard, exchange 2
    first, total = par.exclusive_prefix(hi - lo, torch.device("cpu"))
    has_n = rs.has_n()
    accept = (g.accept.astype(bool) & ~has_n)
    ref_base, n_refs = par.exclusive_prefix(int(accept[lo:hi].sum()), torch.device("cpu"))
    ids, refs = [], []
    r = ref_base
    for i in range(lo, hi):
        ak = O.accepted_kmers(rs.read(i), k, f, g.kept[0])
        if accept[i]:
            ids.append(np.searchsorted(g.kept[0], ak).astype(np.int32))
            refs.append(np.full(len(ak), r, np.int32))
            r += 1
    ids = torch.from_numpy(np.concatenate(ids + [np.empty(0, np.int32)]))
    refs = torch.from_numpy(np.concatenate(refs + [np.empty(0, np.int32)]))
    gi = torch.cat(par.all_gather_v(ids)).numpy()
    gr = torch.cat(par.all_gather_v(refs)).numpy()
    # single-process expectation
    ei, er, rr = [], [], 0
    for i in range(rs.n_reads):
        if accept[i]:
            ak = O.accepted_kmers(rs.read(i), k, f, g.kept[0])
            ei.append(np.searchsorted(g.kept[0], ak).astype(np.int32))
            er.append(np.full(len(ak), rr, np.int32))
            rr += 1
    ok2 = (first == lo and total == rs.n_reads and n_refs == rr
           and np.array_equal(gi, np.concatenate(ei + [np.empty(0, np.int32)]))
           and np.array_equal(gr, np.concatenate(er + [np.empty(0, np.int32)])))
    ret[rank] = (bool(ok1), bool(ok2))
    dist.destroy_process_group()


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s3m_ont_n_ratio"])
def test_two_rank_exchanges_equal_single_process(cfg):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), cfg, ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True, True), 1: (True, True)}
