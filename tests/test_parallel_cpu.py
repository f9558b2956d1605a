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


def _exchange_worker(rank, world, port, ret):
    """The three cl_exchange callbacks (what csrc/stream.hip calls through the C struct) on host pointers, ragged sizes."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from colord_amd import parallel as par
    ex = par.TorchExchange(None)
    X = ex.c_struct
    ok = X.rank == rank and X.world == world
    U64P = C.POINTER(C.c_uint64)
    # all_gather_host: 3 values per rank, rank order
    mine = np.array([rank + 1, 10 * (rank + 1), 2 ** 40 + rank], np.uint64)
    out = np.zeros(3 * world, np.uint64)
    ok &= X.all_gather_host(None, mine.ctypes.data_as(U64P), 3, out.ctypes.data_as(U64P)) == 0
    ok &= np.array_equal(out, np.concatenate([[r + 1, 10 * (r + 1), 2 ** 40 + r] for r in range(world)]).astype(np.uint64))
    # all_to_all_v: rank r sends (r + 1) * (d + 2) k-mers to rank d (8-byte elements, ragged, one empty pair)
    cnt = lambda s, d: 0 if (s, d) == (1, 0) else (s + 1) * (d + 2)
    send = np.concatenate([np.full(cnt(rank, d), 1000 * rank + d, np.uint64) for d in range(world)] + [np.empty(0, np.uint64)])
    sb = np.array([8 * cnt(rank, d) for d in range(world)], np.uint64)
    rb = np.array([8 * cnt(s, rank) for s in range(world)], np.uint64)
    recv = np.zeros(int(rb.sum()) // 8, np.uint64)
    ok &= X.all_to_all_v(None, send.ctypes.data, sb.ctypes.data_as(U64P), recv.ctypes.data, rb.ctypes.data_as(U64P)) == 0
    ok &= np.array_equal(recv, np.concatenate([np.full(cnt(s, rank), 1000 * s + rank, np.uint64) for s in range(world)] + [np.empty(0, np.uint64)]))
    # all_gather_v: odd byte counts (no padding to the largest shard), one rank contributes nothing
    n_of = lambda r: 0 if r == 1 else 5 + 7 * r
    mine_b = np.full(n_of(rank), 65 + rank, np.uint8)
    rbb = np.array([n_of(r) for r in range(world)], np.uint64)
    got = np.zeros(int(rbb.sum()), np.uint8)
    ok &= X.all_gather_v(None, mine_b.ctypes.data if len(mine_b) else None, len(mine_b), got.ctypes.data, rbb.ctypes.data_as(U64P)) == 0
    ok &= np.array_equal(got, np.concatenate([np.full(n_of(r), 65 + r, np.uint8) for r in range(world)]))
    ok &= ex.err is None and ex.bytes_moved == (int(rb.sum()) - int(rb[rank])) + (int(rbb.sum()) - n_of(rank))
    # gather_to_root: variable-length payloads to rank 0
    parts = par.gather_to_root(torch.full((3 + 4 * rank,), rank, dtype=torch.uint8))
    if rank == 0:
        ok &= [p.tolist() for p in parts] == [[r] * (3 + 4 * r) for r in range(world)]
    else:
        ok &= parts is None
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_callbacks_over_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_exchange_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}
