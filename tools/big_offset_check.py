#!/usr/bin/env python3
"""Offsets beyond 2^31 elements / 2^32 bytes inside ONE chunk of the compressor (round-4 verdict: the v_readlane sign extension at 2.4 GB of
coder output was found in a bench, by luck).  The same reads are compressed twice through the bench's own path: as ONE chunk of --bases
(default 2.2 Gbases: 2.2 G quality symbols and ~2.4 G `dna` symbols in one model stage, one sort, one tuple walk, one emission — element
indices above 2^31, byte offsets above 2^34) and in chunks of a quarter of it.  With the reference's part cut (reader packs, cut over the whole read sequence) chunking changes no byte, so the two whole
streams and their part sizes must be equal; printed with their digests.  Usage: tools/big_offset_check.py [bases]   (exit status 1 on a difference)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch, xxhash
import bench
from colord_amd import ontsim
from colord_amd.device import Context

bases = float(sys.argv[1]) if len(sys.argv) > 1 else 2.2e9
k, a = 25, 22
ctx, qctx = Context(0), Context(0)
table = ontsim.ReadTable(seed=7, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=int(bases))
prm = bench.params_for(k, a)
res = {}
for name, cb in (("one_chunk", bases * 1.01), ("four_chunks", bases / 4)):
    shard = bench.Shard(ctx, table, 0, table.n_reads, cb, 1 << 22, True)    # (the reference's part cut: the reader packs do not depend on the chunking)
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    big = max(int(c[0].total_bases) for c in shard.chunks)
    sink = bench.HostSink(ctx.device, int(shard.n_bases * 0.22) + (1 << 26), int(shard.n_bases * 0.30) + (1 << 26), int(big * 0.30) + (1 << 24), int(big * 0.36) + (1 << 24))
    t0 = time.perf_counter()
    info = bench.hot_path_step(ctx, qctx, shard, prm, True, None, None, None, shard.n_bases, ref_cut=True, sink=sink)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dsz = np.concatenate(sink.dna_sizes); qsz = np.concatenate(sink.qual_sizes)
    hd = xxhash.xxh3_128(); hd.update(dsz.tobytes()); hd.update(memoryview(sink.h_dna.numpy()[:int(dsz.sum())]))
    hq = xxhash.xxh3_128(); hq.update(qsz.tobytes()); hq.update(memoryview(sink.h_qual.numpy()[:int(qsz.sum())]))
    res[name] = {"chunks": len(shard.chunks), "largest_chunk_bases": big, "seconds": round(dt, 2), "dna_bytes": int(dsz.sum()), "qual_bytes": int(qsz.sum()), "parts": int(len(dsz)),
                 "dna_xxh3": hd.hexdigest(), "qual_xxh3": hq.hexdigest()}
    print(name, json.dumps(res[name]), flush=True)
    shard.free(); del sink, shard
    torch.cuda.empty_cache()
same = all(res["one_chunk"][f] == res["four_chunks"][f] for f in ("dna_bytes", "qual_bytes", "parts", "dna_xxh3", "qual_xxh3"))
print("EQUAL" if same else "DIFFERENT", f"({table.n_reads} reads, {bases / 1e9:.2f} Gbases, k={k} a={a}, coder parts = reader packs of 4 Mi symbols)")
sys.exit(0 if same else 1)
