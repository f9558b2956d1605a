#!/usr/bin/env python3
"""Diagnostic: size distribution of the level-0 gaps the encoder has to align on the bench's synthetic reads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from colord_amd.device import Context
from colord_amd.synth_device import make_reads_device
import bench

bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
ctx = Context(0)
codes, offsets, quals = make_reads_device(ctx.device, seed=1234, genome_len=max(1_000_000, int(bases / 16.7)), target_bases=bases, with_quals=True, read_seed=1000)
reads = ctx.pack_reads(codes, offsets)
p = bench.PRESET; k = 25
kset, st = ctx.count_filter(ctx.kmer_scan(reads, k, p["f"]), k, p["ci"], p["cs"])
lists = ctx.accepted_kmers(kset, reads, k, p["f"])
mean_read_len = int(float(st.tot_kmers * p["f"]) / reads.n_reads + k - 1)
sparse_range = max(1, int((p["g"] * st.n_unique_counted * p["f"]) / mean_read_len))
acc = ctx.ref_accept(reads.n_reads, 0, sparse_range, p["exponent"])
accept = torch.from_numpy(acc.copy()).to(ctx.device) & (reads.has_n() == 0).to(torch.uint8)
index = ctx.index_build(kset, lists, accept, 0, p["cs"])
crefs, votes, cnt = ctx.candidates(index, lists, p["c"])
refs = ctx.select_reads(reads, accept)
anc = ctx.anchor_candidates(reads, refs, crefs, cnt, p["a"])
n_c = anc.n_cands().cpu().numpy(); tab = anc.cands().cpu().numpy().view(np.uint32); off = anc.cand_offsets().cpu().numpy(); data = anc.data().cpu().numpy().view(np.uint32)
rl = reads.lengths().cpu().numpy().view(np.uint32); fl = refs.lengths().cpu().numpy().view(np.uint32)
print("reads", reads.n_reads, "bases", reads.total_bases, "refs", refs.n_reads, "with candidates", int((n_c > 0).sum()), "anchors", anc.total)
rows, cols, kind = [], [], []
c = p["c"]
for r in np.nonzero(n_c > 0)[0]:
    rid, rev, tot, na = tab[r, 0]
    a = data[off[r * c]:off[r * c] + na]
    pe, pr, ln = a[:, 1].astype(np.int64), a[:, 2].astype(np.int64), a[:, 0].astype(np.int64)
    ne = np.concatenate([[pe[0]], pe[1:] - (pe[:-1] + ln[:-1]), [rl[r] - (pe[-1] + ln[-1])]])
    nr = np.concatenate([[pr[0]], pr[1:] - (pr[:-1] + ln[:-1]), [fl[rid] - (pr[-1] + ln[-1])]])
    kd = np.ones(len(ne), np.int64); kd[0] = 0; kd[-1] = 2
    use = np.where(kd == 1, nr, np.minimum(2 * ne, nr))
    rows.append(np.where(kd == 1, use, ne)); cols.append(np.where(kd == 1, ne, use)); kind.append(kd)
rows, cols, kind = np.concatenate(rows), np.concatenate(cols), np.concatenate(kind)
triv = (rows == 0) | (cols == 0)
small = ~triv & (rows <= 256) & (cols <= 256)
large = ~triv & ~small
cells = rows.astype(float) * cols
print("gaps", len(rows), "trivial", int(triv.sum()), "small", int(small.sum()), "large", int(large.sum()))
print("cells: small %.3g large %.3g" % (cells[small].sum(), cells[large].sum()))
for nm, sel in (("inner large", large & (kind == 1)), ("flank large", large & (kind != 1))):
    if sel.any():
        print(nm, int(sel.sum()), "rows pct", np.percentile(rows[sel], [50, 90, 99, 100]).astype(int), "cols pct", np.percentile(cols[sel], [50, 90, 99, 100]).astype(int), "cells %.3g" % cells[sel].sum(),
              "hirschberg", int((sel & (20 * ((rows + 63) // 64) * cols + 8 * cols >= (1 << 20))).sum()))
