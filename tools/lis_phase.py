#!/usr/bin/env python3
"""Phase timing of k_lis_anchors (COLORD_HIP_LIS_DBG: 1 = stop after the LIS forward pass, 2 = after the predecessor walk) on the anchor
stage alone: reads of the bench recipe -> k-mer set -> candidates -> cl_anchor_candidates, kernel times by HIP events."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from colord_amd.device import Context
    from colord_amd import ontsim
    bases = int(float(sys.argv[2]))
    t = ontsim.ReadTable(seed=5, genome_len=max(1_000_000, bases // 17), target_bases=bases)
    ctx = Context(0, timing=True)
    codes, off, _ = ontsim.device_reads(t, ctx.device, 0, t.n_reads, with_quals=False)
    reads = ctx.pack_reads(codes, off)
    k, f, ci, cs, c, a = 25, 12, 4, 80, 5, 22
    km = ctx.kmer_scan(reads, k, f)
    kset, st = ctx.count_filter(km, k, ci, cs)
    lists = ctx.accepted_kmers(kset, reads, k, f)
    acc = ctx.ref_accept(t.n_reads, 0, max(1, t.n_reads // 49), 1.0)
    index = ctx.index_build(kset, lists, torch.from_numpy(acc), 0, cs)
    accept = torch.from_numpy(acc.copy()).to(ctx.device) & (reads.has_n() == 0).to(torch.uint8)
    ref_arena = ctx.select_reads(reads, accept)
    crefs, _, cnt = ctx.candidates(index, lists, c)
    ctx.acc.clear()
    for _ in range(2):
        anc = ctx.anchor_candidates(reads, ref_arena, crefs, cnt, a)
    from colord_amd.device import _check
    torch.cuda.synchronize(); _check(ctx, 0)
    for n in ("k_lis_anchors", "k_match", "k_table_insert", "k_task_pairs"):
        if n in ctx.acc:
            print(f"   {n}: {ctx.acc[n][0] / 2:.1f} ms per call of the stage ({ctx.acc[n][1] // 2} launches)")
else:
    bases = sys.argv[1] if len(sys.argv) > 1 else "1e9"
    for dbg in ("0", "1", "2"):
        print("COLORD_HIP_LIS_DBG =", dbg, flush=True)
        subprocess.run([sys.executable, __file__, "child", bases], env=dict(os.environ, COLORD_HIP_LIS_DBG=dbg))
