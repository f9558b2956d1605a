#!/usr/bin/env python3
"""The anchor stage alone (reads of the bench recipe -> k-mer set -> candidates -> cl_anchor_candidates) with the m-mer tables in HBM
(COLORD_HIP_ANCHORS_LDS=0), in LDS for reads up to 12 288 m-mers, and for reads up to 24 576: kernel times by HIP events, wall time of the
stage, and that the three give the same anchors."""
import os, sys, subprocess, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from colord_amd.device import Context, _check
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
    anc = ctx.anchor_candidates(reads, ref_arena, crefs, cnt, a); anc.free()
    ctx.acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        anc = ctx.anchor_candidates(reads, ref_arena, crefs, cnt, a)
    torch.cuda.synchronize(); _check(ctx, 0); wall = (time.perf_counter() - t0) / 2
    h = hashlib.sha256()
    for x in (anc.n_cands(), anc.cands(), anc.cand_offsets(), anc.data()):
        h.update(x.cpu().numpy().tobytes())
    print(f"   stage {wall * 1e3:.0f} ms for {bases / 1e9:.2f} Gbases; anchors {anc.data().numel() // 3}; sha {h.hexdigest()[:16]}")
    for n in sorted(ctx.acc, key=lambda n: -ctx.acc[n][0])[:9]:
        print(f"   {n}: {ctx.acc[n][0] / 2:.1f} ms per call of the stage ({ctx.acc[n][1] // 2} launches)")
else:
    bases = sys.argv[1] if len(sys.argv) > 1 else "1e9"
    for v in (sys.argv[2:] or ["0", "12288", "24576"]):
        print("COLORD_HIP_ANCHORS_LDS =", v, flush=True)                   # (or NAME=value: any other switch, e.g. COLORD_HIP_MATCH_SEG=0)
        extra = dict([v.split("=", 1)]) if "=" in v else {"COLORD_HIP_ANCHORS_LDS": v}
        subprocess.run([sys.executable, __file__, "child", bases], env=dict(os.environ, **extra))
