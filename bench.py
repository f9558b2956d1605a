#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native CoLoRd compress data path on synthetic ONT reads.

One "step" = one pass of the whole hot path over one per-GPU shard of synthetic ONT reads that is already
resident in HBM as a packed read arena (+ raw quality bytes):
    a1 canonical k-mer scan + murmur-modulo filter -> a2 exact count/threshold -> a3 membership table
    -> a4 accepted k-mers per read -> a6 acceptor -> a5 index + candidates -> a7 reference-read arena
    -> a8 m-mer anchors -> a10/a11 gap alignment, cost decisions, recursion -> a12 tuple streams
    -> a14/a16 DNA stream range coder;   a13/a15 quality stream range coder (4-avg, level 1)
Not in the step (rows "f / next" of DESIGN.md): FASTQ parsing, the header (ID) stream, the archive container.

Contract: python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run, one rank/GPU).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bases", type=float, default=2.0e9, help="synthetic bases per GPU (weak scaling)")
    ap.add_argument("--coverage", type=float, default=16.7, help="genome = total bases / coverage (50 Gbases over 3 Gb)")
    ap.add_argument("--k", type=int, default=25)           # compression.cpp:84-88 for a 50 Gbase input
    ap.add_argument("--cpu-sample-bases", type=float, default=1.5e8)
    ap.add_argument("--pack-symbols", type=int, default=1 << 16,
                    help="part (= range-coder restart) size in symbols.  4194304 reproduces the reference's packs (defs.h:45) and its exact "
                         "bytes; smaller parts are equally valid archives (the reference decoder follows the part table), cost 8 flush "
                         "bytes each (+0.04 %% at 64 Ki) and expose the parallelism the per-part dependent chain needs")
    ap.add_argument("--no-qual", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


# ONT default ("memory") preset: arg_parse.cpp:89-408 / SURVEY App. B
PRESET = dict(f=12, ci=4, cs=80, c=5, g=1.0, exponent=1.0, a=22, min_part_alt=64, max_rec=3)   # a: compression.cpp:84-88 with k=25


QUAL_CTX = None       # second context of the same GPU for the quality stream (single-GPU runs)


class StepTimes:
    """Per-kernel HIP-event times (context streams) accumulated by colord_amd.device.Context."""
    def __init__(self, *ctxs):
        self.ms, self.launches, self.bytes = {}, {}, {}
        for ctx in ctxs:
            if ctx is None:
                continue
            for n, v in ctx.acc.items():
                self.ms[n] = self.ms.get(n, 0.0) + v[0]
                self.launches[n] = self.launches.get(n, 0) + v[1]
                self.bytes[n] = self.bytes.get(n, 0.0) + v[2]


def reference_part_bounds(lengths: np.ndarray, pack_symbols: int) -> np.ndarray:
    """Greedy part cut of the reference reader: a pack closes once sum(len+1) >= pack_symbols (in_reads.cpp:62-77)."""
    acc = np.cumsum(lengths.astype(np.int64) + 1)
    bounds = [0]
    base = 0
    while True:
        i = int(np.searchsorted(acc, base + pack_symbols, side="left"))
        if i >= len(acc):
            break
        bounds.append(i + 1)
        base = int(acc[i])
    if bounds[-1] != len(lengths):
        bounds.append(len(lengths))
    return np.asarray(bounds, dtype=np.uint32)


def hot_path_step(ctx, reads, k, quals=None, qual_off=None, part_bounds=None, est_bounds=None):
    """One pass of the stages built so far (single- or multi-GPU).  Returns sizes for reporting."""
    from colord_amd import parallel as par
    p = PRESET
    w, rank = par.world(), par.rank()
    if w == 1 and quals is not None and not os.environ.get("BENCH_STAGE_TIMES") and not os.environ.get("BENCH_PY_STAGES"):   # BENCH_PY_STAGES: the stage-by-stage path of the multi-GPU runs, on one GPU
        # single GPU: the whole path is one native call (cl_compress_shard, the C++ wiring of the stages)
        prm = dict(k=k, f=p["f"], ci=p["ci"], cs=p["cs"], c=p["c"], anchor_len=p["a"], min_part_alt=p["min_part_alt"], max_rec=p["max_rec"], min_anchors=1,
                   level=1, source=0, sparse=1, sparse_g=p["g"], sparse_exponent=p["exponent"], cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)
        qc = (QUAL_CTX or ctx).qual_coder(2, 0, 1, (7, 14, 26), ())   # on a second context: coded concurrently with the DNA path
        dc = ctx.dna_coder(p["c"], 1, 0)
        dna, dsz, qual, qsz, inf = ctx.compress_shard(reads, prm, part_bounds, est_bounds, dc, qc, quals, qual_off)
        qc.free(); dc.free()
        return dict(tot_kmers=inf["tot_kmers"], kept=inf["n_kept_kmers"], refs=inf["n_refs"], anchors=inf["n_anchors"], tuple_bytes=inf["tuple_bytes"],
                    dna_bytes=inf["dna_bytes"], qual_bytes=inf["qual_bytes"], qual_parts=len(qsz), sparse_range=inf["sparse_range"])
    stage_t = {} if os.environ.get("BENCH_STAGE_TIMES") else None
    t_last = [time.perf_counter()]

    def lap(name):                                         # wall time per stage (diagnostic; adds syncs)
        if stage_t is not None:
            torch.cuda.synchronize()
            now = time.perf_counter()
            stage_t[name] = stage_t.get(name, 0.0) + (now - t_last[0]) * 1e3
            t_last[0] = now
    qjob = None
    if quals is not None and QUAL_CTX is not None and stage_t is None:
        # the quality stream of level 1 needs nothing of the DNA path: a second context of the same GPU codes it on a
        # host thread meanwhile (the native call releases the GIL), as cl_compress_shard does on one GPU
        import threading
        qres = {}

        def qrun():
            try:
                qc_ = QUAL_CTX.qual_coder(2, 0, 1, (7, 14, 26), ())
                qres["out"] = qc_.encode(reads, quals, qual_off, part_bounds)
                qc_.free()
            except Exception as e:          # surfaced after the join
                qres["err"] = e
        qjob = threading.Thread(target=qrun)
        qjob.start()
    km = ctx.kmer_scan(reads, k, p["f"])
    n_surv = km.numel()
    km = par.exchange_kmers(km)                            # exchange 1a: k-mers to the owner of their key
    kset, st = ctx.count_filter(km, k, p["ci"], p["cs"])
    tot_kmers, n_unique, n_reads_total = par.all_reduce_sum_ints(st.tot_kmers, st.n_unique_counted, reads.n_reads)
    if w > 1:                                              # exchange 1b: replicate the filtered set
        allk = torch.cat(par.all_gather_v(kset.keys()))
        allc = torch.cat(par.all_gather_v(kset.counts()))
        kset.free()
        kset = ctx.kmer_set_from_keys(allk, allc, k)
    lists = ctx.accepted_kmers(kset, reads, k, p["f"])
    # host scalars exactly as compression.cpp:443,501 derives them
    mean_read_len = int(float(tot_kmers * p["f"]) / n_reads_total + k - 1)
    sparse_range = max(1, int((p["g"] * n_unique * p["f"]) / mean_read_len))
    first_read, _ = par.exclusive_prefix(reads.n_reads, ctx.device)
    acc_all = ctx.ref_accept(n_reads_total, 0, sparse_range, p["exponent"])    # same stream on every rank
    acc = acc_all[first_read:first_read + reads.n_reads]
    accept = torch.from_numpy(acc.copy()).to(ctx.device) & (reads.has_n() == 0).to(torch.uint8)
    ref_base, n_refs_total = par.exclusive_prefix(int(accept.sum().item()), ctx.device)
    ids, refs, bounds, _ = ctx.index_entries(lists, accept, ref_base)
    if w > 1:                                              # exchange 2: replicate the k-mer -> reference reads index
        ids = torch.cat(par.all_gather_v(ids))
        refs = torch.cat(par.all_gather_v(refs))
    index = ctx.index_build_pairs(kset, ids, refs, bounds, n_refs_total, 0, p["cs"])
    crefs, votes, cnt = ctx.candidates(index, lists, p["c"])
    lap("a1-a7 k-mers, index, candidates")
    out = dict(survivors=n_surv, tot_kmers=tot_kmers, kept=kset.size, accepted=lists.total, refs=n_refs_total,
               index_entries=index.entries, with_candidates=int((cnt > 0).sum().item()))
    if quals is not None:
        # a13+a15: quality stream, ONT default 4-avg at level 1 (contexts do not need the edit script at level 1).
        # One model domain per GPU (the adaptive models live for the whole shard), parts cut like the reference's packs.
        if qjob is None:
            qc = ctx.qual_coder(2, 0, 1, (7, 14, 26), ())
            payload, sizes = qc.encode(reads, quals, qual_off, part_bounds)
            qc.free()
            lap("a13+a15 quality stream")
            out.update(qual_bytes=int(payload.numel()), qual_parts=len(sizes))
        # a8 + a10-a12 + a14: DNA stream.  Reference reads = the accepted reads of all ranks (CReferenceReads is one
        # process-wide store in the reference; each rank replicates it), anchors against the candidates, edit scripts,
        # tuple streams, DNA coder at level 1.  The estimator packs are the reference's reader packs (4 Mi symbols,
        # defs.h:45); the coder parts are the same as the quality stream's (the decoder requires that, entr_qual.h:150-170).
        my_refs = ctx.select_reads(reads, accept)
        if w > 1:                                          # exchange 3: replicate the reference reads
            pk = torch.cat(par.all_gather_v(my_refs.packed()[:my_refs.total_words]))
            iv = torch.cat(par.all_gather_v(my_refs.invalid()[:my_refs.total_words]))
            ln = torch.cat(par.all_gather_v(my_refs.lengths()))
            my_refs.free()
            my_refs = ctx.reads_from_arena(pk, iv, ln)
        lap("reference reads")
        anc = ctx.anchor_candidates(reads, my_refs, crefs, cnt, p["a"])
        lap("a8 anchors")
        es, es_off, es_nt = ctx.encode_reads(reads, my_refs, anc, p["a"], p["min_part_alt"], p["max_rec"], 1.0, est_bounds)
        lap("a10-a12 encoder")
        n_plain = int((es[es_off[:-1]] >> 4 != 10).sum().item())
        dc = ctx.dna_coder(p["c"], 1, 0)
        dpayload, dsizes = dc.encode(my_refs, es, es_off, es_nt, part_bounds)
        dc.free()
        lap("a14 DNA stream")
        out.update(dna_bytes=int(dpayload.numel()), tuple_bytes=int(es.numel()), reads_stored_plain=n_plain, anchors=int(anc.total))
        anc.free(); my_refs.free()
        if qjob is not None:
            qjob.join()
            if "err" in qres:
                raise qres["err"]
            payload, sizes = qres["out"]
            out.update(qual_bytes=int(payload.numel()), qual_parts=len(sizes))
    index.free(); lists.free(); kset.free()
    if stage_t is not None:
        print("stage wall ms:", {k_: round(v, 1) for k_, v in stage_t.items()}, file=sys.stderr)
    return out


def cpu_baseline(sample_bases: float, k_hint: int):
    """The UNMODIFIED reference binary (oracle/_ref/colord, built by oracle/Makefile.ref) timed on this
    host's cores on a bounded sample of the same synthetic recipe.  It runs the WHOLE compressor."""
    from colord_amd.synth import make_reads
    from colord_amd.fastq import write_fastq
    ref = os.path.join(ROOT, "oracle", "_ref", "colord")
    if not os.path.exists(ref):
        return None
    cores = os.cpu_count() or 1
    rs = make_reads(seed=101, genome_len=int(sample_bases / 16.7), target_bases=int(sample_bases))
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "sample.fastq")
        write_fastq(fq, rs)
        t0 = time.time()
        subprocess.check_call([ref, "compress-ont", "-t", str(cores), fq, os.path.join(tmp, "o.colord")],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        size = os.path.getsize(os.path.join(tmp, "o.colord"))
    return {"value": len(rs.bases) / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "reference",
            "sample": f"oracle/_ref/colord compress-ont -t {cores} on {len(rs.bases)} synthetic ONT bases ({rs.n_reads} reads), "
                      f"whole compressor (parsing, header stream and archive included); {dt:.2f} s wall, archive {size} B = {size / len(rs.bases):.4f} B/base"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: launch with torch.distributed.run for --gpus > 1", file=sys.stderr)
            sys.exit(2)
    backend = os.environ.get("BENCH_BACKEND", "nccl")      # "gloo": functional check of the multi-rank path with ranks sharing GPUs
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from colord_amd.device import Context
    from colord_amd.synth_device import make_reads_device

    ctx = Context(local, timing=True)
    global QUAL_CTX
    QUAL_CTX = Context(local, timing=True) if not os.environ.get("BENCH_NO_OVERLAP") else None
    bases = int(args.bases)
    genome_len = max(1_000_000, int(bases * world / args.coverage))
    # same genome on every rank (seed), different reads per rank
    codes, offsets, quals = make_reads_device(ctx.device, seed=1234, genome_len=genome_len, target_bases=bases,
                                              with_quals=True, read_seed=1000 + rank)
    n_reads_local = offsets.numel() - 1
    reads = ctx.pack_reads(codes, offsets)
    local_bases = reads.total_bases
    del codes
    qual_off = offsets.contiguous()
    part_bounds = reference_part_bounds(reads.lengths().cpu().numpy().view(np.uint32), args.pack_symbols)
    est_bounds = reference_part_bounds(reads.lengths().cpu().numpy().view(np.uint32), 1 << 22)      # reader packs (defs.h:45)
    qargs = {} if args.no_qual else dict(quals=quals, qual_off=qual_off, part_bounds=part_bounds, est_bounds=est_bounds)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        hot_path_step(ctx, reads, args.k, **qargs)
    ctx.acc.clear()
    if QUAL_CTX is not None:
        QUAL_CTX.acc.clear()
    sync()
    t0 = time.perf_counter()
    info = None
    for _ in range(args.steps):
        info = hot_path_step(ctx, reads, args.k, **qargs)
    sync()
    dt = time.perf_counter() - t0
    red_dev = ctx.device if backend == "nccl" else torch.device("cpu")
    tdev = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tb = torch.tensor([local_bases], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(tdev, op=dist.ReduceOp.MAX)
        dist.all_reduce(tb)
    dt = float(tdev.item())
    total_bases = int(tb.item())

    times = StepTimes(ctx, QUAL_CTX)
    if rank == 0:
        # dominant kernel by measured HIP-event time on the context stream; `achieved` = the library's algorithmic HBM
        # byte count of those launches (per-kernel formulas in DESIGN.md) / their measured duration
        dom = max(times.ms, key=times.ms.get)
        launches = times.launches[dom]
        avg_ms = times.ms[dom] / launches
        roof = {"bound": "hbm", "kernel": dom, "avg_ms": avg_ms, "launches_per_step": launches / args.steps,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        if times.bytes.get(dom, 0) > 0:
            ach = times.bytes[dom] / (times.ms[dom] * 1e-3) / 1e9
            roof.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "alg_bytes_per_launch": times.bytes[dom] / launches})
        else:
            roof.update({"achieved": None, "frac": None})
        roof["kernel_ms_per_step"] = {n: times.ms[n] / args.steps for n in sorted(times.ms, key=times.ms.get, reverse=True)}
        roof["kernel_achieved_GBps"] = {n: times.bytes[n] / (times.ms[n] * 1e-3) / 1e9 for n in sorted(times.ms, key=times.ms.get, reverse=True)
                                        if times.bytes.get(n, 0) > 0 and times.ms[n] > 0}
        cb = None if args.no_cpu_baseline else cpu_baseline(args.cpu_sample_bases, args.k)
        line = {
            "metric": "input Gbases/s + archive size vs ref, ONT 50 Gb at 1/2/4/8 MI355X", "value": total_bases * args.steps / dt / 1e9,
            "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic ONT, {local_bases} bases/GPU ({n_reads_local} reads, N50~20kb), genome {genome_len} bp, "
                                   f"k={args.k} f={PRESET['f']} ci={PRESET['ci']} cs={PRESET['cs']} c={PRESET['c']} (ONT default preset)",
                       "stages": "a1 k-mer scan, a2 count/filter, a3 set build, a4 accepted k-mers, a6 acceptor, a5 index+candidates"
                                 + ("" if args.no_qual else f", a13+a15 quality stream (4-avg, level 1, parts of {args.pack_symbols} symbols)")
                                 + ("" if args.no_qual else ", a8 m-mer anchors, a10-a12 gap alignment + cost decisions + tuple streams, a14+a16 DNA stream coder"),
                       "stream_bytes_per_base": (None if args.no_qual or not info else round((info.get("dna_bytes", 0) + info.get("qual_bytes", 0)) / max(local_bases, 1), 4)),
                       "parallelism": f"reads sharded x{world}, k-mer set replicated" if world > 1 else "single GPU",
                       "sizes": info},
            "roofline": roof, "cpu_baseline": cb,
        }
        print(json.dumps(line))
    reads.free()
    ctx.close()
    if QUAL_CTX is not None:
        QUAL_CTX.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
