#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native CoLoRd compress data path on the configuration BASELINE.json's metric is
quoted on: a synthetic ONT set of 50 Gbases (N50 ~ 20 kb, 4-avg qualities, 3 Gb genome), on 1, 2, 4 or 8 MI355X.

One "step" = one pass of the WHOLE hot path over the whole input, which is already resident in HBM as 2-bit read arenas
plus raw quality bytes (cut in chunks of ~1 Gbase, the unit the streaming compressor works on, csrc/stream.hip):
    pass 1   a1 canonical k-mer scan + murmur-modulo filter of every chunk -> a2 exact counts by key range -> a3 set
    pass 2a  a4 accepted k-mers, a6 acceptor, a7 reference-read store, a5 k-mer -> reference-reads index over the whole input
    pass 2b  per chunk: a5 candidates, a8 m-mer anchors, a10/a11 gap alignment + cost decisions + recursion, a12 tuples,
             a14/a16 `dna` stream parts, a13/a15 `qual` stream parts (4-avg, level 1) with coders that persist across chunks
The step ends with every compressed part in (pinned) HOST memory — SURVEY §8d's T_core: each chunk's parts are copied out while the
next chunk is coded (double-buffered staging on the device, a copy stream).
With N > 1 the reads are sharded in file order (total work fixed: "scaling": "strong"), the k-mer set, the reference reads and
the index are replicated through the two exchanges of SURVEY.md §8e (RCCL via torch.distributed), and the compressed parts
are gathered to rank 0 inside the step.  Not in the step: FASTQ parsing, the header (ID) stream, the archive container.
After the timed steps the output is checked: the first `dna` parts of the last pass are decoded by the library's host decoder
(cl_dna_decode_part) and compared with the generator's bases (`round_trip_checked`).

Also in the JSON line: `archive_vs_ref` — on a bounded sample of the same recipe, written as FASTQ by the host form of the
generator (bit-identical to the device form), the archive of `colord_amd/colord_hip` (same library, reference part cut)
divided by the archive of the unmodified reference (`oracle/_ref/colord`), which is also the timed `cpu_baseline`.

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

T_PROCESS_START = time.time()
# The pipeline keeps up to ten HIP streams busy (three contexts with side streams, the encode lane); with the runtime's default
# of 4 hardware queues, streams that share a queue serialise (measured at 50 Gbases: 41.4 s/step with 4, 39.6 with 8, 39.3 with
# 16).  Read by the HIP runtime when it initialises, so it has to be in the environment before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 TB/s achievable)
README_DERIVED_GBASES_S = 0.030   # the only published figure (README time/size of the `memory` preset, hardware not stated)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bases", type=float, default=5.0e10, help="synthetic bases of the WHOLE input (all GPUs together; BASELINE.json: 50 Gbases)")
    ap.add_argument("--coverage", type=float, default=16.7, help="genome = bases / coverage (50 Gbases over 3 Gb)")
    ap.add_argument("--chunk-bases", type=float, default=1.0e9, help="bases per chunk of the streaming compressor")
    ap.add_argument("--k", type=int, default=0, help="k-mer length; 0 = the reference's choice for this input size (compression.cpp:62-93)")
    ap.add_argument("--a", type=int, default=0, help="anchor length; 0 = the reference's choice")
    ap.add_argument("--cpu-sample-bases", type=float, default=1.0e9, help="bases of the sample the reference (CPU baseline) and the archive-size check run on, at the k / a of the main run")
    ap.add_argument("--pack-symbols", type=int, default=1 << 16,
                    help="part (= range-coder restart) size in symbols.  4194304 reproduces the reference's packs (defs.h:45) and its exact "
                         "bytes; smaller parts are equally valid archives (the reference decoder follows the part table), cost 8 flush "
                         "bytes each (+0.04 %% at 64 Ki) and expose the parallelism the per-part dependent chain needs")
    ap.add_argument("--deadline-s", type=float, default=1500.0,
                    help="wall-clock budget of the whole process (the driver stops a run after 1800 s).  If (steps + warmup) passes over --bases "
                         "cannot finish inside it — estimated up front at 1.1 Gbases/s/GPU, then checked against the first warm-up pass — the input "
                         "is cut to a prefix of the file that can, and `config.workload` says so")
    ap.add_argument("--e2e-bases", type=float, default=2.0e10, help="bases of the FASTQ the command-line compressor is timed on, file to archive (T_e2e); 0: skip")
    ap.add_argument("--no-qual", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-cut", action="store_true", help="skip the extra pass with the reference's 4-Mi-symbol coder parts (the byte-identical mode)")
    return ap.parse_args()


# ONT default ("memory") preset: arg_parse.cpp:89-408 / SURVEY App. B
PRESET = dict(f=12, ci=4, cs=80, c=5, g=1.0, exponent=1.0, min_part_alt=64, max_rec=3)


QA = (2, 0, 1, (7, 14, 26), ())    # ONT default quality mode: 4-avg at level 1 (arg_parse.cpp:410-450): (mode, source, level, thresholds, -)


def kmer_anchor_len(bases: float):
    """adjustKmerAndAnchorLen (compression.cpp:62-93) on the number of bases."""
    for lim, k, a in ((1e9, 20, 16), (4e9, 21, 18), (16e9, 23, 21), (48e9, 24, 22), (128e9, 25, 22)):
        if bases < lim:
            return k, a
    return 26, 23


class StepTimes:
    """Per-kernel HIP-event times (context streams) accumulated by colord_amd.device.Context."""
    def __init__(self, *ctxs):
        self.ms, self.launches, self.bytes, self.cells = {}, {}, {}, {}
        for ctx in ctxs:
            if ctx is None:
                continue
            for n, v in ctx.acc.items():
                self.ms[n] = self.ms.get(n, 0.0) + v[0]
                self.launches[n] = self.launches.get(n, 0) + v[1]
                self.bytes[n] = self.bytes.get(n, 0.0) + v[2]
                self.cells[n] = self.cells.get(n, 0.0) + (v[3] if len(v) > 3 else 0.0)


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


def params_for(k: int, a: int) -> dict:
    p = PRESET
    return dict(k=k, f=p["f"], ci=p["ci"], cs=p["cs"], c=p["c"], anchor_len=a, min_part_alt=p["min_part_alt"], max_rec=p["max_rec"], min_anchors=1,
                level=1, source=0, sparse=1, sparse_g=p["g"], sparse_exponent=p["exponent"], cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)


class Shard:
    """This rank's reads, resident in HBM as chunks of whole reader packs: (arena, part bounds, pack bounds, quals, base offsets)."""

    def __init__(self, ctx, table, r0: int, r1: int, chunk_bases: float, pack_symbols: int, with_quals: bool):
        from colord_amd import ontsim
        self.chunks, self.n_reads, self.n_bases = [], r1 - r0, 0
        # lengths of all reads first (cheap): the reader packs are cut over the rank's whole read sequence (in_reads.cpp:62-77)
        lens = []
        sub = [r0 + x for x in _sub_cuts(table, r0, r1, 4e9)]
        for a, b in zip(sub[:-1], sub[1:]):
            lens.append(_device_lengths(table, ctx.device, a, b))
        lens = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
        packs = reference_part_bounds(lens, 1 << 22)
        acc = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
        cuts, target = [0], chunk_bases
        for p in range(1, len(packs)):
            if acc[packs[p]] >= target or p == len(packs) - 1:
                cuts.append(p)
                target = acc[packs[p]] + chunk_bases
        for ca, cb in zip(cuts[:-1], cuts[1:]):
            a, b = int(packs[ca]), int(packs[cb])
            codes, off, quals = ontsim.device_reads(table, ctx.device, r0 + a, r0 + b, with_quals=with_quals)
            assert np.array_equal((off[1:] - off[:-1]).cpu().numpy().astype(np.uint32), lens[a:b])
            arena = ctx.pack_reads(codes, off)
            del codes
            est = (packs[ca:cb + 1] - packs[ca]).astype(np.uint32)
            parts = est if pack_symbols == (1 << 22) else reference_part_bounds(lens[a:b], pack_symbols)
            self.chunks.append((arena, parts, est, quals, off))
            self.n_bases += int(arena.total_bases)
        self.n_parts = sum(len(c[1]) - 1 for c in self.chunks)

    def free(self):
        for c in self.chunks:
            c[0].free()

    def keep_prefix(self, n_chunks: int):
        """Drops the trailing chunks (the input becomes a prefix of this rank's part of the file)."""
        for c in self.chunks[n_chunks:]:
            c[0].free()
        self.chunks = self.chunks[:n_chunks]
        self.n_bases = sum(int(c[0].total_bases) for c in self.chunks)
        self.n_reads = sum(int(c[0].n_reads) for c in self.chunks)
        self.n_parts = sum(len(c[1]) - 1 for c in self.chunks)


def _sub_cuts(table, r0, r1, bases):
    acc = np.cumsum(table.len_src[r0:r1].astype(np.int64))
    out, base = [0], 0
    while out[-1] < r1 - r0:
        i = min(max(int(np.searchsorted(acc, base + bases, side="left")) + 1, out[-1] + 1), r1 - r0)
        out.append(i)
        base = int(acc[i - 1])
    return out


def _device_lengths(table, device, r0, r1):
    from colord_amd import ontsim
    L = ontsim._lib()
    n = r1 - r0
    d_start = torch.from_numpy(table.start[r0:r1].view(np.int64)).to(device)
    d_len = torch.from_numpy(table.len_src[r0:r1].view(np.int32)).to(device)
    d_strand = torch.from_numpy(table.strand[r0:r1]).to(device)
    out = torch.empty(n, dtype=torch.int32, device=device)
    if L.os_dev_lengths(table.seed, table.gseed, d_start.data_ptr(), d_len.data_ptr(), d_strand.data_ptr(), r0, n, out.data_ptr(), torch.cuda.current_stream(device).cuda_stream):
        raise RuntimeError("os_dev_lengths failed")
    return out.cpu().numpy().view(np.uint32)


class HostSink:
    """Where the parts of a pass end up: pinned host buffers, filled chunk by chunk from two staging buffers per stream on the
    device while the next chunk is coded (a copy stream of its own; a staging buffer is reused two chunks later, after its copy)."""

    def __init__(self, device, dna_cap: int, qual_cap: int, stage_dna: int, stage_qual: int):
        self.device = device
        self.h_dna = torch.empty(dna_cap, dtype=torch.uint8, pin_memory=True)
        self.h_qual = torch.empty(qual_cap, dtype=torch.uint8, pin_memory=True) if qual_cap else None
        self.d_dna = [torch.empty(stage_dna, dtype=torch.uint8, device=device) for _ in range(2)]
        self.d_qual = [torch.empty(stage_qual, dtype=torch.uint8, device=device) for _ in range(2)] if qual_cap else None
        self.stream = torch.cuda.Stream(device)
        self.events = [None, None]
        self.reset()

    def reset(self):
        self.do = self.qo = self.i = 0
        self.dna_sizes, self.qual_sizes = [], []

    def stage(self):
        k = self.i & 1
        if self.events[k] is not None:
            self.events[k].synchronize()                    # the copy out of this buffer, two chunks ago
        return self.d_dna[k], (self.d_qual[k] if self.d_qual is not None else None)

    def take(self, n_dna: int, n_qual: int, dsz, qsz):
        k = self.i & 1
        with torch.cuda.stream(self.stream):                 # (encode() returned: the parts are complete)
            self.h_dna[self.do:self.do + n_dna].copy_(self.d_dna[k][:n_dna], non_blocking=True)
            if self.h_qual is not None and n_qual:
                self.h_qual[self.qo:self.qo + n_qual].copy_(self.d_qual[k][:n_qual], non_blocking=True)
            ev = torch.cuda.Event(); ev.record(self.stream)
        self.events[k] = ev
        self.do += n_dna; self.qo += n_qual; self.i += 1
        self.dna_sizes.append(np.array(dsz, dtype=np.uint64)); self.qual_sizes.append(np.array(qsz, dtype=np.uint64))

    def finish(self):
        self.stream.synchronize()


def pass_digest(sink: "HostSink | None", dna_out, qual_out, info: dict) -> list:
    """Digest of EVERY part of the pass that just ended: per chunk one xxh3-128 over (part sizes + part bytes) of each stream, taken from
    where the step left them (the pinned host buffers at N = 1, the device buffers else).  The timed passes must all give the same
    list: nothing of a pass may depend on how its six contexts happened to interleave (the race detector the verdict of round 3
    asked for).  Runs between the per-step timing brackets, never inside one."""
    import xxhash
    from concurrent.futures import ThreadPoolExecutor
    if sink is None:
        out = []
        for name, buf, n in (("dna", dna_out, info["dna_bytes"]), ("qual", qual_out, info["qual_bytes"])):
            if buf is None or not n:
                continue
            w = buf[:n - n % 8].view(torch.int64)
            acc = 0
            for a in range(0, w.numel(), 1 << 24):          # wrapping 64-bit position-weighted sums, 128 MB at a time
                x = w[a:a + (1 << 24)]
                acc = (acc * 1000003 + int((x * (torch.arange(x.numel(), device=x.device, dtype=torch.int64) * 2654435761 + 1)).sum().item())) & ((1 << 64) - 1)
            out.append(f"{name}:{n}:{acc:016x}:{bytes(buf[n - n % 8:n].cpu().numpy()).hex()}")
        return out
    jobs = []
    for name, h, sizes in (("dna", sink.h_dna, sink.dna_sizes), ("qual", sink.h_qual, sink.qual_sizes)):
        if h is None:
            continue
        hv, o = h.numpy(), 0
        for ci, sz in enumerate(sizes):
            n = int(sz.sum())
            jobs.append((name, ci, sz, hv[o:o + n]))
            o += n

    def one(j):
        name, ci, sz, view = j
        x = xxhash.xxh3_128()
        x.update(sz.tobytes())
        x.update(memoryview(view))                          # (xxhash releases the GIL on large buffers)
        return f"{name}[{ci}]:{len(sz)}:{len(view)}:{x.hexdigest()}"
    with ThreadPoolExecutor(16) as ex:
        return list(ex.map(one, jobs))


def hot_path_step(ctx, qctx, shard: Shard, prm: dict, with_qual: bool, exchange, dna_out, qual_out, expected_bases: int, ref_cut: bool = False, sink: "HostSink | None" = None, part_sizes: "list | None" = None):
    """One pass of the whole compress data path over the shard (all chunks).  Returns sizes for reporting.
    ref_cut: the coder parts are the reference's reader packs (4 Mi symbols) instead of the bench's --pack-symbols.
    sink: the parts go to host memory chunk by chunk (T_core); else they stay in dna_out / qual_out on the device."""
    from colord_amd import parallel as par
    qa = QA if with_qual else None
    cmp_ = ctx.compressor(prm, qa, qctx, exchange, expected_bases=expected_bases)
    try:
        for ch in shard.chunks:
            cmp_.count_add(ch[0])
        if exchange is not None:
            exchange.phase = "kmers"                         # exchange 1: k-mers to their owners, kept set back to everybody
        st = cmp_.count_finish()
        for ch in shard.chunks:
            cmp_.refs_add(ch[0])
        if exchange is not None:
            exchange.phase = "refs"                          # exchange 2: reference reads and index entries to everybody
        cmp_.refs_finish()
        if exchange is not None:
            exchange.phase = ""
        do, qo = 0, 0
        tot = dict(n_anchors=0, tuple_bytes=0, dna_bytes=0, qual_bytes=0)
        if sink is not None:
            sink.reset()
        if not os.environ.get("BENCH_NO_LOOKAHEAD"):
            for ch in shard.chunks:                          # every chunk is resident: announce them all, the lanes keep lanes + 1 ahead
                cmp_.prepare(ch[0], ch[2], ch[2] if ref_cut else ch[1], ch[3] if with_qual else None, ch[4])
        for arena, parts, est, quals, off in shard.chunks:
            if sink is not None:
                d_dst, q_dst = sink.stage()
            else:
                d_dst, q_dst = dna_out[do:], (qual_out[qo:] if with_qual else None)
            _, dsz, _, qsz, inf = cmp_.encode(arena, est if ref_cut else parts, est, quals, off, d_dst, q_dst if with_qual else None)
            if sink is not None:
                sink.take(inf["dna_bytes"], inf["qual_bytes"], dsz, qsz)
            if part_sizes is not None:
                part_sizes.append((np.array(dsz, dtype=np.uint64), np.array(qsz, dtype=np.uint64)))
            do += inf["dna_bytes"]; qo += inf["qual_bytes"]
            for k_ in tot:
                tot[k_] += inf[k_]
        if sink is not None:
            sink.finish()
        info = cmp_.info()
    except Exception:
        if exchange is not None and exchange.err is not None:
            raise exchange.err
        raise
    finally:
        cmp_.free()
    if par.world() > 1:
        # SURVEY §8e "collective for results": the parts of every rank go to the rank that writes the archive
        t_g = time.perf_counter()
        par.gather_to_root(dna_out[:do])
        if with_qual:
            par.gather_to_root(qual_out[:qo])
        torch.cuda.synchronize()
        if exchange is not None:
            exchange.log.append(("parts", "gather_to_root", time.perf_counter() - t_g, 0))
    return dict(tot_kmers=int(st.tot_kmers), kept=int(st.n_unique_counted), refs=info["n_refs_total"], sparse_range=info["sparse_range"],
                anchors=tot["n_anchors"], tuple_bytes=tot["tuple_bytes"], dna_bytes=tot["dna_bytes"], qual_bytes=tot["qual_bytes"], parts=shard.n_parts, chunks=len(shard.chunks))


def quantised_quals_4avg(q: np.ndarray, off: np.ndarray, thresholds=(7, 14, 26)) -> np.ndarray:
    """What a decoder returns for `-q 4-avg` (the ONT default): the bin of a quality q - 33 is the number of thresholds <= it; per read
    and bin the encoder codes A = (uint32)(double(sum) / double(count) * 256) (quality_coder_impl.cpp:438-450, 821-834) and the decoder
    spreads it by error diffusion, as += A / 256; v = (uint32)(as - qs); qs += v (quality_coder_impl.cpp:506-559): the j-th base of a bin
    gets floor(j A / 256) - floor((j - 1) A / 256) — integers throughout (A / 256 and its multiples are exact doubles).  q: ASCII
    qualities of whole reads back to back, off: their offsets (starting at 0)."""
    n_reads = len(off) - 1
    v = q.astype(np.int64) - 33
    bins = np.zeros(len(q), np.int64)
    for t in thresholds:
        bins += v >= t
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64), np.diff(off).astype(np.int64))
    key = read_of * 4 + bins
    cnt = np.bincount(key, minlength=n_reads * 4).astype(np.float64)
    sm = np.bincount(key, weights=v.astype(np.float64), minlength=n_reads * 4)      # (sums < 2^53: exact)
    with np.errstate(invalid="ignore", divide="ignore"):
        A = np.where(cnt > 0, np.floor(sm / np.where(cnt > 0, cnt, 1.0) * 256.0), 0.0).astype(np.int64)
    out = np.empty(len(q), np.uint8)
    for b in range(4):                                       # rank j (1-based) of every base among the bases of its read and bin
        m = bins == b
        cs = np.cumsum(m, dtype=np.int64)
        start = cs[off[:-1].astype(np.int64)] - m[off[:-1].astype(np.int64)] if len(q) else cs[:0]
        j = (cs - np.repeat(start, np.diff(off).astype(np.int64)))[m]
        a = A[key[m]]
        out[m] = ((j * a) // 256 - ((j - 1) * a) // 256 + 33).astype(np.uint8)
    return out


def round_trip_check(ctx, table, shard: Shard, sink: HostSink, prm: dict, info: dict, r0: int, max_bases: float = 3.0e8, qa=None):
    """Decodes the first `dna` AND `qual` parts of the pass that is in `sink` (host memory) with the library's host decoders — the inverse
    path, csrc/decode.hip — and compares the bases with the generator's and the qualities with the generator's after the reference's
    4-avg quantisation (quantised_quals_4avg).  (A part needs every reference read before it, so the check runs from the start of the
    stream; its length is bounded by the decoders' speed, ~40 Mbases/s on one host thread each.)"""
    import ctypes as C
    from colord_amd import _native as N, ontsim
    lib = N.load()
    d = N._P()
    if lib.cl_dna_decoder_create(prm["c"], prm["level"], 0, 0, 0 if prm["sparse"] else 1, int(info["sparse_range"]), float(prm["sparse_exponent"]), C.byref(d)) != 0:
        return {"ok": False, "why": "cl_dna_decoder_create"}
    qd = None
    if qa is not None and sink.h_qual is not None:
        mode, source, level, fwd, rev = qa
        Q = N.QualParams(mode=mode, source=source, level=level, n_fwd=len(fwd), n_rev=len(rev))
        for i, v in enumerate(fwd):
            Q.fwd[i] = v
        for i, v in enumerate(rev):
            Q.rev[i] = v
        qd = N._P()
        if lib.cl_qual_decoder_create(C.byref(Q), C.byref(qd)) != 0:
            lib.cl_dna_decoder_free(d)
            return {"ok": False, "why": "cl_qual_decoder_create"}
    arena, parts = shard.chunks[0][0], shard.chunks[0][1]
    sizes = sink.dna_sizes[0]
    h = sink.h_dna.numpy()
    qsizes = sink.qual_sizes[0] if qd is not None else None
    hq = sink.h_qual.numpy() if qd is not None else None
    t0 = time.time()
    o = 0; qo = 0; n_reads = 0; n_bases = 0; ok = True; why = ""; qual_ok = qd is not None; n_qual = 0
    codes, off, quals = ontsim.device_reads(table, ctx.device, r0, r0 + int(arena.n_reads), with_quals=qd is not None)
    codes, off = codes.cpu().numpy(), off.cpu().numpy()
    quals = quals.cpu().numpy() if qd is not None else None
    for p in range(len(parts) - 1):
        nr = int(parts[p + 1] - parts[p]); sz = int(sizes[p])
        exp = codes[off[parts[p]]:off[parts[p + 1]]]
        buf = np.ascontiguousarray(h[o:o + sz])
        out = np.empty(len(exp) + 64, np.uint8); offs = np.zeros(nr + 1, np.uint64); got = C.c_uint64(0)
        st = lib.cl_dna_decode_part(d, buf.ctypes.data, sz, nr, out.ctypes.data, len(out), offs.ctypes.data, C.byref(got))
        if st != 0 or got.value != len(exp) or not np.array_equal(out[:len(exp)] & 7, exp):
            ok = False; why = f"dna part {p}: status {st}, {got.value} bases decoded, {len(exp)} expected"
            break
        if qd is not None:
            qsz = int(qsizes[p])
            qbuf = np.ascontiguousarray(hq[qo:qo + qsz])
            qout = np.zeros(len(exp) + 64, np.uint8)
            st = lib.cl_qual_decode_part(qd, qbuf.ctypes.data, qsz, out.ctypes.data, offs.ctypes.data, nr, qout.ctypes.data)
            a, b = int(off[parts[p]]), int(off[parts[p + 1]])
            want = quantised_quals_4avg(quals[a:b], off[parts[p]:parts[p + 1] + 1] - off[parts[p]], qa[3])
            if st != 0 or not np.array_equal(qout[:len(exp)], want):
                ok = False; qual_ok = False; why = f"qual part {p}: status {st}, first difference at base {int(np.argmax(qout[:len(exp)] != want)) if st == 0 else -1}"
                break
            qo += qsz; n_qual += len(exp)
        o += sz; n_reads += nr; n_bases += len(exp)
        if n_bases >= max_bases:
            break
    lib.cl_dna_decoder_free(d)
    if qd is not None:
        lib.cl_qual_decoder_free(qd)
    return {"ok": ok, "why": why, "parts": p + 1, "reads": n_reads, "bases": n_bases, "qual_checked": bool(qual_ok and ok and n_qual > 0), "quals": n_qual, "seconds": round(time.time() - t0, 1),
            "what": "the first `dna` and `qual` parts of the last timed pass, host memory -> cl_dna_decode_part / cl_qual_decode_part -> compared with the "
                    "generator's bases and with its qualities after the reference's 4-avg quantisation (bench.py quantised_quals_4avg)"}


def cpu_baseline_and_size_check(ctx, qctx, sample_bases: float, coverage: float, pack_symbols: int, k: int, a: int):
    """The UNMODIFIED reference binary (oracle/_ref/colord, built by oracle/Makefile.ref) timed on this host's cores on a
    bounded sample of the same synthetic recipe, and — on the very same FASTQ — the archive of colord_hip (this library)
    against the reference's archive: the second half of the metric.  Both compressors get the k-mer and anchor lengths of the main
    run (`-k`, `-a`: at 50 Gbases the reference picks k = 25 / a = 22, compression.cpp:84-88; for the sample's own size it would
    pick shorter ones), so the archive comparison and the baseline are at the parameters the headline number is measured with."""
    from colord_amd import ontsim, archive as AR
    ref = os.path.join(ROOT, "oracle", "_ref", "colord")
    ours = os.path.join(ROOT, "colord_amd", "colord_hip")
    if not os.path.exists(ref):
        return None, None
    cores = os.cpu_count() or 1
    table = ontsim.ReadTable(seed=101, genome_len=max(1_000_000, int(sample_bases / coverage)), target_bases=int(sample_bases))
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "sample.fastq")
        n_bases = ontsim.write_fastq(table, fq)
        os.sync()                                           # (the input's own write-back is not part of what is timed: without this the timed command's output waits behind 10 GB of dirty pages)
        ka = ["-k", str(k), "-a", str(a)]
        # (the unmodified reference has been seen to die of SIGSEGV on a 256-thread host, once in several runs of the same command: it
        # is run again — then with fewer threads — before the baseline is given up; `cores` reports the threads of the run that counted)
        dt, rc, tries = None, 0, 0
        for threads in (cores, cores, min(cores, 64), min(cores, 16)):
            tries += 1
            t0 = time.time()
            rc = subprocess.call([ref, "compress-ont", "-t", str(threads)] + ka + [fq, os.path.join(tmp, "ref.colord")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if rc == 0:
                dt, cores = time.time() - t0, threads
                break
        if dt is None:
            return {"value": None, "unit": "Gbases/s", "cores": cores, "kind": "reference",
                    "sample": f"oracle/_ref/colord compress-ont {' '.join(ka)} on {n_bases} synthetic ONT bases failed {tries} times (last exit status {rc})"}, None
        ref_size = os.path.getsize(os.path.join(tmp, "ref.colord"))
        ref_arc = AR.read_archive(os.path.join(tmp, "ref.colord"))
        ref_streams = {n: sum(len(p) for _, p in s.parts) for n, s in ref_arc.items()}
        cb = {"value": n_bases / dt / 1e9, "unit": "Gbases/s", "cores": cores, "kind": "reference",
              "sample": f"oracle/_ref/colord compress-ont -t {cores} -k {k} -a {a} on {n_bases} synthetic ONT bases ({table.n_reads} reads, same recipe, genome {table.genome_len} bp), "
                        f"whole compressor (parsing, header stream and archive included); {dt:.2f} s wall, archive {ref_size} B = {ref_size / n_bases:.4f} B/base"
                        + (f"; run {tries} of the command (the earlier ones crashed inside the reference)" if tries > 1 else "")}
        # SURVEY 8d asks for the reference at `-t 8` beside `-t <all cores>`: on a fifth of the sample (a prefix of the same reads), so that
        # the leg stays within half a minute; both runs are whole compressors, parsing and archive included
        try:
            t8 = ontsim.ReadTable(seed=101, genome_len=max(1_000_000, int(sample_bases / coverage)), target_bases=int(sample_bases / 5))
            fq8 = os.path.join(tmp, "sample_t8.fastq")
            nb8 = ontsim.write_fastq(t8, fq8)
            os.sync()
            t0 = time.time()
            rc8 = subprocess.call([ref, "compress-ont", "-t", "8"] + ka + [fq8, os.path.join(tmp, "ref8.colord")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt8 = time.time() - t0
            cb["t8"] = {"value": nb8 / dt8 / 1e9 if rc8 == 0 else None, "unit": "Gbases/s", "cores": 8, "sample": f"the same command with -t 8 on {nb8} bases of the same recipe; {dt8:.2f} s wall" + ("" if rc8 == 0 else f" (exit status {rc8})")}
            # the inverse path on the same small sample: `colord_hip decompress` (host code: one dependent chain per stream and model domain,
            # three stream threads as the reference's decompressor) beside the reference's own decompressor on the reference's archive
            if rc8 == 0 and os.path.exists(ours):
                dec = {}
                for who, exe, arc in (("colord_hip", ours, os.path.join(tmp, "ref8.colord")), ("reference", ref, os.path.join(tmp, "ref8.colord"))):
                    t0 = time.time()
                    rcd = subprocess.call([exe, "decompress", arc, os.path.join(tmp, "dec8.fastq")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                    dtd = time.time() - t0
                    dec[who] = {"value": nb8 / dtd / 1e9 if rcd == 0 else None, "unit": "Gbases/s", "seconds": round(dtd, 2), "threads": 3}
                    if rcd == 0 and who == "colord_hip":
                        import hashlib
                        h = hashlib.sha256()
                        with open(os.path.join(tmp, "dec8.fastq"), "rb") as f:
                            for blk in iter(lambda: f.read(1 << 24), b""):
                                h.update(blk)
                        dec["sha256_colord_hip"] = h.hexdigest()
                    elif rcd == 0:
                        import hashlib
                        h = hashlib.sha256()
                        with open(os.path.join(tmp, "dec8.fastq"), "rb") as f:
                            for blk in iter(lambda: f.read(1 << 24), b""):
                                h.update(blk)
                        dec["same_output_as_reference"] = bool(h.hexdigest() == dec.get("sha256_colord_hip"))
                    if os.path.exists(os.path.join(tmp, "dec8.fastq")):
                        os.remove(os.path.join(tmp, "dec8.fastq"))
                dec["sample"] = f"the reference's archive of the {nb8}-base sample above, file to FASTQ; both decoders are host code (no GPU)"
                cb["decompress"] = dec
            os.remove(fq8)
        except Exception as e:
            cb["t8"] = {"value": None, "error": repr(e)[:200]}
        size = {"sample_bases": n_bases, "k": k, "a": a, "ref_archive_bytes": ref_size, "ref_dna_bytes": ref_streams.get("dna"), "ref_qual_bytes": ref_streams.get("qual")}
        # (1) the command-line compressor of this build on the same file: whole archive, reference part cut
        if os.path.exists(ours):
            t0 = time.time()
            r = subprocess.run([ours, "compress-ont"] + ka + [fq, os.path.join(tmp, "hip.colord")], capture_output=True, text=True)
            if r.returncode == 0:
                hs = os.path.getsize(os.path.join(tmp, "hip.colord"))
                arc = AR.read_archive(os.path.join(tmp, "hip.colord"))
                same = all([p for _, p in arc[n].parts] == [p for _, p in ref_arc[n].parts] for n in ("dna", "qual", "header", "meta"))
                size.update({"hip_archive_bytes": hs, "archive_vs_ref": hs / ref_size, "cli_wall_s": round(time.time() - t0, 2),
                             "streams_byte_identical_to_ref": bool(same)})
            else:
                size["cli_error"] = (r.stderr or r.stdout)[-300:]
        # (2) the bench path itself (device generator, chunked compressor) on the same reads, with the bench's part cut and with the reference's
        for cut, name in ((1 << 22, "ref_cut"), (pack_symbols, f"cut_{pack_symbols}")):
            shard = Shard(ctx, table, 0, table.n_reads, 1e9, cut, True)
            dna_out = torch.empty(int(shard.n_bases * 0.5) + (1 << 20), dtype=torch.uint8, device=ctx.device)
            qual_out = torch.empty(int(shard.n_bases * 0.6) + (1 << 20), dtype=torch.uint8, device=ctx.device)
            psz = []
            inf = hot_path_step(ctx, qctx, shard, params_for(k, a), True, None, dna_out, qual_out, shard.n_bases, ref_cut=(cut == 1 << 22), part_sizes=psz)
            size[f"streams_vs_ref_{name}"] = (inf["dna_bytes"] + inf["qual_bytes"]) / (ref_streams["dna"] + ref_streams["qual"])
            size[f"dna_qual_bytes_{name}"] = [inf["dna_bytes"], inf["qual_bytes"]]
            if cut == 1 << 22:
                # the parts the bench path wrote against the parts of the reference's archive, by SHA-256 (and part by part: the sizes)
                import hashlib
                eq = {}
                for nm, buf, tot, col in (("dna", dna_out, inf["dna_bytes"], 0), ("qual", qual_out, inf["qual_bytes"], 1)):
                    mine = bytes(buf[:tot].cpu().numpy())
                    theirs = b"".join(p for _, p in ref_arc[nm].parts)
                    my_sizes = [int(x) for ch in psz for x in ch[col]]
                    eq[nm] = bool(hashlib.sha256(mine).digest() == hashlib.sha256(theirs).digest() and my_sizes == [len(p) for _, p in ref_arc[nm].parts])
                    size[f"bench_{nm}_sha256"] = hashlib.sha256(mine).hexdigest()
                    size[f"ref_{nm}_sha256"] = hashlib.sha256(theirs).hexdigest()
                size["bench_parts_sha256_equal_ref"] = bool(eq["dna"] and eq["qual"])
            shard.free()
            del dna_out, qual_out
            if cut == pack_symbols:
                break
    return cb, size


def e2e_cli(bases: float, coverage: float, k: int, a: int, part_symbols: int, gpus: int = 1, gpu_list=None, timeout_s: float = 900.0):
    """T_e2e: `colord_hip compress-ont` from open(FASTQ) to close(archive) — parsing, upload, all three passes, the header
    stream, the archive — on a synthetic FASTQ of the same recipe written by the host generator, k / a of the main run: once with the
    part cut of the headline number (`--part-symbols`, what `value` is measured with) and once with the reference's (the archive is
    the reference's, byte for byte)."""
    from colord_amd import ontsim
    ours = os.path.join(ROOT, "colord_amd", "colord_hip")
    if not os.path.exists(ours) or bases <= 0:
        return None
    import shutil
    free = shutil.disk_usage(tempfile.gettempdir()).free    # the FASTQ takes 2 bytes per base, the archive 0.4: a smaller sample where the disk is short
    if free < 2.6 * bases + (4 << 30):
        bases = max(0.0, (free - (4 << 30)) / 2.6)
        if bases < 2.0e8:
            return {"error": f"{free / 1e9:.0f} GB free in {tempfile.gettempdir()}: no room for the FASTQ of the command-line leg"}
    table = ontsim.ReadTable(seed=103, genome_len=max(1_000_000, int(bases / coverage)), target_bases=int(bases))
    with tempfile.TemporaryDirectory() as tmp:
        fq = os.path.join(tmp, "e2e.fastq")
        t0 = time.time()
        n_bases = ontsim.write_fastq(table, fq)
        os.sync()                                           # (the input's own write-back is not part of what is timed: without this the timed command's output waits behind 10 GB of dirty pages)
        t_gen = time.time() - t0
        out = {}
        if gpus == 1:
            # One UNTIMED run of the same command first (as the timed passes have their warm-up passes): the first GPU process on a device that has
            # idled — here: while the FASTQ was written — runs 1.4-1.5 x slower, every kernel and copy of it: 13.0-15.1 s against 9.6-10.4 s for the
            # runs after it at 20 Gbases (profiles/r06_e2e_20Gbases.txt; a 1.3-s run on 0.5 Gbases in front was enough on one box and not on another)
            warm = os.path.join(tmp, "e2e_warm.colord")
            try:
                subprocess.run([ours, "compress-ont", "-k", str(k), "-a", str(a), "--part-symbols", str(part_symbols), fq, warm], capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                pass
            if os.path.exists(warm):
                os.remove(warm)
        for name, ps in (("headline_cut", part_symbols), ("ref_cut", 1 << 22)):
            if name == "ref_cut" and ps == part_symbols:
                continue
            multi = []
            if gpus > 1:                                    # the product's multi-GPU host: one rank thread per GPU, RCCL (host-staged where ranks share a GPU)
                gl = gpu_list or list(range(gpus))
                multi = ["--gpus", str(gpus), "--gpu-list", ",".join(str(x) for x in gl), "--transport", "rccl" if len(set(gl)) == len(gl) else "host"]
                if name == "ref_cut":
                    continue
            arc = os.path.join(tmp, f"e2e_{name}.colord")      # (a new file per run: replacing a multi-GB file makes close() wait for its blocks — ext4's replace-via-truncate rule, 0.8 s at 20 Gbases)
            cmd = [ours, "compress-ont", "-v", "-k", str(k), "-a", str(a), "--part-symbols", str(ps)] + multi + [fq, arc]
            time.sleep(8.0)                                 # the driver clears what the process before gave back (this one's pools, the run before) while the next one starts:
            t0 = time.time()                                # measured 5.2-5.3 s after a pause against 5.5-6.8 s back to back (profiles/r05_e2e_pause_5Gbases.txt)
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                out[name] = {"error": f"no archive after {timeout_s:.0f} s: {' '.join(cmd[:-2])}"}
                continue
            dt = time.time() - t0
            if r.returncode != 0:
                out[name] = {"error": (r.stderr or r.stdout)[-300:]}
                continue
            phases = [l.strip() for l in r.stderr.splitlines() if l.strip().startswith(("[", "# "))][-16:]
            out[name] = {"value": n_bases / dt / 1e9, "unit": "Gbases/s", "seconds": round(dt, 2), "part_symbols": ps, "archive_bytes": os.path.getsize(arc), "phases": phases}
            os.remove(arc)
        first = out.get("headline_cut") or {}
        res = {"value": first.get("value"), "unit": "Gbases/s", "seconds": first.get("seconds"), "bases": n_bases, "fastq_bytes": os.path.getsize(fq), "fastq_written_in_s": round(t_gen, 1),
               "what": f"colord_hip compress-ont -k {k} -a {a} --part-symbols N" + (f" --gpus {gpus}" if gpus > 1 else "") + " file -> archive, whole process (mapped file indexed by several threads, chunks filled by parallel copies into "
                       f"pinned double buffers); `value` is with the headline's part cut ({part_symbols}), `ref_cut` with the reference's 4194304 (byte-identical archive). "
                       f"INPUT STATE: the FASTQ was written by this process just before, `sync`ed, and is PAGE-CACHE RESIDENT when the timed command starts (a warm-file number, "
                       f"not a cold-disk one); an 8-s pause precedes the timed command (the driver's clean-up of the memory the process before gave back is not part of it). "
                       f"PROTOCOL (round 6): at N = 1 this leg runs FIRST, before this process has touched the GPU — behind the timed passes, with this process's HIP context alive "
                       f"and device memory of its pools not yet back with the driver, the same command took 37-51 s instead of 10 (profiles/r06_n_*) — and ONE UNTIMED run of "
                       f"the same command precedes the timed one (the first GPU process on a device that idled while the FASTQ was written runs 1.4-1.5 x slower)"
                       + ("; the C++ host: one rank thread per GPU, exchanges over RCCL (or host-staged), each rank writes its own parts" if gpus > 1 else "")}
        res.update(out)
        return res


def multi_gpu_cli_leg(rank: int, world: int, args, k: int, a: int):
    """Every rank calls this after it has released its GPU memory.  Rank 0 waits until all have (a key per rank in the process group's
    store: host-side waits, no collective spinning on the GPUs the leg is about to use), runs `colord_hip --gpus N` on a FASTQ of
    --e2e-bases x N bases (at most 20 Gbases) and releases the others."""
    from datetime import timedelta
    from torch.distributed import distributed_c10d as c10d
    st = c10d._get_default_store()
    st.set(f"bench_freed_{rank}", b"1")
    if rank != 0:
        try:
            st.wait(["bench_e2e_done"], timedelta(seconds=1500))
        except Exception:
            pass
        return None
    res = None
    try:
        st.wait([f"bench_freed_{r}" for r in range(world)], timedelta(seconds=180))
        n_dev = torch.cuda.device_count()
        res = e2e_cli(min(args.e2e_bases * world, 2.0e10), args.coverage, k, a, args.pack_symbols, gpus=world, gpu_list=[r % n_dev for r in range(world)], timeout_s=900.0)
    finally:
        st.set("bench_e2e_done", b"1")
    return res


def load_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the newest committed PMC passes (tools/pmc_traffic.py -> profiles/r0N_traffic.json)."""
    path = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_traffic.json") for r in (6, 5, 4, 3)) if os.path.exists(p)), None)
    if path is None:
        return None, None
    t = json.load(open(path))
    # (the file says which commit of the kernels it was measured on: `measured_at_commit`, written when it is copied into profiles/)
    t["source"] = f"{os.path.basename(path)} (kernels as of commit {t.get('measured_at_commit', 'unknown')}): " + str(t.get("source"))
    key = kernel.split("<")[0].strip()
    if key == "k_sort_scatter":                             # (with / without a value array: tools/pmc_traffic.py keeps them apart)
        key += "<true>" if ", true" in kernel else "<false>"
    e = t.get("kernels", {}).get(key)
    if not e:
        return None, t.get("source")
    return e["hbm_bytes_per_launch"], t.get("source")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # `python bench.py --gpus N` launches itself: the same command shape as N = 1.  One rank per GPU over RCCL; on a box with fewer
            # GPUs than ranks the ranks share the GPUs and the collectives go over gloo (a functional run of the N-rank path, not a measurement)
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if torch.cuda.device_count() < args.gpus:
                env.setdefault("BENCH_BACKEND", "gloo")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                   os.path.abspath(__file__)] + sys.argv[1:]
            sys.stdout.flush(); sys.stderr.flush()
            os.execvpe(cmd[0], cmd, env)
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
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
    from colord_amd import ontsim, parallel as par

    preflight = None
    if world > 1:
        # N > 1 preflight, before any context exists: the C++ host's RCCL collectives alone (`colord_hip rccl-selftest`: all-gather, all-to-all-v
        # and all-gather-v with uneven and empty shares, every byte checked) on as many GPUs as the box has for the ranks — a broken fabric or a
        # broken RCCL shows here in seconds, with its message in the line, instead of as a hang of the 50-Gbase passes
        if rank == 0:
            exe = os.path.join(ROOT, "colord_amd", "colord_hip")
            n_self = min(world, torch.cuda.device_count())
            t0 = time.time()
            try:
                r_ = subprocess.run([exe, "rccl-selftest", "--gpus", str(n_self)], capture_output=True, text=True, timeout=300)
                preflight = {"command": f"colord_hip rccl-selftest --gpus {n_self}", "rc": r_.returncode, "seconds": round(time.time() - t0, 2), "gpus": n_self,
                             "output": ((r_.stdout or "") + (r_.stderr or "")).strip()[-400:]}
            except Exception as e:
                preflight = {"command": f"colord_hip rccl-selftest --gpus {n_self}", "rc": None, "seconds": round(time.time() - t0, 2), "error": repr(e)[:300]}
        dist.barrier()

    # N = 1: the command-line leg (T_e2e) FIRST, while this process has not touched the GPU (see `t_e2e.what`)
    e2e_first = None
    if world == 1 and args.e2e_bases > 0 and not args.no_cpu_baseline:
        k_, a_ = kmer_anchor_len(float(args.bases))
        k_, a_ = (args.k or k_), (args.a or a_)
        try:
            e2e_first = e2e_cli(args.e2e_bases, args.coverage, k_, a_, args.pack_symbols)
        except Exception as e:
            e2e_first = {"error": repr(e)[:400]}
    ctx = Context(local, timing=not os.environ.get("BENCH_NO_TIMING"))
    qctx = Context(local, timing=not os.environ.get("BENCH_NO_TIMING")) if not os.environ.get("BENCH_NO_OVERLAP") else None
    bases = float(args.bases)
    # what the passes may take: the deadline minus what is spent already, the CPU-baseline / size-check leg (~1.5 min) and a margin
    reserve_s = (100.0 if (world == 1 and not args.no_cpu_baseline) else 30.0) + 30.0
    pass_budget_s = max(10.0, args.deadline_s - (time.time() - T_PROCESS_START) - reserve_s - bases / 5e9)
    est_s = (args.steps + args.warmup) * bases / (2.0e9 * world)             # (25 s a pass of 50 Gbases on one GPU: measured 17-18)
    reduced = False
    if est_s > pass_budget_s:
        bases = max(1e8, pass_budget_s * 2.0e9 * world / (args.steps + args.warmup))
        reduced = True
    genome_len = max(1_000_000, int(bases / args.coverage))
    table = ontsim.ReadTable(seed=1, genome_len=genome_len, target_bases=int(bases))        # the same table on every rank
    k, a = kmer_anchor_len(bases)
    k, a = (args.k or k), (args.a or a)
    # this rank's contiguous range of the file: equal shares of the source bases
    acc = np.cumsum(table.len_src.astype(np.int64))
    r0 = int(np.searchsorted(acc, acc[-1] * rank / world, side="left")) if rank else 0
    r1 = int(np.searchsorted(acc, acc[-1] * (rank + 1) / world, side="left")) if rank + 1 < world else table.n_reads
    t_gen = time.perf_counter()
    shard = Shard(ctx, table, r0, r1, args.chunk_bases, args.pack_symbols, not args.no_qual)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    torch.cuda.empty_cache()                                # the generator's temporaries go back to the device (the library has its own pools)
    # where the parts go: at N = 1 to pinned host memory chunk by chunk, through two staging buffers per stream on the device (the step
    # ends with the parts in host memory); at N > 1 they stay on the device for the gather to rank 0
    sink, dna_out, qual_out = None, None, None
    if world == 1:
        cb = max((int(c[0].total_bases) for c in shard.chunks), default=0)
        sink = HostSink(ctx.device, int(shard.n_bases * 0.20) + (1 << 26), 0 if args.no_qual else int(shard.n_bases * 0.26) + (1 << 26),
                        int(cb * 0.30) + (1 << 24), int(cb * 0.36) + (1 << 24))
    else:
        dna_out = torch.empty(int(shard.n_bases * 0.20) + (1 << 26), dtype=torch.uint8, device=ctx.device)
        qual_out = None if args.no_qual else torch.empty(int(shard.n_bases * 0.26) + (1 << 26), dtype=torch.uint8, device=ctx.device)
    exchange = par.TorchExchange(ctx.device) if world > 1 else None
    prm = params_for(k, a)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        return hot_path_step(ctx, qctx, shard, prm, not args.no_qual, exchange, dna_out, qual_out, shard.n_bases, sink=sink)

    for w in range(args.warmup):
        t_w = time.perf_counter()
        step()
        if w == 0 and args.warmup + args.steps > 1:
            # the first pass is the measurement the estimate above lacked: if the remaining passes cannot finish inside the
            # deadline at this rate, every rank keeps the same leading fraction of its chunks (a prefix of its part of the file)
            torch.cuda.synchronize()
            t_pass = time.perf_counter() - t_w
            left = args.deadline_s - (time.time() - T_PROCESS_START) - reserve_s
            frac = torch.tensor([min(1.0, max(0.0, left) / max(1e-9, (args.warmup - 1 + args.steps) * t_pass))], dtype=torch.float64, device=ctx.device if backend == "nccl" else "cpu")
            if world > 1:
                dist.all_reduce(frac, op=dist.ReduceOp.MIN)
            if float(frac.item()) < 1.0:
                shard.keep_prefix(max(1, int(len(shard.chunks) * float(frac.item()))))
                reduced = True
    ctx.acc.clear()
    if qctx is not None:
        qctx.acc.clear()
    # K steps, each bracketed by barrier + synchronize on both sides; between two brackets (never inside one) the pass that just
    # ended is digested part by part where the step left it: `parts_digest_stable` = every timed pass gave the same bytes
    info, dt, step_s, own_s, digests = None, 0.0, [], [], []
    for _ in range(args.steps):
        sync()
        t0 = time.perf_counter()
        info = step()
        torch.cuda.synchronize()
        own_s.append(time.perf_counter() - t0)               # this rank's own work (before the closing barrier: the imbalance between ranks shows here)
        sync()
        step_s.append(time.perf_counter() - t0)
        dt += step_s[-1]
        if not os.environ.get("BENCH_NO_DIGEST"):
            digests.append(pass_digest(sink, dna_out, qual_out, info))
    digest_stable = bool(digests) and all(d == digests[0] for d in digests)
    digest_diff = ""
    for i, d in enumerate(digests):
        for x, y in zip(d, digests[0]):
            if x != y and not digest_diff:
                digest_diff = f"pass {i}: {x} != pass 0: {y}"
        if len(d) != len(digests[0]) and not digest_diff:
            digest_diff = f"pass {i}: {len(d)} entries != {len(digests[0])}"
    if world > 1 and digests:
        ok_t = torch.tensor([1 if digest_stable else 0], dtype=torch.int64, device=ctx.device if backend == "nccl" else "cpu")
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        digest_stable = bool(ok_t.item())
    red_dev = ctx.device if backend == "nccl" else torch.device("cpu")
    tdev = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tb = torch.tensor([shard.n_bases, info["dna_bytes"], info["qual_bytes"], shard.n_reads], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(tdev, op=dist.ReduceOp.MAX)
        dist.all_reduce(tb)
    dt = float(tdev.item())
    total_bases, total_dna, total_qual, total_reads = (int(x) for x in tb.tolist())
    multi = None
    if world > 1:
        # what a first run on N GPUs needs to be read: every rank's own step times, bases and stream bytes, and where its time between the
        # GPUs went (seconds / bytes received per exchange of the timed passes), gathered to rank 0
        mine = {"rank": rank, "device": int(local), "step_s": [round(x, 3) for x in step_s], "own_step_s": [round(x, 3) for x in own_s], "bases": int(shard.n_bases), "reads": int(shard.n_reads),
                "dna_bytes": int(info["dna_bytes"]), "qual_bytes": int(info["qual_bytes"]), "input_generation_s": round(t_gen, 1),
                "exchange": exchange.summary() if exchange is not None else None, "exchange_bytes_received": int(exchange.bytes_moved) if exchange is not None else 0}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            ex_tot = {}
            for r_ in allr:
                for k_, e in (r_["exchange"] or {}).items():
                    t_ = ex_tot.setdefault(k_, {"max_seconds_per_step": 0.0, "bytes_received_all_ranks_per_step": 0, "calls_per_step": e["calls"] / max(1, args.steps + args.warmup)})
                    t_["max_seconds_per_step"] = max(t_["max_seconds_per_step"], round(e["seconds"] / max(1, args.steps + args.warmup), 4))
                    t_["bytes_received_all_ranks_per_step"] += e["bytes_received"] // max(1, args.steps + args.warmup)
            multi = {"backend": backend, "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "process_group_ranks": dist.get_world_size(),
                     "devices": sorted({r_["device"] for r_ in allr}), "ranks_share_gpus": len({r_["device"] for r_ in allr}) < world,
                     "per_rank": allr, "exchange_s": ex_tot,
                     "exchange_note": "per exchange (`kmers`: k-mers to their owners + kept set to all, inside cl_compressor_count_finish; `refs`: reference reads + index entries to all, "
                                      "inside cl_compressor_refs_finish; `parts`: compressed parts to rank 0) the slowest rank's seconds per pass and the bytes all ranks received per pass, "
                                      "averaged over warm-up + timed passes (the callbacks do not know which pass they serve)",
                     "rccl_selftest": preflight}
    if os.environ.get("BENCH_NO_TIMING"):                   # diagnostic: the pass time without the per-kernel events (no JSON line)
        if rank == 0:
            print(f"[bench] no kernel events: {dt / args.steps * 1e3:.1f} ms per step", file=sys.stderr)
        return

    from colord_amd.device import _check
    torch.cuda.synchronize()
    for c_ in (ctx, qctx):                                  # (the kernel times of what completed last: collection never waits)
        if c_ is not None:
            _check(c_, 0)
    times = StepTimes(ctx, qctx)
    if rank == 0:
        # dominant kernel by measured HIP-event time on the context streams; `achieved` = the library's algorithmic HBM
        # byte count of those launches (per-kernel formulas in DESIGN.md) / their measured duration
        dom = max(times.ms, key=times.ms.get)
        launches = times.launches[dom]
        avg_ms = times.ms[dom] / launches
        traffic, traffic_src = load_traffic(dom)
        roof = {"bound": "hbm", "kernel": dom, "avg_ms": avg_ms, "launches_per_step": launches / args.steps,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": traffic, "traffic_source": traffic_src}
        if times.bytes.get(dom, 0) > 0:
            ach = times.bytes[dom] / (times.ms[dom] * 1e-3) / 1e9
            roof.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "alg_bytes_per_launch": times.bytes[dom] / launches})
        else:
            roof.update({"achieved": None, "frac": None})
        order = sorted(times.ms, key=times.ms.get, reverse=True)
        roof["kernel_ms_per_step"] = {n: round(times.ms[n] / args.steps, 3) for n in order[:40]}
        roof["kernel_achieved_GBps"] = {n: round(times.bytes[n] / (times.ms[n] * 1e-3) / 1e9, 1) for n in order if times.bytes.get(n, 0) > 0 and times.ms[n] > 0}
        # the aligners as what they are — dynamic-programming kernels: cell updates per second (rows x columns of every gap once; a
        # large gap's second and third sweep, Hirschberg, are not counted) over the summed HIP-event time of the class
        al = [n for n in times.ms if n.startswith(("k_align_", "k_giant_"))]
        al_cells, al_ms = sum(times.cells.get(n, 0.0) for n in al), sum(times.ms[n] for n in al)
        roof["aligners_gcups"] = {"value": al_cells / (al_ms * 1e-3) / 1e9 if al_ms > 0 else None, "cells_per_step": al_cells / args.steps, "kernel_ms_per_step": al_ms / args.steps,
                                  "by_kernel": {n: round(times.cells.get(n, 0.0) / (times.ms[n] * 1e-3) / 1e9, 1) for n in al if times.ms[n] > 0 and times.cells.get(n, 0.0) > 0},
                                  "what": "DP cells (rows x columns of every aligned gap, once) / summed HIP-event time of k_align_small<1..4>, k_align_quad, k_align_wave and the k_giant_* launches; "
                                          "kernels share the machine, so this is a rate inside the pipeline, not a peak"}
        # compulsory floor of the whole path (3.2 B/base) against the step time
        roof["whole_path_floor_frac"] = 3.2 * total_bases * args.steps / dt / 1e9 / (HBM_PEAK_GBS * world)
        # the same input once more in the byte-identical mode: coder parts = the reference's reader packs (4 Mi symbols; each part is
        # one dependent chain of the interval coder, 64 times longer than with the bench's default cut) — when the deadline allows
        ref_cut = None
        rt = None
        if world == 1 and not args.no_ref_cut and args.pack_symbols != (1 << 22):
            left = args.deadline_s - (time.time() - T_PROCESS_START) - reserve_s
            if left > 4.0 * (dt / args.steps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rt = round_trip_check(ctx, table, shard, sink, prm, info, r0, qa=QA if not args.no_qual else None) if sink is not None else None   # (before the sink is overwritten)
                warm2 = left > 9.0 * (dt / args.steps)
                if warm2:                                  # (an untimed pass first, as for the default cut: other buffer sizes, other pool shape)
                    hot_path_step(ctx, qctx, shard, prm, not args.no_qual, None, dna_out, qual_out, shard.n_bases, ref_cut=True, sink=sink)
                    torch.cuda.synchronize()
                t1 = time.perf_counter()
                inf2 = hot_path_step(ctx, qctx, shard, prm, not args.no_qual, None, dna_out, qual_out, shard.n_bases, ref_cut=True, sink=sink)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                ref_cut = {"part_symbols": 1 << 22, "ms_per_step": dt2 * 1e3, "value": total_bases / dt2 / 1e9, "unit": "Gbases/s", "steps": 1, "warmup": 1 if warm2 else 0,
                           "dna_bytes": inf2["dna_bytes"], "qual_bytes": inf2["qual_bytes"], "parts": sum(len(c[2]) - 1 for c in shard.chunks),
                           "stream_bytes_vs_default_cut": (inf2["dna_bytes"] + inf2["qual_bytes"]) / max(1, total_dna + total_qual),
                           "note": "same input, one pass, coder parts = the reference's reader packs: the streams are the reference's bytes (size_check.streams_vs_ref_ref_cut)"}
        if rt is None and sink is not None:
            rt = round_trip_check(ctx, table, shard, sink, prm, info, r0, qa=QA if not args.no_qual else None)
        timer_txt = ("T_core (SURVEY 8d): packed bases + quality bytes resident in HBM -> every compressed part in pinned host memory" if sink is not None
                     else "packed bases + quality bytes resident in HBM -> every compressed part gathered to rank 0 (device)")
        cb, size, e2e = (None, None, None)
        if world > 1 and args.e2e_bases > 0 and not args.no_cpu_baseline:
            # N > 1, second leg: the PRODUCT's multi-GPU host (`colord_hip --gpus N`: C++, one rank thread per GPU, RCCL) file -> archive on
            # the same GPUs.  Every rank gives its memory back first (the ranks of this process group stay alive, idle, until the leg is through).
            shard.free()
            del dna_out, qual_out
            dna_out = qual_out = None
            ctx.close()
            if qctx is not None:
                qctx.close()
            torch.cuda.empty_cache()
            try:
                e2e = multi_gpu_cli_leg(rank, world, args, k, a)
            except Exception as e:
                e2e = {"error": repr(e)[:400]}
            ctx = Context(local)
            qctx = Context(local) if qctx is not None else None
        if not args.no_cpu_baseline and world == 1:
            shard.free()                                    # the sample runs (and the command-line compressor) need the memory:
            del dna_out, qual_out, sink                     # give everything back, pools included, and start from fresh contexts
            ctx.close()
            if qctx is not None:
                qctx.close()
            torch.cuda.empty_cache()
            ctx = Context(local)
            qctx = Context(local) if qctx is not None else None
            try:                                            # (whatever happens in the side legs, the line of the timed passes is printed)
                cb, size = cpu_baseline_and_size_check(ctx, qctx, args.cpu_sample_bases, args.coverage, args.pack_symbols, k, a)
            except Exception as e:
                cb, size = {"value": None, "unit": "Gbases/s", "cores": os.cpu_count() or 1, "kind": "reference", "sample": f"failed: {e!r}"[:400]}, None
            ctx.close()
            if qctx is not None:
                qctx.close()
            torch.cuda.empty_cache()
            e2e = e2e_first                                  # (measured at the start of the process)
            ctx = Context(local)
            qctx = Context(local) if qctx is not None else None
        value = total_bases * args.steps / dt / 1e9
        line = {
            "metric": "input Gbases/s + archive size vs ref, ONT 50 Gb at 1/2/4/8 MI355X", "value": value,
            "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": value / README_DERIVED_GBASES_S,
            "vs_baseline_note": "BASELINE.md §1: 0.030 Gbases/s derived from the reference README's time and size for its `memory` preset (human ONT, "
                                "whole program, hardware not stated) — the only published figure; the measured reference on this host is `cpu_baseline`",
            "archive_vs_ref": (size or {}).get("archive_vs_ref"),
            "round_trip_checked": bool(rt and rt.get("ok")), "qual_round_trip_checked": bool(rt and rt.get("ok") and rt.get("qual_checked")), "round_trip": rt,
            "parts_digest_stable": digest_stable,
            "parts_digest": {"passes": len(digests), "entries_per_pass": len(digests[0]) if digests else 0, "first_difference": digest_diff,
                             "what": "per timed pass and chunk an xxh3-128 over (part sizes, part bytes) of `dna` and of `qual`, taken from the pinned host buffers "
                                     "between the per-step timing brackets; all passes must agree", "step_s": [round(x, 3) for x in step_s]},
            "bench_parts_sha256_equal_ref": (size or {}).get("bench_parts_sha256_equal_ref"),
            "timer": timer_txt,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"synthetic ONT {total_bases / 1e9:.2f} Gbases ({total_reads} reads, N50~20kb, 4-avg quals), genome {genome_len} bp, "
                                   f"k={k} a={a} f={PRESET['f']} ci={PRESET['ci']} cs={PRESET['cs']} c={PRESET['c']} (compress-ont default preset)"
                                   + (f"; REDUCED from {args.bases / 1e9:.1f} Gbases (a prefix of the set) so that {args.steps}+{args.warmup} passes fit --deadline-s {args.deadline_s:.0f}" if reduced else ""),
                       "chunks_per_gpu": info["chunks"], "chunk_bases": args.chunk_bases, "part_symbols": args.pack_symbols,
                       "stages": "pass 1: a1 a2 a3; pass 2a: a4 a6 a7 a5 (index); pass 2b per chunk: a5 (candidates) a8 a10 a11 a12 a14 a16"
                                 + ("" if args.no_qual else " + a13 a15 (4-avg, level 1)") + ("; parts gathered to rank 0" if world > 1 else ""),
                       "stream_bytes_per_base": round((total_dna + total_qual) / max(total_bases, 1), 4),
                       "dna_bytes": total_dna, "qual_bytes": total_qual,
                       "parallelism": f"reads sharded x{world} in file order, k-mer set + reference reads + index replicated (RCCL), one model domain per GPU" if world > 1 else "single GPU",
                       "input_generation_s": round(t_gen, 1), "rank0_sizes": info},
            "roofline": roof, "cpu_baseline": cb, "size_check": size, "ref_cut": ref_cut, "t_e2e": e2e,
        }
        if multi is not None:
            # one model domain per rank is the only place sharding changes bytes: stream bytes per base here over the committed one-GPU run's
            one = None
            for r_ in (6, 5):
                p1 = os.path.join(ROOT, "profiles", f"r0{r_}_bench_50Gbases_default.json")
                if os.path.exists(p1):
                    try:
                        j1 = json.loads(open(p1).read().strip().splitlines()[-1])
                        one = {"stream_bytes_per_base": j1["config"]["stream_bytes_per_base"], "source": os.path.basename(p1), "workload": j1["config"]["workload"][:60]}
                        break
                    except Exception:
                        pass
            here = (total_dna + total_qual) / max(total_bases, 1)
            multi["domain_loss_vs_one_rank"] = {"value": here / one["stream_bytes_per_base"] if one else None, "stream_bytes_per_base": round(here, 4), "one_rank": one,
                                                "what": "stream bytes per base of this run / of the committed 1-GPU run of the same recipe (meaningful at the same --bases only)"}
            line["multi_gpu"] = multi
        print(json.dumps(line))
    else:
        shard.free()
        if world > 1 and args.e2e_bases > 0 and not args.no_cpu_baseline:
            del dna_out, qual_out
            ctx.close()
            if qctx is not None:
                qctx.close()
            torch.cuda.empty_cache()
            multi_gpu_cli_leg(rank, world, args, k, a)
            ctx = Context(local)
            qctx = Context(local) if qctx is not None else None
    ctx.close()
    if qctx is not None:
        qctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
