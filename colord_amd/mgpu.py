"""Multi-GPU compress driver: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI), reads sharded in file
order, ONE archive written by rank 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m colord_amd.mgpu \\
        compress-ont [-p ratio|balanced|memory] [-q MODE] [-G genome.fa [-s]] [--chunk-bases X] input.fastq[.gz] output.colord

What is sharded and what is exchanged (the exchanges themselves are cl_exchange callbacks of the library's
chunked compressor, csrc/stream.hip, carried out by colord_amd.parallel.TorchExchange):
  * rank r compresses the r-th contiguous range of the file (equal shares of the bases), chunk by chunk;
  * k-mers go to the rank owning their key range, kept keys are all-gathered  -> replicated k-mer set;
  * reference reads and their index entries are all-gathered                   -> replicated reference store + index,
    so every read sees exactly the candidates the single-process run gives it (tuple streams are identical);
  * each rank is ONE MODEL DOMAIN of the DNA and quality coders (its adaptive models start fresh at its first read): the only
    place sharding changes bytes.  The archive records the first `dna` part of every domain in an extra stream `hipdomains`
    (u32 n, then per domain u64 first read, u64 first part); `colord_hip decompress` restarts its models there.  With one rank the
    archive is the reference's format byte for byte; with several, the reference's decompressor cannot decode it (it has one
    model set for the whole stream) — this build's decompressor can;
  * compressed parts are gathered to rank 0 (point-to-point sends), which adds the `header`, `meta` and `info` streams;
  * reference-genome mode (-G, compression.cpp:405-447): every rank reads the genome; rank 0 counts its k-mers and contributes the
    pseudo reads (reference reads 0 .. n-1 of the replicated store), -s stores the genome in the archive (`ref-genome`).
Host-side plumbing in Python; every byte of `dna` / `qual` comes from the HIP library, `header` from its host id coder.
"""
from __future__ import annotations
import ctypes as C
import os
import struct
import sys
import time
import numpy as np
import torch
import torch.distributed as dist

from . import _native as N, archive as AR, parallel as par
from .fastq import ReadSet, read_fastx, read_fastx_range, read_genome, genome_pseudo_reads

# arg_parse.cpp:89-408 — [source][priority]: level, ci, cs, f, c, max_rec, min_part_alt, qual_mode, sparse, g
PRESETS = {
    0: {"ratio": (3, 2, 120, 8, 10, 6, 48, 2, 0, 1), "balanced": (2, 3, 100, 9, 8, 5, 48, 2, 1, 2), "memory": (1, 4, 80, 12, 5, 3, 64, 2, 1, 1)},
    1: {"ratio": (3, 2, 120, 8, 10, 6, 48, 8, 0, 1), "balanced": (2, 3, 100, 9, 8, 5, 48, 8, 1, 2), "memory": (1, 4, 80, 12, 5, 3, 64, 8, 1, 1)},
    2: {"ratio": (3, 2, 150, 20, 12, 6, 48, 1, 0, 1), "balanced": (2, 3, 120, 30, 10, 5, 48, 1, 1, 6), "memory": (2, 3, 100, 40, 8, 5, 48, 1, 1, 3)},
}
QUAL_NAMES = ["org", "5-avg", "4-avg", "2-avg", "5-fix", "4-fix", "2-fix", "avg", "none"]
QUAL_DEFAULTS = {0: ((), ()), 1: ((7, 14, 26, 93), ()), 2: ((7, 14, 26), ()), 3: ((7,), ()), 4: ((7, 14, 26, 93), (3, 10, 18, 35, 93)),
                 5: ((7, 14, 26), (3, 10, 18, 35)), 6: ((7,), (1, 13)), 7: ((), ()), 8: ((), (0,))}
READS_PACK = 2 << 21


def kmer_anchor_len(est_bases: float):
    for lim, k, a in ((1e9, 20, 16), (4e9, 21, 18), (16e9, 23, 21), (48e9, 24, 22), (128e9, 25, 22)):
        if est_bases < lim:
            return k, a
    return 26, 23


def _packs(lens: np.ndarray, size: int = READS_PACK) -> np.ndarray:
    """Reader packs (in_reads.cpp:62-77): a pack closes once sum(len + 1) >= 4 Mi."""
    acc = np.cumsum(lens.astype(np.int64) + 1)
    out, base = [0], 0
    while True:
        i = int(np.searchsorted(acc, base + size, side="left"))
        if i >= len(acc):
            break
        out.append(i + 1)
        base = int(acc[i])
    if out[-1] != len(lens):
        out.append(len(lens))
    return np.asarray(out, dtype=np.uint32)


def encode_headers(headers, plus_eq, header_mode: int):
    """The `header` stream (CEntrComprHeaders, entr_header.cpp:23-45): packs of >= 4 Mi id bytes -> [(n ids, payload)]."""
    lib = N.load()
    h = N._P()
    if lib.cl_id_coder_create(header_mode, C.byref(h)) != 0:
        raise RuntimeError("cl_id_coder_create")
    parts, i, n = [], 0, len(headers)
    try:
        while i < n:
            j, acc = i, 0
            while j < n:
                acc += len(headers[j]); j += 1
                if acc >= READS_PACK:
                    break
            ids = np.frombuffer(b"".join(headers[i:j]), np.uint8)
            off = np.concatenate([[0], np.cumsum([len(x) for x in headers[i:j]])]).astype(np.uint64)
            plus = np.asarray(plus_eq[i:j], np.uint8)
            out = np.empty(2 * len(ids) + 64, np.uint8)
            got = C.c_uint64(0)
            if lib.cl_id_encode_part(h, ids.ctypes.data if len(ids) else None, off.ctypes.data, plus.ctypes.data, j - i, out.ctypes.data, len(out), C.byref(got)) != 0:
                raise RuntimeError("header stream: " + lib.cl_id_coder_error(h).decode())
            parts.append((j - i, out[:got.value].tobytes()))
            i = j
    finally:
        lib.cl_id_coder_free(h)
    return parts


def compress_readset(rs: ReadSet, out_path: str | None, source: int = 0, priority: str = "memory", qual_mode: int | None = None, header_mode: int = 0,
                     k: int = 0, a: int = 0, chunk_bases: float = 1.0e9, est_bases: float | None = None, command: str = "", file_bytes: int = 0, device: int | None = None,
                     presharded: bool = False, genome: str | None = None, store_genome: bool = False):
    """Compresses `rs`.  presharded = False: every rank holds the same ReadSet and takes its share (tests); True: `rs` is this
    rank's OWN contiguous share of the file (read_fastx_range: no rank holds the whole input), totals and headers travel through
    the process group.  Rank 0 writes out_path and returns a dict of sizes."""
    from .device import Context
    world, rank = par.world(), par.rank()
    local = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
    if dist.is_initialized() and dist.get_backend() != "nccl":
        local %= torch.cuda.device_count()
    level, ci, cs, f, c, max_rec, min_alt, qm, sparse, g = PRESETS[source][priority]
    qm = qm if qual_mode is None else qual_mode
    lens_all = np.diff(rs.offsets).astype(np.int64)
    tot_reads, tot_bases = rs.n_reads, int(lens_all.sum())
    if presharded and world > 1:
        tot_reads, tot_bases = par.all_reduce_sum_ints(rs.n_reads, int(lens_all.sum()))
    if not k:
        k, a = kmer_anchor_len(est_bases if est_bases is not None else float(tot_bases))
    prm = dict(k=k, f=f, ci=ci, cs=cs, c=c, anchor_len=a, min_part_alt=min_alt, max_rec=max_rec, min_anchors=1, level=level, source=source, sparse=sparse,
               sparse_g=float(g), sparse_exponent=1.0, cost_mult=1.0, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0)
    with_qual = rs.is_fastq and rs.quals is not None
    qd = QUAL_DEFAULTS[qm]
    qargs = (qm, source, level, qd[0], qd[1]) if with_qual else None
    # this rank's contiguous share of the file
    acc = np.cumsum(lens_all)
    if presharded:
        r0, r1 = 0, rs.n_reads
    else:
        r0 = int(np.searchsorted(acc, acc[-1] * rank / world, side="left")) if rank else 0
        r1 = int(np.searchsorted(acc, acc[-1] * (rank + 1) / world, side="left")) if rank + 1 < world else rs.n_reads
    lens = lens_all[r0:r1]
    packs = _packs(lens)
    cacc = np.concatenate([[0], np.cumsum(lens)])
    cuts, target = [0], chunk_bases
    for p in range(1, len(packs)):
        if cacc[packs[p]] >= target or p == len(packs) - 1:
            cuts.append(p)
            target = cacc[packs[p]] + chunk_bases
    ctx = Context(local)
    qctx = Context(local) if level == 1 and with_qual else None
    exchange = par.TorchExchange(ctx.device) if world > 1 else None
    chunks = []
    for ca, cb in zip(cuts[:-1], cuts[1:]):
        a_, b_ = r0 + int(packs[ca]), r0 + int(packs[cb])
        o0, o1 = int(rs.offsets[a_]), int(rs.offsets[b_])
        off = torch.from_numpy((rs.offsets[a_:b_ + 1] - o0).astype(np.int64)).to(ctx.device)
        arena = ctx.pack_reads(torch.from_numpy(rs.bases[o0:o1]), off)
        q = torch.from_numpy(rs.quals[o0:o1]).to(ctx.device) if with_qual else None
        chunks.append((arena, (packs[ca:cb + 1] - packs[ca]).astype(np.uint32), q, off))
    cmp_ = ctx.compressor(prm, qargs, qctx, exchange, expected_bases=int(lens.sum()))
    g_codes = g_off = None
    genome_read_len, genome_overlap, n_pseudo = 0, (k - 1) * 10, 0                               # compression.cpp:407,447
    try:
        if genome:
            g_codes, g_off = read_genome(genome)
            gr = ctx.pack_reads(torch.from_numpy(g_codes), torch.from_numpy(g_off.astype(np.int64)))
            cmp_.genome_add(gr)                               # (rank 0 scans it; the others note its size)
            gr.free()
        for ch in chunks:
            cmp_.count_add(ch[0])
        cmp_.count_finish()
        if genome:
            genome_read_len = 20 * cmp_.info()["mean_read_len"]
            if genome_read_len >= 1 << 32:
                raise ValueError("reference genome: pseudo reads too long")
            p_codes, p_off = genome_pseudo_reads(g_codes, g_off, genome_read_len, genome_overlap)
            n_pseudo = len(p_off) - 1
            pr = ctx.pack_reads(torch.from_numpy(p_codes), torch.from_numpy(p_off))
            cmp_.pseudo_reads(pr)
            pr.free()
        for ch in chunks:
            cmp_.refs_add(ch[0])
        cmp_.refs_finish()
        dna, qual, dsz, qsz, counts = [], [], [], [], []
        for arena, pb, q, off in chunks:                     # resident chunks: the encode lanes work ahead of the coders
            cmp_.prepare(arena, pb, pb, q, off)
        for arena, pb, q, off in chunks:
            d, ds, qq, qs, _ = cmp_.encode(arena, pb, pb, q, off)
            dna.append(d.clone()); dsz += [int(x) for x in ds]; counts += [int(x) for x in np.diff(pb)]
            if with_qual:
                qual.append(qq.clone()); qsz += [int(x) for x in qs]
        info = cmp_.info()
    except Exception:
        if exchange is not None and exchange.err is not None:
            raise exchange.err
        raise
    finally:
        cmp_.free()
        for ch in chunks:
            ch[0].free()
    empty = torch.empty(0, dtype=torch.uint8, device=ctx.device)
    dna = torch.cat(dna) if dna else empty
    qual = torch.cat(qual) if qual else empty
    # SURVEY §8e "collective for results": payloads and part tables of every rank to rank 0
    tab = torch.tensor([len(dsz)] + dsz + counts + (qsz if with_qual else []), dtype=torch.int64, device=ctx.device)
    g_dna = par.gather_to_root(dna)
    g_qual = par.gather_to_root(qual) if with_qual else None
    g_tab = par.gather_to_root(tab.view(torch.uint8))
    all_headers, all_plus = rs.headers, rs.plus_eq
    if presharded and world > 1:
        # the `header` stream is coded by rank 0 (one id coder over the whole file): the id bytes of every share go there
        blob = b"".join(rs.headers)
        hl = np.fromiter((len(x) for x in rs.headers), dtype=np.int64, count=len(rs.headers))
        pe = np.asarray(rs.plus_eq, dtype=np.uint8)
        pack = np.concatenate([np.array([len(hl)], np.int64).view(np.uint8), hl.view(np.uint8), pe, np.frombuffer(blob, np.uint8)])
        g_hdr = par.gather_to_root(torch.from_numpy(pack.copy()).to(ctx.device))
        if rank == 0:
            all_headers, all_plus = [], []
            for t_ in g_hdr:
                b_ = t_.cpu().numpy()
                n_ = int(b_[:8].view(np.int64)[0]); hl_ = b_[8:8 + 8 * n_].view(np.int64); pe_ = b_[8 + 8 * n_:8 + 9 * n_]; raw_ = b_[8 + 9 * n_:].tobytes()
                o_ = 0
                for L_, e_ in zip(hl_, pe_):
                    all_headers.append(raw_[o_:o_ + int(L_)]); all_plus.append(bool(e_)); o_ += int(L_)
    res = None
    if rank == 0:
        d_st, q_st = AR.Stream("dna"), AR.Stream("qual")
        domains, first_read = [], 0
        reads_per_rank = []
        for r in range(world):
            t = g_tab[r].view(torch.int64).cpu().numpy()
            np_ = int(t[0]); ds = t[1:1 + np_]; cn = t[1 + np_:1 + 2 * np_]; qs = t[1 + 2 * np_:1 + 3 * np_] if with_qual else None
            domains.append((first_read, len(d_st.parts)))
            raw = g_dna[r].cpu().numpy().tobytes(); o = 0
            for s_, n_ in zip(ds, cn):
                d_st.parts.append((int(n_), raw[o:o + int(s_)])); o += int(s_)
            if with_qual:
                raw = g_qual[r].cpu().numpy().tobytes(); o = 0
                for s_ in qs:
                    q_st.parts.append((0, raw[o:o + int(s_)])); o += int(s_)
            first_read += int(cn.sum()); reads_per_rank.append(int(cn.sum()))
        assert first_read == tot_reads
        h_st = AR.Stream("header"); h_st.parts = encode_headers(all_headers, all_plus, header_mode)
        n = tot_reads
        tot_ref = n + n_pseudo
        if sparse:
            accd = np.zeros(n + n_pseudo, np.uint8)
            N.load().cl_ref_accept(n, n_pseudo, info["sparse_range"], 1.0, accd.ctypes.data)
            tot_ref = int(accd.sum())
        meta = struct.pack("<IIiBQ", tot_ref, c, level, source, n * info["mean_read_len"])      # compression.cpp:704-779
        if with_qual:
            meta += bytes([qm])
            if qm in (8, 4, 5, 6):
                meta += b"".join(struct.pack("<I", v) for v in qd[1])
        meta += bytes([header_mode, 1 if sparse else 0])
        if sparse:
            meta += struct.pack("<Id", info["sparse_range"], 1.0)
        g_st = None
        if genome:                                                                               # compression.cpp:764-777
            n_seqs = len(g_off) - 1
            g_off64 = np.ascontiguousarray(g_off, dtype=np.uint64)
            meta += bytes([1, 1 if store_genome else 0]) + struct.pack("<III", genome_read_len, genome_overlap, n_pseudo)
            if store_genome:                                                                     # reference_genome.cpp:325-370
                cap, got = g_codes.size // 3 + 4096, C.c_uint64(0)
                while True:
                    buf = np.empty(cap, np.uint8)
                    st_ = N.load().cl_genome_encode(g_codes.ctypes.data, g_off64.ctypes.data, n_seqs, buf.ctypes.data, cap, C.byref(got))
                    if st_ != N.CL_E_CAPACITY:
                        break
                    cap = int(got.value)
                if st_ != 0:
                    raise RuntimeError("cannot code the reference genome")
                g_st = AR.Stream("ref-genome"); g_st.parts = [(n_seqs, buf[:got.value].tobytes())]
            else:
                md = np.zeros(16, np.uint8)
                if N.load().cl_genome_md5(g_codes.ctypes.data, g_off64.ctypes.data, n_seqs, md.ctypes.data) != 0:
                    raise RuntimeError("cannot checksum the reference genome")
                meta += md.tobytes()
        else:
            meta += b"\0"
        m_st = AR.Stream("meta"); m_st.parts = [(0, meta)]
        cmd = command.encode()
        inf = struct.pack("<IIIQQIQI", 1, 2, 1, file_bytes, tot_bases, n, int(time.time()), len(cmd)) + cmd      # utils.cpp:326-342
        i_st = AR.Stream("info"); i_st.parts = [(0, inf)]
        streams = [m_st] + ([g_st] if g_st is not None else []) + [h_st, d_st] + ([q_st] if with_qual else [])
        if world > 1:
            dom = AR.Stream("hipdomains")
            dom.parts = [(0, struct.pack("<I", world) + b"".join(struct.pack("<QQ", fr, fp) for fr, fp in domains))]
            streams.append(dom)
        streams.append(i_st)
        if out_path:
            AR.write_archive(out_path, streams)
        res = dict(n_reads=n, n_bases=tot_bases, k=k, a=a, dna_bytes=sum(len(p) for _, p in d_st.parts), qual_bytes=sum(len(p) for _, p in q_st.parts),
                   header_bytes=sum(len(p) for _, p in h_st.parts), dna_parts=len(d_st.parts), refs=info["n_refs_total"], domains=domains, reads_per_rank=reads_per_rank, n_pseudo=n_pseudo,
                   exchanged_bytes_rank0=exchange.bytes_moved if exchange else 0)
    ctx.close()
    if qctx is not None:
        qctx.close()
    return res


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] not in ("compress-ont", "compress-pbraw", "compress-pbhifi"):
        print(__doc__, file=sys.stderr)
        return 1
    source = {"compress-ont": 0, "compress-pbraw": 1, "compress-pbhifi": 2}[argv[0]]
    prio, qm, hm, chunk, k, a, pos, genome, store = "memory", None, 0, 1.0e9, 0, 0, [], None, False
    i = 1
    while i < len(argv):
        x = argv[i]
        if x in ("-p", "--priority"):
            prio = argv[i + 1]; i += 2
        elif x in ("-q", "--qual"):
            if argv[i + 1] not in QUAL_NAMES:
                print(f"unknown quality mode '{argv[i + 1]}' ({', '.join(QUAL_NAMES)})", file=sys.stderr)
                return 1
            qm = QUAL_NAMES.index(argv[i + 1]); i += 2
        elif x in ("-i", "--identifier"):
            if argv[i + 1] not in ("org", "main", "none"):
                print(f"unknown identifier mode '{argv[i + 1]}' (org, main, none)", file=sys.stderr)
                return 1
            hm = ["org", "main", "none"].index(argv[i + 1]); i += 2
        elif x in ("-k", "--kmer-len"):
            k = int(argv[i + 1]); i += 2
        elif x in ("-a", "--anchor-len"):
            a = int(argv[i + 1]); i += 2
        elif x in ("-G", "--reference-genome"):
            genome = argv[i + 1]; i += 2
        elif x in ("-s", "--store-reference"):
            store = True; i += 1
        elif x == "--chunk-bases":
            chunk = float(argv[i + 1]); i += 2
        else:
            pos.append(x); i += 1
    if len(pos) != 2 or bool(k) != bool(a):
        print("expected input and output paths (and -k together with -a)", file=sys.stderr)
        return 1
    # (options are checked before the process group exists: a bad value must not leave the other ranks waiting in a collective)
    if prio not in PRESETS[source]:
        print(f"unknown priority '{prio}' (ratio, balanced, memory)", file=sys.stderr)
        return 1
    if k and not (15 <= k <= 28 and 10 <= a <= k):
        print("k-mer length 15..28, anchor length 10..k", file=sys.stderr)
        return 1
    if not os.path.isfile(pos[0]):
        print(f"cannot open {pos[0]}", file=sys.stderr)
        return 1
    if genome is not None and not os.path.isfile(genome):
        print(f"cannot open {genome}", file=sys.stderr)
        return 1
    if store and genome is None:
        print("-s needs -G", file=sys.stderr)
        return 1
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("COLORD_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.cuda.set_device(local % torch.cuda.device_count())
            dist.init_process_group(backend)
    t0 = time.time()
    size = os.path.getsize(pos[0])
    gz = open(pos[0], "rb").read(2) == b"\x1f\x8b"
    # every rank reads ITS byte range of a plain 4-line FASTQ (no rank parses or holds the whole file); other inputs (gzip, FASTA,
    # multi-line records) are read whole by every rank, which keeps its share
    rank_ = int(os.environ.get("RANK", "0"))
    shard, shard_err = None, None
    if world > 1:
        try:                                              # (a rank whose share is malformed votes 0 below instead of leaving the others in the collective)
            shard = read_fastx_range(pos[0], rank_, world)
        except ValueError as e:
            shard_err = e
    ok = torch.tensor([1 if (shard is not None or world == 1) else 0], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if (world > 1 and dist.get_backend() == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # (all ranks take the same path)
    if shard_err is not None and world > 1 and int(ok.item()) == 0:
        print(f"colord_amd.mgpu: rank {rank_}: {shard_err}", file=sys.stderr)
    if world > 1 and int(ok.item()) == 1:
        rs = shard[0]
        est = size * 0.49
        res = compress_readset(rs, pos[1], source, prio, qm, hm, k, a, chunk, est, "colord_amd.mgpu " + " ".join(argv), size, presharded=True, genome=genome, store_genome=store)
    else:
        rs = read_fastx(pos[0])
        est = size * ((2.08 if rs.is_fastq else 3.98) if gz else (0.49 if rs.is_fastq else 0.98))      # compression.cpp:52-61
        res = compress_readset(rs, pos[1], source, prio, qm, hm, k, a, chunk, est, "colord_amd.mgpu " + " ".join(argv), size, genome=genome, store_genome=store)
    if res is not None:
        print(f"colord_amd.mgpu: {res['n_reads']} reads, {res['n_bases']} bases on {world} GPU(s), k={res['k']} a={res['a']}; dna {res['dna_bytes']} B ({res['dna_parts']} parts), "
              f"qual {res['qual_bytes']} B, header {res['header_bytes']} B; {res['refs']} reference reads; {time.time() - t0:.2f} s", file=sys.stderr)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
