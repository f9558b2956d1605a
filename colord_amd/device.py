"""Thin object layer over the C ABI for tests and bench.py: torch tensors own the caller-side device
memory; everything computed comes from the HIP library (no CPU fallback anywhere)."""
from __future__ import annotations
import ctypes as C
import numpy as np
import torch
from . import _native as N


def _check(ctx, st):
    if ctx is not None and getattr(ctx, "timing", False):
        for n, (ms, k, b, cells) in ctx.kernel_times().items():
            a = ctx.acc.setdefault(n, [0.0, 0, 0.0, 0.0])
            a[0] += ms
            a[1] += k
            a[2] += b
            a[3] += cells
    if st != N.CL_OK:
        msg = N.load().cl_last_error(ctx.h).decode() if ctx is not None and ctx.h else ""
        raise N.ColordHipError(st, msg)


def _view(ptr, n, dtype, device):
    """Copy n elements of a library-owned device array into a new torch tensor."""
    out = torch.empty(int(n), dtype=dtype, device=device)
    if n:
        torch.cuda.synchronize(device)
        rc = _hip().hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(int(n) * out.element_size()), 3)  # D2D
        if rc != 0:
            raise N.ColordHipError(N.CL_E_HIP, f"hipMemcpy failed ({rc})")
    return out


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _HIP.hipMemcpy.restype = C.c_int
    return _HIP


class _OrderedLib:
    """The C ABI as this binding calls it: every entry point first waits for torch's current stream on the device.  The library works
    on streams of its own (non-blocking: nothing orders them after torch's), so a tensor torch is still computing — offsets rebased by a
    subtraction, a slice made contiguous — could be read by the library before it was written.  That was the one unexplained failure of
    round 4 (`cl_reads_pack: offsets not monotone` once in seven runs of test_chunked_equals_one_call_200_mbases; reproduced at will under
    COLORD_HIP_SYNC_DEBUG, which shifts the timing): a race of the HARNESS, not of the library's pool.  Waiting on an idle stream costs
    microseconds; a host that embeds the library orders its own streams."""
    _HOST_ONLY = ("cl_last_error", "cl_ctx_kernel_times", "cl_ctx_last_kernel_ms", "cl_ctx_set_timing", "cl_ref_accept")

    def __init__(self, lib, device):
        self._lib, self._device, self._cache = lib, device, {}

    def __getattr__(self, name):
        f = self._cache.get(name)
        if f is None:
            raw = getattr(self._lib, name)
            if not name.startswith("cl_") or name in self._HOST_ONLY or name.endswith("_free") or name.endswith("_destroy"):
                f = raw
            else:
                dev = self._device

                def f(*a, _raw=raw, _dev=dev):
                    torch.cuda.current_stream(_dev).synchronize()
                    return _raw(*a)
            self._cache[name] = f
        return f


class Context:
    def __init__(self, device: int = 0, timing: bool = False):
        self.lib = N.load()
        if not torch.cuda.is_available():
            raise N.ColordHipError(N.CL_E_HIP, "no GPU visible: the HIP path cannot run (no CPU fallback)")
        self.h = N._P()
        st = self.lib.cl_ctx_create(device, C.byref(self.h))
        if st != N.CL_OK:
            raise N.ColordHipError(st, "cl_ctx_create failed")
        self.device = torch.device("cuda", device)
        self.lib = _OrderedLib(self.lib, self.device)
        self.timing = timing
        self.acc = {}                  # {kernel: [ms, launches]} accumulated over API calls while timing is on
        if timing:
            self.lib.cl_ctx_set_timing(self.h, 1)

    def close(self):
        if self.h:
            self.lib.cl_ctx_destroy(self.h)
            self.h = None

    def kernel_ms(self, name: str):
        ms, n = C.c_double(0), C.c_uint32(0)
        self.lib.cl_ctx_last_kernel_ms(self.h, name.encode(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def kernel_times(self) -> dict:
        """{kernel: (ms, launches)} of the last API call (timing must be on)."""
        buf = C.create_string_buffer(1 << 16)
        need = C.c_uint64(0)
        if self.lib.cl_ctx_kernel_times(self.h, buf, len(buf), C.byref(need)) != N.CL_OK:
            return {}
        out = {}
        for line in buf.value.decode().splitlines():
            f = line.split("\t")
            out[f[0]] = (float(f[1]), int(f[2]), float(f[3]), float(f[4]) if len(f) > 4 else 0.0)
        return out

    # ---- arena ----
    def pack_reads(self, codes: torch.Tensor, offsets: torch.Tensor, ascii: bool = False) -> "Reads":
        assert codes.dtype == torch.uint8 and offsets.dtype in (torch.int64, torch.uint64)
        codes = codes.to(self.device).contiguous()
        offsets = offsets.to(self.device).contiguous()
        h = N._P()
        _check(self, self.lib.cl_reads_pack(self.h, codes.data_ptr(), offsets.data_ptr(), offsets.numel() - 1, int(ascii), C.byref(h)))
        return Reads(self, h)

    def select_reads(self, reads: "Reads", keep: torch.Tensor) -> "Reads":
        """CReferenceReads: the arena of the reads with keep[i] != 0 (reference id = rank among the kept reads)."""
        assert keep.dtype == torch.uint8 and keep.numel() == reads.n_reads
        h = N._P()
        _check(self, self.lib.cl_reads_select(self.h, reads.h, keep.contiguous().data_ptr(), C.byref(h)))
        return Reads(self, h)

    def reads_from_arena(self, packed: torch.Tensor, inv: torch.Tensor, lens: torch.Tensor) -> "Reads":
        """Arena from already packed words (word-aligned reads back to back), e.g. gathered from all ranks."""
        h = N._P()
        n = lens.numel()
        _check(self, self.lib.cl_reads_from_arena(self.h, packed.contiguous().data_ptr() if n else None, inv.contiguous().data_ptr() if n else None,
                                                  lens.contiguous().data_ptr() if n else None, n, C.byref(h)))
        return Reads(self, h)

    def pack_readset(self, rs) -> "Reads":
        return self.pack_reads(torch.from_numpy(rs.bases), torch.from_numpy(rs.offsets))

    # ---- a1 ----
    def kmer_scan(self, reads: "Reads", k: int, f: int, cap: int | None = None) -> torch.Tensor:
        if cap is None:
            cap = int(reads.total_bases // max(f, 1) * 1.3) + 4096 if f > 1 else int(reads.total_bases) + 64
        while True:
            out = torch.empty(cap, dtype=torch.int64, device=self.device)
            n = C.c_uint64(0)
            st = self.lib.cl_kmer_scan(self.h, reads.h, k, f, out.data_ptr(), cap, C.byref(n))
            if st == N.CL_E_CAPACITY:
                cap = int(n.value)
                continue
            _check(self, st)
            return out[:n.value]

    # ---- a2 + a3 ----
    def count_filter(self, kmers: torch.Tensor, k: int, ci: int, cs: int):
        kmers = kmers.contiguous()
        h = N._P()
        st = N.KmerStats()
        _check(self, self.lib.cl_kmer_count_filter(self.h, kmers.data_ptr(), kmers.numel(), k, ci, cs, C.byref(h), C.byref(st)))
        return KmerSet(self, h), st

    def kmer_set_from_keys(self, keys: torch.Tensor, counts: torch.Tensor, k: int) -> "KmerSet":
        """Replicated set from gathered per-rank partitions: sort by key on device, then build the table."""
        keys = keys.contiguous().clone()
        idx = torch.arange(keys.numel(), dtype=torch.int32, device=self.device)
        self.sort_u64(keys, idx, 0, 2 * k)
        counts = counts.contiguous()[idx.long()].contiguous()
        h = N._P()
        _check(self, self.lib.cl_kmer_set_create(self.h, keys.data_ptr(), counts.data_ptr(), keys.numel(), k, C.byref(h)))
        return KmerSet(self, h)

    # ---- a4 ----
    def accepted_kmers(self, kset: "KmerSet", reads: "Reads", k: int, f: int) -> "KmerLists":
        h = N._P()
        _check(self, self.lib.cl_accepted_kmers(self.h, kset.h, reads.h, k, f, C.byref(h)))
        return KmerLists(self, h)

    # ---- a6 ----
    def ref_accept(self, n_reads: int, n_pseudo: int, rng: int, exponent: float) -> np.ndarray:
        out = np.zeros(n_reads + n_pseudo, np.uint8)
        st = self.lib.cl_ref_accept(n_reads, n_pseudo, rng, exponent, out.ctypes.data)
        _check(self, st)
        return out

    # ---- a5 ----
    def index_build(self, kset, lists, accept: torch.Tensor, n_pseudo: int, max_kmer_count: int) -> "Index":
        accept = accept.to(self.device).contiguous()
        assert accept.dtype == torch.uint8 and accept.numel() == lists.n_reads
        h = N._P()
        _check(self, self.lib.cl_index_build(self.h, kset.h, lists.h, accept.data_ptr(), n_pseudo, max_kmer_count, C.byref(h)))
        return Index(self, h)

    def index_entries(self, lists, accept: torch.Tensor, ref_base: int):
        accept = accept.to(self.device).contiguous()
        bounds = torch.empty(lists.n_reads + 1, dtype=torch.int32, device=self.device)
        n, nacc = C.c_uint64(0), C.c_uint32(0)
        st = self.lib.cl_index_entries_of(self.h, lists.h, accept.data_ptr(), ref_base, None, None, 0, C.byref(n), bounds.data_ptr(), C.byref(nacc))
        if st not in (N.CL_OK, N.CL_E_CAPACITY):
            _check(self, st)
        ids = torch.empty(max(1, n.value), dtype=torch.int32, device=self.device)
        refs = torch.empty(max(1, n.value), dtype=torch.int32, device=self.device)
        if n.value:
            _check(self, self.lib.cl_index_entries_of(self.h, lists.h, accept.data_ptr(), ref_base, ids.data_ptr(), refs.data_ptr(), n.value, C.byref(n), None, None))
        return ids[:n.value], refs[:n.value], bounds, nacc.value

    def index_build_pairs(self, kset, ids, refs, bounds, n_refs_total: int, n_pseudo: int, max_kmer_count: int) -> "Index":
        ids, refs = ids.contiguous().clone(), refs.contiguous().clone()
        h = N._P()
        _check(self, self.lib.cl_index_build_pairs(self.h, kset.h, ids.data_ptr(), refs.data_ptr(), ids.numel(), bounds.data_ptr(),
                                                   bounds.numel() - 1, n_refs_total, n_pseudo, max_kmer_count, C.byref(h)))
        return Index(self, h)

    def candidates(self, index, lists, c: int):
        n = lists.n_reads
        refs = torch.empty((n, c), dtype=torch.int32, device=self.device)
        votes = torch.empty((n, c), dtype=torch.int32, device=self.device)
        cnt = torch.empty(n, dtype=torch.int32, device=self.device)
        _check(self, self.lib.cl_candidates(self.h, index.h, lists.h, c, refs.data_ptr(), votes.data_ptr(), cnt.data_ptr()))
        return refs, votes, cnt

    def candidates_common(self, index, lists, c: int, refs, cnt):
        n = lists.n_reads
        off = torch.empty(n * c + 1, dtype=torch.int64, device=self.device)
        need = C.c_uint64(0)
        st = self.lib.cl_candidates_common(self.h, index.h, lists.h, c, refs.data_ptr(), cnt.data_ptr(), off.data_ptr(), None, 0, C.byref(need))
        if st not in (N.CL_OK, N.CL_E_CAPACITY):
            _check(self, st)
        common = torch.empty(max(1, need.value), dtype=torch.int64, device=self.device)
        _check(self, self.lib.cl_candidates_common(self.h, index.h, lists.h, c, refs.data_ptr(), cnt.data_ptr(), off.data_ptr(),
                                                   common.data_ptr(), need.value, C.byref(need)))
        return off, common[:need.value]

    # ---- a13 + a15 ----
    def qual_coder(self, mode: int, source: int, level: int, fwd=(), rev=()) -> "QualCoder":
        prm = N.QualParams()
        prm.mode, prm.source, prm.level = mode, source, level
        prm.n_fwd, prm.n_rev = len(fwd), len(rev)
        for i, v in enumerate(fwd):
            prm.fwd[i] = v
        for i, v in enumerate(rev):
            prm.rev[i] = v
        h = N._P()
        _check(self, self.lib.cl_qual_coder_create(self.h, C.byref(prm), C.byref(h)))
        return QualCoder(self, h)

    # ---- a8 ----
    def anchor_candidates(self, reads: "Reads", refs: "Reads", cand_refs: torch.Tensor, cand_n: torch.Tensor, anchor_len: int,
                          frac_always=0.9, frac_min=0.5, max_matches_mult=10.0, min_anchors=1, hifi=None) -> "Anchors":
        """hifi = (kmer_len, modulo, common_off, common) from candidates_common: HiFi k-mer anchors first (a9)."""
        c = cand_refs.shape[1]
        h = N._P()
        if hifi is not None:
            k, f, coff, common = hifi
            _check(self, self.lib.cl_anchor_candidates_hifi(self.h, reads.h, refs.h, cand_refs.contiguous().data_ptr(), cand_n.contiguous().data_ptr(), c, anchor_len,
                                                            frac_always, frac_min, max_matches_mult, min_anchors, k, f, coff.contiguous().data_ptr(),
                                                            common.contiguous().data_ptr() if common.numel() else None, C.byref(h)))
            return Anchors(self, h, reads.n_reads, c)
        _check(self, self.lib.cl_anchor_candidates(self.h, reads.h, refs.h, cand_refs.contiguous().data_ptr(), cand_n.contiguous().data_ptr(), c, anchor_len,
                                                   frac_always, frac_min, max_matches_mult, min_anchors, C.byref(h)))
        return Anchors(self, h, reads.n_reads, c)

    # ---- a12 (plain forms) ----
    def encode_plain(self, reads: "Reads"):
        n = reads.n_reads
        cap = int(reads.total_bases) + n
        es = torch.empty(max(cap, 1), dtype=torch.uint8, device=self.device)
        off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        nt = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        need = C.c_uint64(0)
        _check(self, self.lib.cl_encode_plain(self.h, reads.h, es.data_ptr(), cap, off.data_ptr(), nt.data_ptr(), C.byref(need)))
        return es[:need.value], off, nt[:n]

    # ---- a10-a12 ----
    def encode_reads(self, reads: "Reads", refs: "Reads", anchors: "Anchors", anchor_len: int, min_part_alt: int, max_rec: int, cost_mult: float,
                     pack_bounds=None):
        """CEncoder::Encode over the arena: returns (tuple bytes, byte offsets [n+1], tuple counts [n])."""
        import numpy as np
        n = reads.n_reads
        pb = np.ascontiguousarray(np.asarray([0, n] if pack_bounds is None else pack_bounds, dtype=np.uint32))
        off = torch.empty(n + 1, dtype=torch.int64, device=self.device)
        nt = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        need = C.c_uint64(0)
        cap = int(reads.total_bases) + 16 * n + 4096       # a stream never exceeds the plain form (1 B per base + header) by more than its headers
        for _ in range(2):
            es = torch.empty(max(cap, 1), dtype=torch.uint8, device=self.device)
            st = self.lib.cl_encode_reads(self.h, reads.h, refs.h, anchors.h, anchors.c, anchor_len, min_part_alt, max_rec, cost_mult,
                                          pb.ctypes.data, len(pb) - 1, es.data_ptr(), cap, off.data_ptr(), nt.data_ptr(), C.byref(need))
            if st == N.CL_E_CAPACITY and need.value > cap:
                cap = need.value
                continue
            _check(self, st)
            break
        return es[:need.value], off, nt[:n]

    def es_flags(self, reads: "Reads", es: torch.Tensor, es_off: torch.Tensor, base_off: torch.Tensor) -> torch.Tensor:
        """Per-base classes for the quality coder at levels 2-3 (quality_coder_impl.cpp:25-75) from the tuple streams."""
        flags = torch.empty(max(int(reads.total_bases), 1), dtype=torch.uint8, device=self.device)
        _check(self, self.lib.cl_es_flags(self.h, reads.h, es.contiguous().data_ptr() if es.numel() else None, es_off.contiguous().data_ptr(),
                                          base_off.contiguous().data_ptr(), flags.data_ptr()))
        return flags[:int(reads.total_bases)]

    def estimator_logs(self, count: torch.Tensor, total: torch.Tensor) -> torch.Tensor:
        """The decision logarithm -log2(count * (1/total)) as the encoder's kernels evaluate it (calc_logs, utils.h:800-810)."""
        count, total = count.contiguous(), total.contiguous()
        assert count.dtype == torch.int32 and total.dtype == torch.int32 and count.numel() == total.numel()
        out = torch.empty(count.numel(), dtype=torch.float64, device=self.device)
        _check(self, self.lib.cl_estimator_logs(self.h, count.data_ptr(), total.data_ptr(), count.numel(), out.data_ptr()))
        return out

    # ---- the whole data path of one shard in one native call ----
    def compress_shard(self, reads: "Reads", params: dict, part_bounds, pack_bounds, dna: "DnaCoder", qual: "QualCoder | None" = None,
                       quals: torch.Tensor | None = None, base_off: torch.Tensor | None = None):
        """cl_compress_shard: returns (dna payload, dna part sizes, qual payload or None, qual part sizes, info dict)."""
        import numpy as np
        P = N.CompressParams()
        for k_, v in params.items():
            setattr(P, k_, v)
        pb = np.ascontiguousarray(np.asarray(part_bounds, dtype=np.uint32)); kb = np.ascontiguousarray(np.asarray(pack_bounds, dtype=np.uint32))
        npart = len(pb) - 1
        dcap = int(reads.total_bases) + 64 * npart + 4096
        dout = torch.empty(dcap, dtype=torch.uint8, device=self.device)
        dsz = np.zeros(max(npart, 1), np.uint64)
        qout, qsz, qcap = None, np.zeros(max(npart, 1), np.uint64), 0
        if qual is not None:
            qcap = int(int(reads.total_bases) * 1.35) + 64 * npart + 4096
            qout = torch.empty(qcap, dtype=torch.uint8, device=self.device)
        info = N.CompressInfo()
        _check(self, self.lib.cl_compress_shard(self.h, C.byref(P), reads.h, quals.data_ptr() if qual is not None else None,
                                                base_off.contiguous().data_ptr() if qual is not None else None,
                                                pb.ctypes.data, npart, kb.ctypes.data, len(kb) - 1, dna.h, qual.h if qual is not None else None,
                                                dout.data_ptr(), dcap, dsz.ctypes.data, qout.data_ptr() if qout is not None else None, qcap, qsz.ctypes.data, C.byref(info)))
        if qual is not None and qual.ctx is not self:
            _check(qual.ctx, N.CL_OK)                       # collect the kernel times of the concurrent quality stream
        inf = {n_: getattr(info, n_) for n_, _ in N.CompressInfo._fields_ if n_ != "pad"}
        return dout[:inf["dna_bytes"]], dsz[:npart], (qout[:inf["qual_bytes"]] if qout is not None else None), qsz[:npart], inf

    # ---- the same path chunk by chunk (any input size; one rank of a multi-GPU run when `exchange` is given) ----
    def compressor(self, params: dict, qual_params=None, qual_ctx: "Context | None" = None, exchange=None, expected_bases: int = 0) -> "Compressor":
        """cl_compressor_create.  qual_params: (mode, source, level, fwd, rev) or None."""
        P = N.CompressParams()
        for k_, v in params.items():
            setattr(P, k_, v)
        Q = None
        if qual_params is not None:
            mode, source, level, fwd, rev = qual_params
            Q = N.QualParams(mode=mode, source=source, level=level, n_fwd=len(fwd), n_rev=len(rev))
            for i, v in enumerate(fwd):
                Q.fwd[i] = v
            for i, v in enumerate(rev):
                Q.rev[i] = v
        h = N._P()
        _check(self, self.lib.cl_compressor_create(self.h, qual_ctx.h if qual_ctx is not None else None, C.byref(P), C.byref(Q) if Q is not None else None,
                                                   C.byref(exchange.c_struct) if exchange is not None else None, int(expected_bases), C.byref(h)))
        return Compressor(self, h, qual_ctx, Q is not None, exchange)

    # ---- a14 ----
    def dna_coder(self, max_alt_refs: int, level: int, start_read_id: int = 0) -> "DnaCoder":
        h = N._P()
        _check(self, self.lib.cl_dna_coder_create(self.h, max_alt_refs, level, start_read_id, C.byref(h)))
        return DnaCoder(self, h)

    def sort_u64(self, keys: torch.Tensor, vals: torch.Tensor | None = None, begin_bit=0, end_bit=64):
        if vals is None:
            _check(self, self.lib.cl_sort_u64(self.h, keys.data_ptr(), keys.numel(), begin_bit, end_bit))
        else:
            _check(self, self.lib.cl_sort_u64_u32(self.h, keys.data_ptr(), vals.data_ptr(), keys.numel(), begin_bit, end_bit))


class _Obj:
    _free = None

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def free(self):
        if self.h:
            getattr(self.ctx.lib, self._free)(self.h)
            self.h = None

    def _arr(self, getter, n, dtype):
        ptr = getattr(self.ctx.lib, getter)(self.h)
        return _view(ptr, n, dtype, self.ctx.device)


class Reads(_Obj):
    _free = "cl_reads_free"

    @property
    def n_reads(self): return self.ctx.lib.cl_reads_count(self.h)
    @property
    def total_bases(self): return self.ctx.lib.cl_reads_total_bases(self.h)
    @property
    def total_words(self): return self.ctx.lib.cl_reads_total_words(self.h)
    def packed(self): return self._arr("cl_reads_packed", self.total_words + 1, torch.int64)
    def invalid(self): return self._arr("cl_reads_invalid", self.total_words + 1, torch.int32)
    def word_offsets(self): return self._arr("cl_reads_word_offsets", self.n_reads + 1, torch.int64)
    def lengths(self): return self._arr("cl_reads_lengths", self.n_reads, torch.int32)
    def has_n(self): return self._arr("cl_reads_has_n", self.n_reads, torch.uint8)

    def compact(self, i: int) -> bytes:
        buf = np.zeros(1 << 20, np.uint8)
        n = C.c_uint64(0)
        st = self.ctx.lib.cl_reads_compact(self.ctx.h, self.h, i, buf.ctypes.data, buf.size, C.byref(n))
        if st == N.CL_E_CAPACITY:
            buf = np.zeros(n.value, np.uint8)
            st = self.ctx.lib.cl_reads_compact(self.ctx.h, self.h, i, buf.ctypes.data, buf.size, C.byref(n))
        _check(self.ctx, st)
        return bytes(buf[:n.value])


class KmerSet(_Obj):
    _free = "cl_kmer_set_free"

    @property
    def size(self): return self.ctx.lib.cl_kmer_set_size(self.h)
    def keys(self): return self._arr("cl_kmer_set_keys", self.size, torch.int64)
    def counts(self): return self._arr("cl_kmer_set_counts", self.size, torch.int32)

    def check(self, kmers: torch.Tensor) -> torch.Tensor:
        out = torch.empty(kmers.numel(), dtype=torch.uint8, device=self.ctx.device)
        _check(self.ctx, self.ctx.lib.cl_kmer_set_check(self.ctx.h, self.h, kmers.data_ptr(), kmers.numel(), out.data_ptr()))
        return out


class KmerLists(_Obj):
    _free = "cl_kmer_lists_free"

    @property
    def n_reads(self): return self.ctx.lib.cl_kmer_lists_reads(self.h)
    @property
    def total(self): return self.ctx.lib.cl_kmer_lists_total(self.h)
    def offsets(self): return self._arr("cl_kmer_lists_offsets", self.n_reads + 1, torch.int64)
    def kmers(self): return self._arr("cl_kmer_lists_kmers", self.total, torch.int64)
    def ids(self): return self._arr("cl_kmer_lists_ids", self.total, torch.int32)
    def pos(self): return self._arr("cl_kmer_lists_pos", self.total, torch.int32)


class Index(_Obj):
    _free = "cl_index_free"

    @property
    def n_refs(self): return self.ctx.lib.cl_index_n_refs(self.h)
    @property
    def entries(self): return self.ctx.lib.cl_index_entries(self.h)


class QualCoder(_Obj):
    _free = "cl_qual_coder_free"

    def encode(self, reads: "Reads", quals: torch.Tensor, qual_off: torch.Tensor, part_bounds, flags: torch.Tensor | None = None):
        """Returns (payload bytes tensor on device, list of part sizes)."""
        ctx = self.ctx
        pb = np.ascontiguousarray(part_bounds, dtype=np.uint32)
        n_parts = len(pb) - 1
        sizes = np.zeros(max(n_parts, 1), np.uint64)
        cap = max(4096, int(quals.numel() * 1.35) + 64 * n_parts)     # generous: org mode on noisy data is ~1 B/base
        while True:
            out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
            n = C.c_uint64(0)
            st = ctx.lib.cl_qual_encode(ctx.h, self.h, reads.h, quals.data_ptr(), qual_off.data_ptr(),
                                        None if flags is None else flags.data_ptr(), pb.ctypes.data, n_parts,
                                        out.data_ptr(), cap, sizes.ctypes.data, C.byref(n))
            if st == N.CL_E_CAPACITY and n.value > cap:
                raise N.ColordHipError(st, "qual output capacity exceeded after models advanced; retry with a larger cap")
            _check(ctx, st)
            return out[:n.value], [int(x) for x in sizes[:n_parts]]


class Compressor(_Obj):
    """cl_compressor: pass 1 / reference listing / pass 2 over the chunks of an input (csrc/stream.hip)."""
    _free = "cl_compressor_free"

    def __init__(self, ctx, h, qual_ctx, has_qual, exchange):
        super().__init__(ctx, h)
        self.qual_ctx, self.has_qual, self.exchange = qual_ctx, has_qual, exchange       # (the exchange's callbacks must outlive the handle)

    def count_add(self, reads: "Reads"):
        _check(self.ctx, self.ctx.lib.cl_compressor_count_add(self.h, reads.h))

    def count_finish(self):
        st = N.KmerStats()
        _check(self.ctx, self.ctx.lib.cl_compressor_count_finish(self.h, C.byref(st)))
        return st

    def refs_add(self, reads: "Reads"):
        _check(self.ctx, self.ctx.lib.cl_compressor_refs_add(self.h, reads.h))

    def refs_finish(self):
        _check(self.ctx, self.ctx.lib.cl_compressor_refs_finish(self.h))

    def info(self) -> dict:
        st = N.KmerStats(); a, b, m = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0); r, nr = C.c_uint32(0), C.c_uint32(0)
        _check(self.ctx, self.ctx.lib.cl_compressor_info(self.h, C.byref(st), C.byref(a), C.byref(b), C.byref(m), C.byref(r), C.byref(nr)))
        return dict(tot_kmers=st.tot_kmers, n_unique_counted=st.n_unique_counted, first_read=a.value, n_reads_total=b.value, mean_read_len=m.value,
                    sparse_range=r.value, n_refs_total=nr.value)

    def genome_add(self, sequences: "Reads"):
        _check(self.ctx, self.ctx.lib.cl_compressor_genome_add(self.h, sequences.h))

    def pseudo_reads(self, pseudo: "Reads"):
        _check(self.ctx, self.ctx.lib.cl_compressor_pseudo_reads(self.h, pseudo.h))

    def prepare(self, reads: "Reads", pack_bounds, part_bounds=None, quals: torch.Tensor | None = None, base_off: torch.Tensor | None = None):
        """cl_compressor_prepare[_parts]: announce a chunk of a later encode() call; its candidates / anchors / edit scripts are
        computed in the background on an encode lane while the chunks before it are coded — and, with the part bounds of that
        encode() call, the model-independent half of its `dna` coding (walks, sort by context) on the preparation thread."""
        kb = np.ascontiguousarray(np.asarray(pack_bounds, dtype=np.uint32))
        if part_bounds is None:
            _check(self.ctx, self.ctx.lib.cl_compressor_prepare(self.h, reads.h, kb.ctypes.data, len(kb) - 1))
        else:
            pb = np.ascontiguousarray(np.asarray(part_bounds, dtype=np.uint32))
            _check(self.ctx, self.ctx.lib.cl_compressor_prepare_parts(self.h, reads.h, kb.ctypes.data, len(kb) - 1, pb.ctypes.data, len(pb) - 1,
                                                                      quals.data_ptr() if quals is not None else None, base_off.data_ptr() if (quals is not None and base_off is not None) else None))

    def encode(self, reads: "Reads", part_bounds, pack_bounds, quals: torch.Tensor | None = None, base_off: torch.Tensor | None = None,
               dna_out: torch.Tensor | None = None, qual_out: torch.Tensor | None = None):
        """One chunk of pass 2.  Returns (dna payload, dna part sizes, qual payload or None, qual part sizes, info dict); with
        dna_out / qual_out the payloads are written to the START of those caller tensors (views are returned)."""
        ctx = self.ctx
        pb = np.ascontiguousarray(np.asarray(part_bounds, dtype=np.uint32)); kb = np.ascontiguousarray(np.asarray(pack_bounds, dtype=np.uint32))
        npart = len(pb) - 1
        if dna_out is None:
            dna_out = torch.empty(int(reads.total_bases) + 64 * npart + 4096, dtype=torch.uint8, device=ctx.device)
        dsz, qsz = np.zeros(max(npart, 1), np.uint64), np.zeros(max(npart, 1), np.uint64)
        if self.has_qual and qual_out is None:
            qual_out = torch.empty(int(int(reads.total_bases) * 1.35) + 64 * npart + 4096, dtype=torch.uint8, device=ctx.device)
        info = N.CompressInfo()
        _check(ctx, ctx.lib.cl_compressor_encode(self.h, reads.h, quals.data_ptr() if self.has_qual else None, base_off.contiguous().data_ptr() if self.has_qual else None,
                                                 pb.ctypes.data, npart, kb.ctypes.data, len(kb) - 1, dna_out.data_ptr(), dna_out.numel(), dsz.ctypes.data,
                                                 qual_out.data_ptr() if self.has_qual else None, qual_out.numel() if self.has_qual else 0, qsz.ctypes.data, C.byref(info)))
        if self.qual_ctx is not None and self.qual_ctx is not ctx:
            _check(self.qual_ctx, N.CL_OK)                  # collect the kernel times of the concurrent quality stream
        inf = {n_: getattr(info, n_) for n_, _ in N.CompressInfo._fields_ if n_ != "pad"}
        return dna_out[:inf["dna_bytes"]], dsz[:npart], (qual_out[:inf["qual_bytes"]] if self.has_qual else None), qsz[:npart], inf


class DnaCoder(_Obj):
    _free = "cl_dna_coder_free"

    def encode(self, refs: "Reads", es: torch.Tensor, es_off: torch.Tensor, es_ntuples: torch.Tensor, part_bounds):
        """Returns (payload bytes tensor on device, list of part sizes)."""
        ctx = self.ctx
        pb = np.ascontiguousarray(part_bounds, dtype=np.uint32)
        n_parts = len(pb) - 1
        n_reads = es_off.numel() - 1
        sizes = np.zeros(max(n_parts, 1), np.uint64)
        cap = max(4096, int(es.numel() * 0.6) + 64 * n_parts)
        out = torch.empty(cap, dtype=torch.uint8, device=ctx.device)
        n = C.c_uint64(0)
        st = ctx.lib.cl_dna_encode(ctx.h, self.h, refs.h, es.data_ptr(), es_off.data_ptr(), es_ntuples.data_ptr(), n_reads,
                                   pb.ctypes.data, n_parts, out.data_ptr(), cap, sizes.ctypes.data, C.byref(n))
        _check(ctx, st)
        return out[:n.value], [int(x) for x in sizes[:n_parts]]


class Anchors(_Obj):
    _free = "cl_anchors_free"

    def __init__(self, ctx, h, n_reads, c):
        super().__init__(ctx, h)
        self.n_reads, self.c = n_reads, c

    @property
    def total(self): return self.ctx.lib.cl_anchors_total(self.h)
    def n_cands(self): return self._arr("cl_anchors_n_cands", self.n_reads, torch.int32)
    def cands(self): return self._arr("cl_anchors_cands", self.n_reads * self.c * 4, torch.int32).view(self.n_reads, self.c, 4)
    def cand_offsets(self): return self._arr("cl_anchors_cand_offsets", self.n_reads * self.c + 1, torch.int64)
    def data(self): return self._arr("cl_anchors_data", self.total * 3, torch.int32).view(-1, 3)
