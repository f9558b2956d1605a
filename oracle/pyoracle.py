"""ctypes view of liboracle.so (TEST INFRASTRUCTURE ONLY — imported by tests/, smoke() and the
cpu_baseline leg of bench.py; never by colord_amd/)."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


class KmerStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("tot_kmers", C.c_uint64), ("n_unique", C.c_uint64),
                ("n_unique_counted", C.c_uint64), ("total_count_filtered", C.c_uint64)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
        L = C.CDLL(so)
        L.orc_hash_mm.restype = C.c_uint64
        L.orc_hash_mm.argtypes = [C.c_uint64]
        L.orc_kmer_scan.restype = C.c_size_t
        L.orc_kmer_scan.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
        L.orc_count_filter.restype = C.c_size_t
        L.orc_count_filter.argtypes = [u64p, C.c_size_t, C.c_uint32, C.c_uint32, u64p, u32p, C.POINTER(KmerStats)]
        L.orc_accepted_kmers.restype = C.c_size_t
        L.orc_accepted_kmers.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_uint32, u64p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_ref_accept.restype = None
        L.orc_ref_accept.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, u8p]
        L.orc_estimator_logs.restype = None
        L.orc_estimator_logs.argtypes = [u32p, u32p, C.c_size_t, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")]
        L.orc_graph_new.restype = C.c_void_p
        L.orc_graph_new.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_graph_free.argtypes = [C.c_void_p]
        L.orc_graph_add_pseudo.argtypes = [C.c_void_p, u64p, C.c_size_t]
        L.orc_graph_next_read.restype = C.c_uint32
        L.orc_graph_next_read.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int, u32p, u32p, C.c_void_p, C.c_void_p]
        L.orc_graph_n_refs.restype = C.c_uint32
        L.orc_graph_n_refs.argtypes = [C.c_void_p]
        L.orc_refread_compact.restype = C.c_size_t
        L.orc_refread_compact.argtypes = [u8p, C.c_size_t, u8p]
        L.orc_qual_new.restype = C.c_void_p
        L.orc_qual_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_qual_free.argtypes = [C.c_void_p]
        L.orc_qual_encode.argtypes = [C.c_void_p, u8p, u8p, C.c_uint32, C.c_void_p]
        L.orc_qual_finish_part.restype = C.c_size_t
        L.orc_qual_finish_part.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_qual_set_input.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_qual_decode.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_void_p, u8p]
        L.orc_es_flags.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, u8p]
        L.orc_dna_new.restype = C.c_void_p
        L.orc_dna_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.orc_dna_free.argtypes = [C.c_void_p]
        L.orc_dna_add_ref.argtypes = [C.c_void_p, u8p, C.c_uint32]
        L.orc_dna_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]
        L.orc_dna_finish_part.restype = C.c_size_t
        L.orc_dna_finish_part.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_encoder_new.restype = C.c_void_p
        L.orc_encoder_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_encoder_free.argtypes = [C.c_void_p]
        L.orc_encoder_add_ref.argtypes = [C.c_void_p, u8p, C.c_uint32]
        L.orc_encoder_new_pack.argtypes = [C.c_void_p]
        L.orc_encoder_candidates.restype = C.c_uint32
        L.orc_encoder_candidates.argtypes = [C.c_void_p, u8p, C.c_uint32, u32p, C.c_uint32, C.c_void_p, C.c_void_p, u32p, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_encoder_encode.restype = C.c_size_t
        L.orc_encoder_encode.argtypes = [C.c_void_p, u8p, C.c_uint32, C.c_int, u32p, C.c_uint32, C.c_void_p, C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


def kmer_scan(bases: np.ndarray, k: int, f: int) -> np.ndarray:
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    n = lib().orc_kmer_scan(b, len(b), k, f, None, 0)
    out = np.empty(n, dtype=np.uint64)
    lib().orc_kmer_scan(b, len(b), k, f, out.ctypes.data, n)
    return out


def kmer_scan_reads(rs, k: int, f: int) -> np.ndarray:
    return np.concatenate([kmer_scan(rs.read(i), k, f) for i in range(rs.n_reads)] + [np.empty(0, np.uint64)])


def count_filter(kmers: np.ndarray, ci: int, cs: int):
    km = np.ascontiguousarray(kmers, dtype=np.uint64).copy()
    keys = np.empty(len(km), np.uint64)
    cnt = np.empty(len(km), np.uint32)
    st = KmerStats()
    m = lib().orc_count_filter(km, len(km), ci, cs, keys, cnt, C.byref(st))
    return keys[:m].copy(), cnt[:m].copy(), st


def accepted_kmers(bases: np.ndarray, k: int, f: int, kept: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(bases, dtype=np.uint8)
    out = np.empty(max(len(b), 1), dtype=np.uint64)
    n = lib().orc_accepted_kmers(b, len(b), k, f, kept, len(kept), out.ctypes.data, len(out))
    return out[:n].copy()


def ref_accept(n_reads: int, n_pseudo: int, rng: int, exponent: float) -> np.ndarray:
    out = np.zeros(n_reads + n_pseudo, np.uint8)
    lib().orc_ref_accept(n_reads, n_pseudo, rng, exponent, out)
    return out


def estimator_logs(count: np.ndarray, total: np.ndarray) -> np.ndarray:
    """-log2(count * (1/total)) by the host's libm (calc_logs, utils.h:800-810)."""
    c = np.ascontiguousarray(count, dtype=np.uint32); t = np.ascontiguousarray(total, dtype=np.uint32)
    out = np.empty(len(c), np.float64)
    lib().orc_estimator_logs(c, t, len(c), out)
    return out


class Graph:
    def __init__(self, max_candidates: int, max_kmer_count: int):
        self.c = max_candidates
        self.h = lib().orc_graph_new(max_candidates, max_kmer_count)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_graph_free(self.h)
            self.h = None

    def add_pseudo(self, kmers):
        k = np.ascontiguousarray(kmers, np.uint64)
        lib().orc_graph_add_pseudo(self.h, k, len(k))

    def next_read(self, kmers, accept: bool, hifi: bool = False):
        k = np.ascontiguousarray(kmers, np.uint64)
        refs = np.zeros(self.c, np.uint32)
        votes = np.zeros(self.c, np.uint32)
        if hifi:
            common = np.zeros(max(1, self.c * len(k)), np.uint64)
            off = np.zeros(self.c + 1, np.uint32)
            n = lib().orc_graph_next_read(self.h, k, len(k), int(accept), refs, votes, common.ctypes.data, off.ctypes.data)
            return refs[:n].copy(), votes[:n].copy(), [common[off[i]:off[i + 1]].copy() for i in range(n)]
        n = lib().orc_graph_next_read(self.h, k, len(k), int(accept), refs, votes, None, None)
        return refs[:n].copy(), votes[:n].copy(), None


# QualityComprMode (params.h:33-43) and the default thresholds / representatives (arg_parse.cpp:32-84)
QM = dict(org=0, avg5=1, avg4=2, avg2=3, fix5=4, fix4=5, fix2=6, avg=7, none=8)
QUAL_DEFAULTS = {
    0: ((), ()), 7: ((), ()), 8: ((), (0,)),
    1: ((7, 14, 26, 93), ()), 2: ((7, 14, 26), ()), 3: ((7,), ()),
    4: ((7, 14, 26, 93), (3, 10, 18, 35, 93)), 5: ((7, 14, 26), (3, 10, 18, 35)), 6: ((7,), (1, 13)),
}


def es_flags(es: bytes, read_len: int) -> np.ndarray:
    out = np.zeros(max(read_len, 1), np.uint8)
    buf = (C.c_char * len(es)).from_buffer_copy(es)
    lib().orc_es_flags(buf, len(es), read_len, out)
    return out[:read_len]


class QualCoder:
    """CQualityCoder + the per-pack part framing of CEntrComprQuals / CEntrDecomprQuals."""
    def __init__(self, compress: bool, mode: int, source: int, level: int, fwd=None, rev=None):
        d = QUAL_DEFAULTS[mode]
        fwd = np.asarray(d[0] if fwd is None else fwd, np.uint32)
        rev = np.asarray(d[1] if rev is None else rev, np.uint32)
        self.h = lib().orc_qual_new(int(compress), mode, source, level, fwd.ctypes.data, len(fwd), rev.ctypes.data, len(rev))
        self._keep = None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_qual_free(self.h)
            self.h = None

    def encode(self, bases, qual, flags=None):
        b = np.ascontiguousarray(bases, np.uint8)
        q = np.ascontiguousarray(qual, np.uint8)
        f = None if flags is None else np.ascontiguousarray(flags, np.uint8)
        if len(b) == 0:
            b = np.zeros(1, np.uint8); q = np.zeros(1, np.uint8)
        lib().orc_qual_encode(self.h, b, q, len(bases), None if f is None else f.ctypes.data)

    def finish_part(self) -> bytes:
        n = lib().orc_qual_finish_part(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        m = lib().orc_qual_finish_part(self.h, buf.ctypes.data, n)
        return buf[:m].tobytes()

    def set_input(self, data: bytes):
        self._keep = (C.c_char * max(1, len(data))).from_buffer_copy(data or b"\0")
        lib().orc_qual_set_input(self.h, self._keep, len(data))

    def decode(self, bases, flags=None) -> np.ndarray:
        b = np.ascontiguousarray(bases, np.uint8)
        out = np.zeros(max(1, len(b)), np.uint8)
        f = None if flags is None else np.ascontiguousarray(flags, np.uint8)
        if len(b) == 0:
            b = np.zeros(1, np.uint8)
        lib().orc_qual_decode(self.h, b, len(bases), None if f is None else f.ctypes.data, out)
        return out[:len(bases)]


class DnaCoder:
    """CDNACoder + the part framing of CEntrComprReads."""
    def __init__(self, max_alt_refs: int, level: int, start_read_id: int = 0):
        self.h = lib().orc_dna_new(1, max_alt_refs, level, start_read_id)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_dna_free(self.h)
            self.h = None

    def add_ref(self, bases):
        b = np.ascontiguousarray(bases, np.uint8)
        lib().orc_dna_add_ref(self.h, b if len(b) else np.zeros(1, np.uint8), len(b))

    def encode(self, es: bytes, n_tuples: int):
        buf = (C.c_char * len(es)).from_buffer_copy(es)
        lib().orc_dna_encode(self.h, buf, len(es), n_tuples)

    def finish_part(self) -> bytes:
        n = lib().orc_dna_finish_part(self.h, None, 0)
        buf = np.zeros(n, np.uint8)
        m = lib().orc_dna_finish_part(self.h, buf.ctypes.data, n)
        return buf[:m].tobytes()


class Encoder:
    """CEncoder for one encoder thread: per-pack estimator, reference reads added in reference-id order."""
    def __init__(self, a, k, f, source, frac_always=0.9, frac_min=0.5, max_matches_mult=10.0, cost_mult=1.0, min_part_alt=64, max_rec=3, min_anchors=1):
        self.h = lib().orc_encoder_new(a, k, f, source, frac_always, frac_min, max_matches_mult, cost_mult, min_part_alt, max_rec, min_anchors)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_encoder_free(self.h)
            self.h = None

    def add_ref(self, bases):
        b = np.ascontiguousarray(bases, np.uint8)
        lib().orc_encoder_add_ref(self.h, b if len(b) else np.zeros(1, np.uint8), len(b))

    def new_pack(self):
        lib().orc_encoder_new_pack(self.h)

    def candidates(self, read, neighbours, common=None):
        """[(ref_id, rev, tot_anchor_len, [(len, pos_enc, pos_ref), ...]), ...] as the encoder will use them."""
        r = np.ascontiguousarray(read, np.uint8)
        nb = np.ascontiguousarray(neighbours, np.uint32)
        if len(nb) == 0:
            return []
        cptr = coff_ptr = None
        if common is not None:
            coff = np.concatenate([[0], np.cumsum([len(c) for c in common])]).astype(np.uint32)
            call = np.concatenate(list(common) + [np.zeros(0, np.uint64)]).astype(np.uint64)
            if len(call) == 0:
                call = np.zeros(1, np.uint64)
            cptr, coff_ptr = call.ctypes.data, coff.ctypes.data
        oc = np.zeros(4 * len(nb), np.uint32)
        oa = np.zeros(3 * (len(r) + 16) * 2, np.uint32)
        na = C.c_size_t(0)
        n = lib().orc_encoder_candidates(self.h, r, len(r), nb, len(nb), cptr, coff_ptr, oc, oa, len(oa) // 3, C.byref(na))
        out, o = [], 0
        for i in range(n):
            k = int(oc[4 * i + 3])
            out.append((int(oc[4 * i]), int(oc[4 * i + 1]), int(oc[4 * i + 2]), [tuple(int(x) for x in oa[3 * (o + j):3 * (o + j) + 3]) for j in range(k)]))
            o += k
        return out

    def encode(self, read, has_n, neighbours, common=None):
        r = np.ascontiguousarray(read, np.uint8)
        nb = np.ascontiguousarray(neighbours, np.uint32)
        if len(nb) == 0:
            nb = np.zeros(1, np.uint32)
        cptr = coff_ptr = None
        if common is not None:
            coff = np.concatenate([[0], np.cumsum([len(c) for c in common])]).astype(np.uint32)
            call = np.concatenate(list(common) + [np.zeros(0, np.uint64)]).astype(np.uint64)
            if len(call) == 0:
                call = np.zeros(1, np.uint64)
            cptr, coff_ptr = call.ctypes.data, coff.ctypes.data
        out = np.zeros(2 * len(r) + 64, np.uint8)
        nt = C.c_uint32(0)
        n = lib().orc_encoder_encode(self.h, r if len(r) else np.zeros(1, np.uint8), len(r), int(has_n), nb, len(neighbours), cptr, coff_ptr, out, len(out), C.byref(nt))
        return out[:n].tobytes(), nt.value
