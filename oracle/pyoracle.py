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
