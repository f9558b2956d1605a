"""ctypes binding of libcolord_hip.so (the C ABI in include/colord_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Device memory is supplied by the caller (torch CUDA tensors -> data_ptr()).
"""
from __future__ import annotations
import ctypes as C
import os

# more hardware queues than the HIP runtime's default of 4: the contexts of one compressor keep up to ten streams busy and
# streams that share a queue serialise (bench.py has the measurement); only effective before the runtime initialises
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COLORD_HIP_LIBRARY") or os.path.join(_HERE, "libcolord_hip.so")    # (COLORD_HIP_LIBRARY: another build of the same library, for A/B measurements)

CL_OK, CL_E_INVALID, CL_E_HIP, CL_E_CAPACITY, CL_E_NOMEM, CL_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5


class ColordHipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"colord_hip error {status}: {msg}")
        self.status = status


class QualParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("source", C.c_int32), ("level", C.c_int32), ("n_fwd", C.c_uint32), ("fwd", C.c_uint32 * 8),
                ("n_rev", C.c_uint32), ("rev", C.c_uint32 * 8)]


class KmerStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("tot_kmers", C.c_uint64), ("n_unique", C.c_uint64),
                ("n_unique_counted", C.c_uint64), ("total_count_filtered", C.c_uint64)]


_P = C.c_void_p
class CompressParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("f", C.c_uint32), ("ci", C.c_uint32), ("cs", C.c_uint32), ("c", C.c_uint32),
                ("anchor_len", C.c_uint32), ("min_part_alt", C.c_uint32), ("max_rec", C.c_uint32), ("min_anchors", C.c_uint32),
                ("level", C.c_int32), ("source", C.c_int32), ("sparse", C.c_int32),
                ("sparse_g", C.c_double), ("sparse_exponent", C.c_double),
                ("cost_mult", C.c_double), ("frac_always", C.c_double), ("frac_min", C.c_double), ("max_matches_mult", C.c_double)]


class CompressInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_reads", "n_bases", "tot_kmers", "n_kept_kmers", "n_refs", "n_anchors", "tuple_bytes", "dna_bytes", "qual_bytes")] + \
               [("sparse_range", C.c_uint32), ("pad", C.c_uint32)]


_EXCH_GATHER_HOST = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint64))
_EXCH_A2AV = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64))
_EXCH_GATHERV = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64))


class Exchange(C.Structure):
    """cl_exchange: the collectives a multi-GPU cl_compressor calls back into (colord_amd/parallel.py fills it)."""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_uint32), ("world", C.c_uint32),
                ("all_gather_host", _EXCH_GATHER_HOST), ("all_to_all_v", _EXCH_A2AV), ("all_gather_v", _EXCH_GATHERV)]


_SIG = {
    "cl_ctx_create": (C.c_int32, [C.c_int, C.POINTER(_P)]),
    "cl_ctx_destroy": (None, [_P]),
    "cl_last_error": (C.c_char_p, [_P]),
    "cl_ctx_stream": (_P, [_P]),
    "cl_ctx_last_kernel_ms": (C.c_int32, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    "cl_ctx_set_timing": (None, [_P, C.c_int]),
    "cl_ctx_kernel_times": (C.c_int32, [_P, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_reads_pack": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_int, C.POINTER(_P)]),
    "cl_reads_free": (None, [_P]),
    "cl_reads_count": (C.c_uint32, [_P]),
    "cl_reads_total_bases": (C.c_uint64, [_P]),
    "cl_reads_total_words": (C.c_uint64, [_P]),
    "cl_reads_packed": (_P, [_P]),
    "cl_reads_invalid": (_P, [_P]),
    "cl_reads_word_offsets": (_P, [_P]),
    "cl_reads_lengths": (_P, [_P]),
    "cl_reads_has_n": (_P, [_P]),
    "cl_reads_compact": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_kmer_scan": (C.c_int32, [_P, _P, C.c_uint32, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_kmer_count_filter": (C.c_int32, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(KmerStats)]),
    "cl_kmer_set_create": (C.c_int32, [_P, _P, _P, C.c_uint64, C.c_uint32, C.POINTER(_P)]),
    "cl_kmer_set_free": (None, [_P]),
    "cl_kmer_set_size": (C.c_uint64, [_P]),
    "cl_kmer_set_keys": (_P, [_P]),
    "cl_kmer_set_counts": (_P, [_P]),
    "cl_kmer_set_check": (C.c_int32, [_P, _P, _P, C.c_uint64, _P]),
    "cl_accepted_kmers": (C.c_int32, [_P, _P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "cl_kmer_lists_free": (None, [_P]),
    "cl_kmer_lists_reads": (C.c_uint32, [_P]),
    "cl_kmer_lists_total": (C.c_uint64, [_P]),
    "cl_kmer_lists_offsets": (_P, [_P]),
    "cl_kmer_lists_kmers": (_P, [_P]),
    "cl_kmer_lists_ids": (_P, [_P]),
    "cl_kmer_lists_pos": (_P, [_P]),
    "cl_ref_accept": (C.c_int32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, _P]),
    "cl_index_build": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "cl_index_entries_of": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, C.c_uint64, C.POINTER(C.c_uint64), _P, C.POINTER(C.c_uint32)]),
    "cl_index_build_pairs": (C.c_int32, [_P, _P, _P, _P, C.c_uint64, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "cl_index_free": (None, [_P]),
    "cl_index_n_refs": (C.c_uint32, [_P]),
    "cl_index_entries": (C.c_uint64, [_P]),
    "cl_index_ref_rank": (_P, [_P]),
    "cl_candidates": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, _P]),
    "cl_candidates_common": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, _P, _P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_qual_coder_create": (C.c_int32, [_P, C.POINTER(QualParams), C.POINTER(_P)]),
    "cl_qual_coder_ctx": (_P, [_P]),
    "cl_qual_coder_free": (None, [_P]),
    "cl_qual_encode": (C.c_int32, [_P, _P, _P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "cl_anchor_candidates": (C.c_int32, [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.POINTER(_P)]),
    "cl_anchor_candidates_hifi": (C.c_int32, [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, C.POINTER(_P)]),
    "cl_anchors_free": (None, [_P]),
    "cl_anchors_total": (C.c_uint64, [_P]),
    "cl_anchors_n_cands": (_P, [_P]),
    "cl_anchors_cands": (_P, [_P]),
    "cl_anchors_cand_offsets": (_P, [_P]),
    "cl_anchors_data": (_P, [_P]),
    "cl_encode_plain": (C.c_int32, [_P, _P, _P, C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
    "cl_reads_select": (C.c_int32, [_P, _P, _P, C.POINTER(_P)]),
    "cl_reads_from_arena": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, C.POINTER(_P)]),
    "cl_estimator_logs": (C.c_int32, [_P, _P, _P, C.c_uint64, _P]),
    "cl_es_flags": (C.c_int32, [_P, _P, _P, _P, _P, _P]),
    "cl_id_coder_create": (C.c_int32, [C.c_int32, C.POINTER(_P)]),
    "cl_id_coder_free": (None, [_P]),
    "cl_id_coder_error": (C.c_char_p, [_P]),
    "cl_id_encode_part": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_compress_shard": (C.c_int32, [_P, C.POINTER(CompressParams), _P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, C.c_uint64, _P, _P, C.c_uint64, _P,
                                      C.POINTER(CompressInfo)]),
    "cl_candidates_at": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, _P, _P, _P]),
    "cl_compressor_create": (C.c_int32, [_P, _P, C.POINTER(CompressParams), C.POINTER(QualParams), C.POINTER(Exchange), C.c_uint64, C.POINTER(_P)]),
    "cl_compressor_free": (None, [_P]),
    "cl_compressor_count_add": (C.c_int32, [_P, _P]),
    "cl_compressor_count_finish": (C.c_int32, [_P, C.POINTER(KmerStats)]),
    "cl_compressor_refs_add": (C.c_int32, [_P, _P]),
    "cl_compressor_refs_finish": (C.c_int32, [_P]),
    "cl_compressor_encode": (C.c_int32, [_P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, _P, _P, C.c_uint64, _P, C.POINTER(CompressInfo)]),
    "cl_compressor_genome_add": (C.c_int32, [_P, _P]),
    "cl_compressor_pseudo_reads": (C.c_int32, [_P, _P]),
    "cl_genome_encode": (C.c_int32, [_P, _P, C.c_uint32, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cl_genome_decode": (C.c_int32, [_P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "cl_genome_md5": (C.c_int32, [_P, _P, C.c_uint32, _P]),
    "cl_compressor_prepare": (C.c_int32, [_P, _P, _P, C.c_uint32]),
    "cl_compressor_prepare_parts": (C.c_int32, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P]),
    "cl_compressor_info": (C.c_int32, [_P, C.POINTER(KmerStats), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "cl_dna_decoder_create": (C.c_int32, [C.c_uint32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32, C.c_uint32, C.c_double, C.POINTER(_P)]),
    "cl_dna_decoder_free": (None, [_P]),
    "cl_dna_decoder_error": (C.c_char_p, [_P]),
    "cl_dna_decoder_add_ref": (C.c_int32, [_P, _P, C.c_uint32]),
    "cl_dna_decoder_new_domain": (C.c_int32, [_P]),
    "cl_dna_decode_part": (C.c_int32, [_P, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "cl_qual_decoder_create": (C.c_int32, [C.POINTER(QualParams), C.POINTER(_P)]),
    "cl_qual_decoder_free": (None, [_P]),
    "cl_qual_decoder_new_domain": (C.c_int32, [_P]),
    "cl_qual_decode_part": (C.c_int32, [_P, _P, C.c_uint64, _P, _P, C.c_uint32, _P]),
    "cl_id_decoder_create": (C.c_int32, [C.c_int32, C.POINTER(_P)]),
    "cl_id_decoder_free": (None, [_P]),
    "cl_id_decode_part": (C.c_int32, [_P, _P, C.c_uint64, C.c_uint32, _P, C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
    "cl_encode_reads": (C.c_int32, [_P, _P, _P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, _P, C.c_uint32, _P, C.c_uint64, _P, _P, C.POINTER(C.c_uint64)]),
    "cl_dna_coder_create": (C.c_int32, [_P, C.c_uint32, C.c_int32, C.c_uint32, C.POINTER(_P)]),
    "cl_dna_coder_free": (None, [_P]),
    "cl_dna_encode": (C.c_int32, [_P, _P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint64, _P, C.POINTER(C.c_uint64)]),
    "cl_sort_u64": (C.c_int32, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32]),
    "cl_sort_u64_u32": (C.c_int32, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint32]),
}

_lib = None


def load():
    """Load libcolord_hip.so and declare every entry point of include/colord_hip.h.  Fails loudly."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ColordHipError(CL_E_HIP, f"{LIB_PATH} is missing: run `make -C colord_amd/csrc` "
                                 "(or __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def exported_names():
    return sorted(_SIG)
