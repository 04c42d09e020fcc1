"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes binding over oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module. It is the
checker, never the product: rucene_amd/ must not import it (tests/test_layout.py enforces that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# shared 32-byte layout: BlockTermState (oracle/postings.hpp) == rgpu_term_state (include/rucene_gpu.h)
TERM_STATE_DTYPE = np.dtype(
    [("doc_start_fp", "<i8"), ("skip_offset", "<i8"), ("total_term_freq", "<i8"), ("doc_freq", "<i4"),
     ("singleton_doc_id", "<i4")], align=True)
assert TERM_STATE_DTYPE.itemsize == 32

OP_TERM, OP_AND, OP_OR = 0, 1, 2
TIE_RUST_HEAP, TIE_CANONICAL = 0, 1
FLAG_FREQS = 1 << 3
NO_MORE_DOCS = 2**31 - 1


def build(force=False):
    """Compile oracle/liboracle.so with g++ (make)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("oracle_capi.cpp", "store.hpp", "packed.hpp", "postings.hpp", "search.hpp", "norms.hpp", "fst.hpp",
                      "blocktree.hpp", "field_infos.hpp", "segment_infos.hpp", "positions.hpp", "phrase.hpp", "compound.hpp", "elias_fano.hpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _declare(L):
    L.orc_last_error.restype = C.c_char_p
    u8p, u32p, i32p, i64p, f32p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_int32, C.c_int64, C.c_float, C.c_uint64))
    vp = C.c_void_p
    sig = {
        "orc_bp128_pack": (C.c_int, [u32p, u8p, C.c_int]),
        "orc_bp128_unpack": (C.c_int, [u8p, u32p, C.c_int]),
        "orc_bp128_delta_pack": (C.c_int, [u32p, u8p, C.c_uint32, C.c_int]),
        "orc_bp128_delta_unpack": (C.c_int, [u8p, u32p, C.c_uint32, C.c_int]),
        "orc_max_bits_num": (C.c_int, [u32p, C.c_int]),
        "orc_simd_block_advance": (C.c_int, [i32p, C.c_int32]),
        "orc_max_data_size": (C.c_int, []),
        "orc_format_fastest": (C.c_int, [C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_format_byte_count": (C.c_int64, [C.c_int, C.c_int, C.c_int]),
        "orc_legacy_decode": (C.c_int, [C.c_int, C.c_int, u8p, i32p, C.c_int]),
        "orc_legacy_encode": (C.c_int, [C.c_int, C.c_int, i32p, u8p, C.c_int]),
        "orc_legacy_counts": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "orc_write_vint": (C.c_int, [C.c_int32, u8p]),
        "orc_write_vlong": (C.c_int, [C.c_int64, u8p]),
        "orc_read_vint": (C.c_int, [u8p, C.c_int, i32p]),
        "orc_read_vlong": (C.c_int, [u8p, C.c_int, i64p]),
        "orc_crc32": (C.c_uint32, [u8p, C.c_int64]),
        "orc_float_to_byte315": (C.c_uint8, [C.c_float]),
        "orc_byte315_to_float": (C.c_float, [C.c_uint8]),
        "orc_origin_float_to_byte": (C.c_uint8, [C.c_float]),
        "orc_origin_byte_to_float": (C.c_float, [C.c_uint8]),
        "orc_norm_table": (C.c_float, [C.c_int]),
        "orc_bm25_encode_norm": (C.c_uint8, [C.c_float, C.c_int32]),
        "orc_bm25_idf": (C.c_float, [C.c_int64, C.c_int64, C.c_int64]),
        "orc_bm25_avgdl": (C.c_float, [C.c_int64, C.c_int64, C.c_int64]),
        "orc_bm25_weight": (C.c_float, [C.c_float, C.c_float, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, f32p]),
        "orc_bm25_score": (C.c_float, [C.c_float, C.c_float, C.c_float, C.c_int, C.c_float]),
        "orc_writer_new": (vp, [C.c_int32, C.c_int32, C.c_int, u8p, C.c_char_p]),
        "orc_writer_free": (None, [vp]),
        "orc_writer_start_term": (C.c_int, [vp]),
        "orc_writer_add_docs": (C.c_int, [vp, i32p, i32p, C.c_int64]),
        "orc_writer_finish_term": (C.c_int, [vp, C.c_int32, C.c_int64, vp]),
        "orc_writer_close": (C.c_int64, [vp]),
        "orc_writer_size": (C.c_int64, [vp]),
        "orc_writer_copy": (C.c_int, [vp, u8p]),
        "orc_segment_new": (vp, [u8p, C.c_int64, u8p, u64p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, vp, C.c_int64]),
        "orc_segment_free": (None, [vp]),
        "orc_segment_version": (C.c_int, [vp]),
        "orc_segment_set_index_has_freq": (None, [vp, C.c_int]),
        "orc_writer_set_ef": (None, [vp, C.c_int, C.c_int]),
        "orc_searcher_rescore": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_int]),
        "orc_ef_num_longs_for_bits": (C.c_int64, [C.c_int64]),
        "orc_ef_pack_value": (None, [C.c_int64, C.POINTER(C.c_int64), C.c_int, C.c_int32, C.c_int64]),
        "orc_ef_encode_upper": (C.c_int64, [C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.c_int]),
        "orc_ef_roundtrip": (C.c_int64, [C.POINTER(C.c_int64), C.c_int64, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "orc_postings_new": (vp, [vp, vp, C.c_int]),
        "orc_postings_free": (None, [vp]),
        "orc_postings_next": (C.c_int, [vp, i32p]),
        "orc_postings_advance": (C.c_int, [vp, C.c_int32, i32p]),
        "orc_postings_freq": (C.c_int32, [vp]),
        "orc_postings_doc": (C.c_int32, [vp]),
        "orc_decode_term": (C.c_int64, [vp, vp, i32p, i32p]),
        "orc_decode_terms": (C.c_double, [vp, vp, C.c_int64, i32p, i32p, C.c_int]),
        "orc_searcher_new": (vp, [C.POINTER(vp), C.c_int, C.c_float, C.c_float]),
        "orc_searcher_free": (None, [vp]),
        "orc_searcher_stats_leaf": (C.c_int, [vp]),
        "orc_searcher_override_stats": (None, [vp, vp, C.c_int64]),
        "orc_searcher_term_weight": (C.c_float, [vp, C.c_int64, C.c_float, f32p]),
        "orc_search": (C.c_int, [vp, C.c_int, i64p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p, i64p]),
        "orc_search_batch": (C.c_double, [vp, C.c_int, i32p, i32p, i64p, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p, i64p, u64p]),
        "orc_search_batch_not": (C.c_double, [vp, C.c_int, i32p, i32p, i64p, i32p, i64p, i32p, C.c_int, C.c_int, C.c_int, i32p, f32p,
                                              i32p, i64p, u64p]),
        "orc_search_not": (C.c_int, [vp, C.c_int, i64p, C.c_int, i64p, C.c_int, C.c_int, C.c_int, i32p, f32p, i32p, i64p]),
        "orc_mock_req_not": (C.c_int, [i32p, i32p, C.c_int, i32p, i32p, C.c_int, i32p, C.c_int, i32p, C.c_int]),
        "orc_norms_write": (C.c_int, [i64p, C.c_int32, C.c_int32, u8p, C.c_char_p, u8p, i64p, u8p, i64p]),
        "orc_norms_read": (C.c_int, [u8p, C.c_int64, u8p, C.c_int64, C.c_int32, C.c_int32, i64p]),
        "orc_live_docs_write": (C.c_int, [i64p, C.c_int32, C.c_int32, C.c_int32, u8p, C.c_int64, u8p, i64p]),
        "orc_live_docs_read": (C.c_int, [u8p, C.c_int64, C.c_int32, C.c_int32, i64p]),
        "orc_mock_req_opt": (C.c_int, [i32p, i32p, C.c_int, i32p, i32p, C.c_int, i32p, f32p, C.c_int]),
        "orc_searcher_score_docs": (C.c_int, [vp, C.c_int, i64p, C.c_int, C.c_int, i32p, C.c_int64, f32p, u8p]),
        "orc_search_opt": (C.c_int, [vp, C.c_int, i64p, C.c_int, i64p, C.c_int, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p,
                                     f32p, i32p, i64p]),
        "orc_pos_index_build": (vp, [C.c_int32, C.c_int32, C.c_int32, i64p, i32p, i32p, i64p, i32p]),
        "orc_pos_index_build_ex": (vp, [C.c_int32, C.c_int32, C.c_int32, i64p, i32p, i32p, i64p, i32p, C.c_int32, i32p, i32p, i64p, u8p]),
        "orc_pos_index_pay": (C.c_int64, [vp, u8p]),
        "orc_pos_index_from_files": (vp, [u8p, C.c_int64, u8p, C.c_int64, u8p, C.c_int64, C.c_int32, i64p, C.c_int32]),
        "orc_pos_iterate_everything": (C.c_int64, [vp, C.c_int32, C.c_int32, i32p, C.c_int64, C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int64,
                                                   i32p, i32p, i32p, i32p, C.c_int64, u8p, C.c_int64, C.POINTER(C.c_int64)]),
        "orc_pos_index_free": (None, [vp]),
        "orc_pos_index_copy": (None, [vp, u8p, u8p]),
        "orc_pos_index_sizes": (C.c_int64, [vp, i64p, i64p]),
        "orc_pos_term_state": (C.c_int, [vp, C.c_int32, i64p]),
        "orc_pos_iterate": (C.c_int64, [vp, C.c_int32, i32p, C.c_int64, C.c_int32, C.c_int32, i32p, i32p, i32p, C.c_int64, i32p, C.c_int64]),
        "orc_pos_phrase_freqs": (C.c_int64, [vp, i32p, i32p, C.c_int, i32p, i32p, C.c_int64]),
        "orc_pos_phrase_search": (C.c_int, [vp, i32p, i32p, C.c_int, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, i32p, f32p,
                                            i32p, i64p]),
        "orc_pos_sloppy_freqs": (C.c_int64, [vp, i32p, i32p, C.c_int, C.c_int, i32p, f32p, C.c_int64]),
        "orc_pos_phrase_search_ex": (C.c_int, [vp, i32p, i32p, C.c_int, C.c_int, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                               i32p, f32p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
        "orc_pos_phrase_search_slop": (C.c_int, [vp, i32p, i32p, C.c_int, C.c_int, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, i32p, f32p,
                                                 i32p, i64p]),
        "orc_compound_write": (C.c_int, [C.c_int32, u8p, u8p, i64p, u8p, u8p, i64p, u8p, i64p]),
        "orc_compound_read": (C.c_int, [u8p, C.c_int64, u8p, C.c_int64, u8p, C.c_int32, u8p, C.c_int64, i64p, i64p, i64p]),
        "orc_segment_info_write": (C.c_int, [u8p, u8p, i32p, C.c_int32, C.c_int, u8p, i64p]),
        "orc_segment_info_read": (C.c_int, [u8p, C.c_int64, u8p, i32p, i32p, u8p]),
        "orc_segments_file_write": (C.c_int, [C.c_int64, u8p, C.c_int64, C.c_int32, C.c_int32, u8p, u8p, i64p, i32p, u8p, i64p]),
        "orc_segments_file_read": (C.c_int, [u8p, C.c_int64, C.c_int64, i32p, C.c_int32, C.c_int32, u8p, C.c_int64, i64p, u8p, i64p, i32p]),
        "orc_field_infos_write": (C.c_int, [C.c_int32, i32p, i64p, u8p, u8p, C.c_char_p, u8p, i64p]),
        "orc_field_infos_read": (C.c_int, [u8p, C.c_int64, C.c_int32, i32p, i64p, u8p, C.c_int64, i64p]),
        "orc_fst_build": (C.c_int, [u8p, i64p, u8p, i64p, C.c_int64, C.c_int, u8p, i64p]),
        "orc_fst_get": (C.c_int, [u8p, C.c_int64, u8p, C.c_int32, u8p, C.c_int32]),
        "orc_fst_enumerate": (C.c_int64, [u8p, C.c_int64, u8p, C.c_int64, i64p]),
        "orc_fst_reverse_read": (C.c_int, [u8p, C.c_int64, C.c_int64, C.c_int32, u8p, C.c_int32]),
        "orc_blocktree_write": (C.c_int, [C.c_int32, i32p, i32p, u8p, i32p, i64p, u8p, i64p, vp, C.c_int32, C.c_int32, u8p,
                                          C.c_char_p, u8p, i64p, u8p, i64p]),
        "orc_blocktree_open": (vp, [u8p, C.c_int64, u8p, C.c_int64, C.c_int32, i32p, i32p, u8p, C.c_int32]),
        "orc_blocktree_close": (None, [vp]),
        "orc_blocktree_field_stats": (C.c_int, [vp, C.c_int32, i64p]),
        "orc_blocktree_seek_exact": (C.c_int, [vp, C.c_int32, u8p, i64p, C.c_int64, vp, u8p]),
        "orc_mock_conjunction": (C.c_int, [i32p, i32p, C.c_int, C.c_int32, i32p, f32p, C.c_int]),
        "orc_mock_conjunction_initial_score": (C.c_float, [i32p, i32p, C.c_int]),
        "orc_mock_disjunction": (C.c_int, [i32p, i32p, C.c_int, C.c_int, i32p, f32p, C.c_int]),
        "orc_mock_topk": (C.c_int, [i32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i32p, f32p, i64p]),
        "orc_topk_stream": (C.c_int, [i32p, f32p, C.c_int64, C.c_int, C.c_int, i32p, f32p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc < 0:
        raise OracleError("oracle error %d: %s" % (rc, lib().orc_last_error().decode()))
    return rc


# ---- packed helpers ------------------------------------------------------------------------------------------------
def bp128_pack(values, bits):
    v = np.ascontiguousarray(values, dtype=np.uint32)
    assert v.size == 128
    out = np.zeros(512, dtype=np.uint8)
    _check(lib().orc_bp128_pack(_p(v, C.c_uint32), _p(out, C.c_uint8), bits))
    return out[:16 * bits].copy()


def bp128_unpack(encoded, bits):
    enc = np.zeros(512 + 16, dtype=np.uint8)
    enc[:len(encoded)] = encoded
    out = np.zeros(128, dtype=np.uint32)
    _check(lib().orc_bp128_unpack(_p(enc, C.c_uint8), _p(out, C.c_uint32), bits))
    return out


def bp128_delta_pack(values, base, bits):
    v = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.zeros(512, dtype=np.uint8)
    _check(lib().orc_bp128_delta_pack(_p(v, C.c_uint32), _p(out, C.c_uint8), base, bits))
    return out[:16 * bits].copy()


def bp128_delta_unpack(encoded, base, bits):
    enc = np.zeros(512 + 16, dtype=np.uint8)
    enc[:len(encoded)] = encoded
    out = np.zeros(128, dtype=np.uint32)
    _check(lib().orc_bp128_delta_unpack(_p(enc, C.c_uint8), _p(out, C.c_uint32), base, bits))
    return out


def legacy_counts(fmt, bpv):
    a, b = C.c_int(), C.c_int()
    lib().orc_legacy_counts(fmt, bpv, C.byref(a), C.byref(b))
    return a.value, b.value


def legacy_decode(fmt, bpv, blocks, iterations):
    bbc, bvc = legacy_counts(fmt, bpv)
    enc = np.zeros(max(len(blocks), iterations * bbc) + 8, dtype=np.uint8)
    enc[:len(blocks)] = blocks
    out = np.zeros(iterations * bvc + 8, dtype=np.int32)
    _check(lib().orc_legacy_decode(fmt, bpv, _p(enc, C.c_uint8), _p(out, C.c_int32), iterations))
    return out[:iterations * bvc]


def legacy_encode(fmt, bpv, values, iterations):
    bbc, bvc = legacy_counts(fmt, bpv)
    v = np.zeros(iterations * bvc + 8, dtype=np.int32)
    v[:len(values)] = values
    out = np.zeros(iterations * bbc + 8, dtype=np.uint8)
    _check(lib().orc_legacy_encode(fmt, bpv, _p(v, C.c_int32), _p(out, C.c_uint8), iterations))
    return out[:iterations * bbc]


def format_fastest(value_count, bpv, ratio=0.0):
    f, b = C.c_int(), C.c_int()
    lib().orc_format_fastest(value_count, bpv, ratio, C.byref(f), C.byref(b))
    return f.value, b.value


# ---- writer ----------------------------------------------------------------------------------------------------------
class Writer:
    """Line-faithful Lucene50PostingsWriter (docs+freqs). write_term(docs, freqs) -> term-state record."""

    def __init__(self, max_doc, version=1, write_freqs=True, segment_id=None, suffix="Lucene50_0", use_ef=False, with_pf=True):
        sid = np.frombuffer(segment_id or bytes(range(16)), dtype=np.uint8).copy()
        self._h = lib().orc_writer_new(max_doc, version, int(write_freqs), _p(sid, C.c_uint8), suffix.encode())
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())
        if use_ef:  # EfWriterMeta.use_ef: Elias-Fano / bitset doc blocks where they are smaller (never on in Rucene itself)
            lib().orc_writer_set_ef(self._h, 1, int(with_pf))
        self.write_freqs = write_freqs

    def write_term(self, docs, freqs):
        docs = np.ascontiguousarray(docs, dtype=np.int32)
        freqs = np.ascontiguousarray(freqs, dtype=np.int32)
        L = lib()
        _check(L.orc_writer_start_term(self._h))
        _check(L.orc_writer_add_docs(self._h, _p(docs, C.c_int32), _p(freqs, C.c_int32) if self.write_freqs else None, docs.size))
        st = np.zeros(1, dtype=TERM_STATE_DTYPE)
        _check(L.orc_writer_finish_term(self._h, docs.size, int(freqs.sum()), st.ctypes.data))
        return st[0]

    def close(self):
        n = _check(lib().orc_writer_close(self._h))
        out = np.zeros(n, dtype=np.uint8)
        lib().orc_writer_copy(self._h, _p(out, C.c_uint8))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_writer_free(self._h)
            self._h = None


# ---- segment / searcher --------------------------------------------------------------------------------------------
class Segment:
    def __init__(self, doc_bytes, norms, max_doc, terms, doc_base=0, live_docs=None, doc_count=None,
                 sum_total_term_freq=0, sum_doc_freq=0, has_freqs=True):
        self.doc_bytes = np.ascontiguousarray(doc_bytes, dtype=np.uint8)
        self.norms = None if norms is None else np.ascontiguousarray(norms, dtype=np.uint8)
        self.live_docs = None if live_docs is None else np.ascontiguousarray(live_docs, dtype=np.uint64)
        self.terms = np.ascontiguousarray(terms, dtype=TERM_STATE_DTYPE)
        self.max_doc, self.doc_base = int(max_doc), int(doc_base)
        self.doc_count = int(max_doc if doc_count is None else doc_count)
        self.sum_total_term_freq = int(sum_total_term_freq)
        self._h = lib().orc_segment_new(
            _p(self.doc_bytes, C.c_uint8), self.doc_bytes.size, _p(self.norms, C.c_uint8), _p(self.live_docs, C.c_uint64),
            self.max_doc, self.doc_base, self.doc_count, self.sum_total_term_freq, int(sum_doc_freq),
            self.terms.ctypes.data, self.terms.size)
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())
        if not has_freqs:  # the field was indexed with IndexOptions::Docs
            lib().orc_segment_set_index_has_freq(self._h, 0)

    @property
    def version(self):
        return lib().orc_segment_version(self._h)

    def decode_term(self, state):
        st = np.array([state], dtype=TERM_STATE_DTYPE)
        n = int(st[0]["doc_freq"])
        docs = np.zeros(n + 1, dtype=np.int32)
        freqs = np.zeros(n + 1, dtype=np.int32)
        got = _check(lib().orc_decode_term(self._h, st.ctypes.data, _p(docs, C.c_int32), _p(freqs, C.c_int32)))
        assert got == n, (got, n)
        return docs[:n], freqs[:n]

    def decode_terms(self, states, threads=1):
        sts = np.ascontiguousarray(states, dtype=TERM_STATE_DTYPE)
        total = int(sts["doc_freq"].sum())
        docs = np.zeros(total + 1, dtype=np.int32)
        freqs = np.zeros(total + 1, dtype=np.int32)
        secs = lib().orc_decode_terms(self._h, sts.ctypes.data, sts.size, _p(docs, C.c_int32), _p(freqs, C.c_int32), threads)
        return docs[:total], freqs[:total], secs

    def postings(self, state, flags=FLAG_FREQS):
        return Postings(self, state, flags)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_segment_free(self._h)
            self._h = None


class Postings:
    def __init__(self, seg, state, flags):
        self._seg = seg
        self._st = np.array([state], dtype=TERM_STATE_DTYPE)
        self._h = lib().orc_postings_new(seg._h, self._st.ctypes.data, flags)

    def next(self):
        d = C.c_int32()
        _check(lib().orc_postings_next(self._h, C.byref(d)))
        return d.value

    def advance(self, target):
        d = C.c_int32()
        _check(lib().orc_postings_advance(self._h, target, C.byref(d)))
        return d.value

    def freq(self):
        return lib().orc_postings_freq(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_postings_free(self._h)
            self._h = None


class Searcher:
    def __init__(self, segments, k1=1.2, b=0.75):
        self.segments = list(segments)
        arr = (C.c_void_p * len(self.segments))(*[s._h for s in self.segments])
        self._h = lib().orc_searcher_new(arr, len(self.segments), k1, b)
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())

    def rescore(self, op, term_ids, docs, scores, window_size, query_weight, rescore_weight, mode):
        """QueryRescorer::rescore (rescorer.rs:376-390) of one first-pass row (best first) with a TERM / AND / OR term query;
        mode: 0 Avg, 1 Max, 2 Min, 3 Total, 4 Multiply. Returns the new (docs, scores)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int64)
        d = np.ascontiguousarray(docs, dtype=np.int32).copy()
        sc = np.ascontiguousarray(scores, dtype=np.float32).copy()
        _check(lib().orc_searcher_rescore(self._h, op, _p(t, C.c_int64), t.size, _p(d, C.c_int32), _p(sc, C.c_float), d.size, int(window_size),
                                          float(query_weight), float(rescore_weight), int(mode)))
        return d, sc

    def score_docs(self, op, term_ids, docs, min_should_match=0):
        """The oracle's own score of the given docs (global ids, any order) under a TERM / AND / OR term query: per leaf the
        query's scorer advanced from doc to doc (IndexSearcher::score_docs). Returns (scores, matched) in the order given."""
        t = np.ascontiguousarray(term_ids, dtype=np.int64)
        d = np.ascontiguousarray(docs, dtype=np.int32)
        order = np.argsort(d, kind="stable")
        ds = np.ascontiguousarray(d[order])
        sc = np.zeros(ds.size, dtype=np.float32)
        mt = np.zeros(ds.size, dtype=np.uint8)
        if ds.size:
            _check(lib().orc_searcher_score_docs(self._h, op, _p(t, C.c_int64), t.size, int(min_should_match), _p(ds, C.c_int32), ds.size,
                                                 _p(sc, C.c_float), _p(mt, C.c_uint8)))
        scores = np.zeros(ds.size, dtype=np.float32)
        matched = np.zeros(ds.size, dtype=bool)
        scores[order] = sc
        matched[order] = mt != 0
        return scores, matched

    def override_statistics(self, stats_segment, total_max_doc):
        """Score with another leaf's statistics (the index-wide largest leaf living on another shard)."""
        self._stats_segment = stats_segment  # keep alive
        lib().orc_searcher_override_stats(self._h, stats_segment._h, int(total_max_doc))

    def term_weight(self, term_id, boost=1.0):
        cache = np.zeros(256, dtype=np.float32)
        w = lib().orc_searcher_term_weight(self._h, term_id, boost, _p(cache, C.c_float))
        return w, cache

    def search(self, op, term_ids, k, tie_mode=TIE_CANONICAL, boosts=None, min_should_match=0, max_collect_per_leaf=0):
        t = np.ascontiguousarray(term_ids, dtype=np.int64)
        b = None if boosts is None else np.ascontiguousarray(boosts, dtype=np.float32)
        docs = np.zeros(max(k, 1), dtype=np.int32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        n, total = C.c_int32(), C.c_int64()
        _check(lib().orc_search(self._h, op, _p(t, C.c_int64), t.size, _p(b, C.c_float), min_should_match, k, tie_mode,
                                max_collect_per_leaf, _p(docs, C.c_int32), _p(scores, C.c_float), C.byref(n), C.byref(total)))
        return docs[:n.value].copy(), scores[:n.value].copy(), total.value

    def search_not(self, op, term_ids, must_not_ids, k, tie_mode=TIE_CANONICAL):
        """BooleanQuery with MUST_NOT TermQuery clauses (ReqNotScorer)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int64)
        nn = np.ascontiguousarray(must_not_ids, dtype=np.int64)
        docs = np.zeros(max(k, 1), dtype=np.int32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        n, total = C.c_int32(), C.c_int64()
        _check(lib().orc_search_not(self._h, op, _p(t, C.c_int64), t.size, _p(nn, C.c_int64), nn.size, k, tie_mode,
                                    _p(docs, C.c_int32), _p(scores, C.c_float), C.byref(n), C.byref(total)))
        return docs[:n.value].copy(), scores[:n.value].copy(), total.value

    def search_opt(self, op, term_ids, should_ids, k, must_not_ids=(), min_should_match=0, tie_mode=TIE_CANONICAL, exact=False):
        """BooleanQuery with MUST (op AND / TERM) + SHOULD [+ MUST_NOT] TermQuery clauses: ReqOptScorer, with its sequential
        skip-the-optional-clause rule (req_opt_scorer.rs:41-66); exact=True turns that rule off (full sums everywhere)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int64)
        oo = np.ascontiguousarray(should_ids, dtype=np.int64)
        nn = np.ascontiguousarray(must_not_ids, dtype=np.int64)
        docs = np.zeros(max(k, 1), dtype=np.int32)
        scores = np.zeros(max(k, 1), dtype=np.float32)
        n, total = C.c_int32(), C.c_int64()
        _check(lib().orc_search_opt(self._h, op, _p(t, C.c_int64), t.size, _p(oo, C.c_int64), oo.size,
                                    _p(nn, C.c_int64) if nn.size else None, nn.size, min_should_match, int(exact), k, tie_mode,
                                    _p(docs, C.c_int32), _p(scores, C.c_float), C.byref(n), C.byref(total)))
        return docs[:n.value].copy(), scores[:n.value].copy(), total.value

    def search_batch(self, ops, term_offsets, term_ids, k, tie_mode=TIE_CANONICAL, threads=1, not_offsets=None, not_ids=None,
                     min_should_match=None):
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        offs = np.ascontiguousarray(term_offsets, dtype=np.int32)
        tids = np.ascontiguousarray(term_ids, dtype=np.int64)
        nq = ops.size
        if not_offsets is not None or min_should_match is not None:
            if not_offsets is None:
                not_offsets, not_ids = np.zeros(nq + 1, np.int32), np.zeros(0, np.int64)
            noffs = np.ascontiguousarray(not_offsets, dtype=np.int32)
            nids = np.ascontiguousarray(not_ids, dtype=np.int64)
            msms = None if min_should_match is None else np.ascontiguousarray(min_should_match, dtype=np.int32)
            docs = np.full((nq, k), -1, dtype=np.int32)
            scores = np.zeros((nq, k), dtype=np.float32)
            counts = np.zeros(nq, dtype=np.int32)
            totals = np.zeros(nq, dtype=np.int64)
            visited = np.zeros(nq, dtype=np.uint64)
            secs = lib().orc_search_batch_not(self._h, nq, _p(ops, C.c_int32), _p(offs, C.c_int32), _p(tids, C.c_int64),
                                              _p(noffs, C.c_int32), _p(nids, C.c_int64), _p(msms, C.c_int32), k, tie_mode, threads,
                                              _p(docs, C.c_int32),
                                              _p(scores, C.c_float), _p(counts, C.c_int32), _p(totals, C.c_int64),
                                              _p(visited, C.c_uint64))
            if secs < 0:
                raise OracleError(lib().orc_last_error().decode())
            return docs, scores, counts, totals, visited, secs
        docs = np.full((nq, k), -1, dtype=np.int32)
        scores = np.zeros((nq, k), dtype=np.float32)
        counts = np.zeros(nq, dtype=np.int32)
        totals = np.zeros(nq, dtype=np.int64)
        visited = np.zeros(nq, dtype=np.uint64)
        secs = lib().orc_search_batch(self._h, nq, _p(ops, C.c_int32), _p(offs, C.c_int32), _p(tids, C.c_int64), k, tie_mode,
                                      threads, _p(docs, C.c_int32), _p(scores, C.c_float), _p(counts, C.c_int32),
                                      _p(totals, C.c_int64), _p(visited, C.c_uint64))
        if secs < 0:
            raise OracleError(lib().orc_last_error().decode())
        return docs, scores, counts, totals, visited, secs

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_searcher_free(self._h)
            self._h = None


# ---- mock KAT probes -----------------------------------------------------------------------------------------------
def _lists(lists):
    flat = np.concatenate([np.asarray(l, dtype=np.int32) for l in lists]) if lists else np.zeros(0, np.int32)
    offs = np.zeros(len(lists) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    return np.ascontiguousarray(flat), offs


def mock_conjunction(lists, advance_first=-1):
    flat, offs = _lists(lists)
    docs = np.zeros(flat.size + 1, np.int32)
    scores = np.zeros(flat.size + 1, np.float32)
    n = _check(lib().orc_mock_conjunction(_p(flat, C.c_int32), _p(offs, C.c_int32), len(lists), advance_first,
                                          _p(docs, C.c_int32), _p(scores, C.c_float), docs.size))
    return docs[:n].tolist(), scores[:n].tolist()


def mock_conjunction_initial_score(lists):
    flat, offs = _lists(lists)
    return lib().orc_mock_conjunction_initial_score(_p(flat, C.c_int32), _p(offs, C.c_int32), len(lists))


def mock_disjunction(lists, min_should_match=1):
    flat, offs = _lists(lists)
    docs = np.zeros(flat.size + 1, np.int32)
    scores = np.zeros(flat.size + 1, np.float32)
    n = _check(lib().orc_mock_disjunction(_p(flat, C.c_int32), _p(offs, C.c_int32), len(lists), min_should_match,
                                          _p(docs, C.c_int32), _p(scores, C.c_float), docs.size))
    return docs[:n].tolist(), scores[:n].tolist()


def norms_write(values, field_number=0, segment_id=None, suffix=""):
    """Lucene53NormsConsumer for one field: i64 norm values (one per doc) -> (nvm bytes, nvd bytes)."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    sid = np.frombuffer(segment_id if segment_id is not None else bytes(range(16)), dtype=np.uint8).copy()
    ml, dl = C.c_int64(0), C.c_int64(0)
    _check(lib().orc_norms_write(_p(v, C.c_int64), v.size, field_number, _p(sid, C.c_uint8), suffix.encode(), None, C.byref(ml),
                                 None, C.byref(dl)))
    m = np.zeros(ml.value, dtype=np.uint8)
    d = np.zeros(dl.value, dtype=np.uint8)
    _check(lib().orc_norms_write(_p(v, C.c_int64), v.size, field_number, _p(sid, C.c_uint8), suffix.encode(), _p(m, C.c_uint8),
                                 C.byref(ml), _p(d, C.c_uint8), C.byref(dl)))
    return m.tobytes(), d.tobytes()


def norms_read(nvm, nvd, field_number, max_doc):
    """Lucene53NormsProducer: norms(field).get(doc) for every doc."""
    m = np.frombuffer(nvm, dtype=np.uint8).copy()
    d = np.frombuffer(nvd, dtype=np.uint8).copy()
    out = np.zeros(max_doc, dtype=np.int64)
    _check(lib().orc_norms_read(_p(m, C.c_uint8), m.size, _p(d, C.c_uint8), d.size, field_number, max_doc, _p(out, C.c_int64)))
    return out


def live_docs_write(words, max_doc, del_count, segment_id=None, gen=1):
    """Lucene50LiveDocsFormat::write_live_docs: FixedBitSet words -> ".liv" bytes."""
    w = np.ascontiguousarray(words).view(np.int64)
    sid = np.frombuffer(segment_id if segment_id is not None else bytes(range(16)), dtype=np.uint8).copy()
    n = C.c_int64(0)
    _check(lib().orc_live_docs_write(_p(w, C.c_int64), w.size, max_doc, del_count, _p(sid, C.c_uint8), gen, None, C.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    _check(lib().orc_live_docs_write(_p(w, C.c_int64), w.size, max_doc, del_count, _p(sid, C.c_uint8), gen, _p(out, C.c_uint8), C.byref(n)))
    return out.tobytes()


def live_docs_read(liv, max_doc, del_count):
    b = np.frombuffer(liv, dtype=np.uint8).copy()
    out = np.zeros((max_doc + 63) // 64, dtype=np.int64)
    _check(lib().orc_live_docs_read(_p(b, C.c_uint8), b.size, max_doc, del_count, _p(out, C.c_int64)))
    return out.view(np.uint64)


def mock_req_opt(req_lists, opt_lists):
    """ReqOptScorer(Conjunction(req_lists) | the single list, DisjunctionSumScorer(opt_lists)): next() to exhaustion
    -> (docs, scores)."""
    rf, ro = _lists(req_lists)
    of, oo = _lists(opt_lists)
    docs = np.zeros(max(1, rf.size), dtype=np.int32)
    scores = np.zeros(max(1, rf.size), dtype=np.float32)
    n = _check(lib().orc_mock_req_opt(_p(rf, C.c_int32), _p(ro, C.c_int32), len(req_lists), _p(of, C.c_int32), _p(oo, C.c_int32),
                                      len(opt_lists), _p(docs, C.c_int32), _p(scores, C.c_float), docs.size))
    return docs[:n].tolist(), scores[:n].tolist()


def mock_req_not(req_lists, not_lists, targets=()):
    """ReqNotScorer(Conjunction(req_lists) | the single list, Disjunction(not_lists) | the single list):
    next() to exhaustion, or advance(t) for each target."""
    rf, ro = _lists(req_lists)
    nf, no = _lists(not_lists)
    tg = np.ascontiguousarray(targets, dtype=np.int32)
    out = np.zeros(max(1, rf.size + tg.size), dtype=np.int32)
    n = _check(lib().orc_mock_req_not(_p(rf, C.c_int32), _p(ro, C.c_int32), len(req_lists), _p(nf, C.c_int32), _p(no, C.c_int32),
                                      len(not_lists), _p(tg, C.c_int32), tg.size, _p(out, C.c_int32), out.size))
    return out[:n].copy()


def mock_topk(docs, k, n_leaves=1, max_collect_per_leaf=0, tie_mode=TIE_RUST_HEAP, use_bulk_scorer=False):
    d = np.ascontiguousarray(docs, dtype=np.int32)
    od = np.zeros(k + 1, np.int32)
    os_ = np.zeros(k + 1, np.float32)
    total = C.c_int64()
    n = _check(lib().orc_mock_topk(_p(d, C.c_int32), d.size, n_leaves, max_collect_per_leaf, k, tie_mode, int(use_bulk_scorer),
                                   _p(od, C.c_int32), _p(os_, C.c_float), C.byref(total)))
    return od[:n].tolist(), os_[:n].tolist(), total.value


def topk_stream(docs, scores, k, tie_mode):
    d = np.ascontiguousarray(docs, dtype=np.int32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    od = np.zeros(k + 1, np.int32)
    os_ = np.zeros(k + 1, np.float32)
    n = _check(lib().orc_topk_stream(_p(d, C.c_int32), _p(s, C.c_float), d.size, k, tie_mode, _p(od, C.c_int32), _p(os_, C.c_float)))
    return od[:n].copy(), os_[:n].copy()


# ---- FST + block-tree term dictionary --------------------------------------------------------------------------------

# FullTermState (oracle/blocktree.hpp): the shared 32-byte term state + the position/payload pointers
FULL_TERM_STATE_DTYPE = np.dtype(
    [("base", TERM_STATE_DTYPE), ("pos_start_fp", "<i8"), ("pay_start_fp", "<i8"), ("last_pos_block_offset", "<i8")], align=True)
assert FULL_TERM_STATE_DTYPE.itemsize == 56
IO_DOCS, IO_DOCS_FREQS, IO_DOCS_FREQS_POS, IO_DOCS_FREQS_POS_OFFS = 1, 2, 3, 4


def _flatten_bytes(items):
    offs = np.zeros(len(items) + 1, dtype=np.int64)
    for i, b in enumerate(items):
        offs[i + 1] = offs[i] + len(b)
    flat = np.frombuffer(b"".join(items), dtype=np.uint8).copy() if offs[-1] else np.zeros(1, dtype=np.uint8)
    return flat, offs


def fst_build(pairs, share_non_singleton=True):
    """FstBuilder over sorted (input bytes, output bytes) pairs -> saved FST bytes (b"" when the builder yields None)."""
    fi, oi = _flatten_bytes([p[0] for p in pairs])
    fo, oo = _flatten_bytes([p[1] for p in pairs])
    n = C.c_int64(0)
    _check(lib().orc_fst_build(_p(fi, C.c_uint8), _p(oi, C.c_int64), _p(fo, C.c_uint8), _p(oo, C.c_int64), len(pairs),
                               int(share_non_singleton), None, C.byref(n)))
    out = np.zeros(max(1, n.value), dtype=np.uint8)
    _check(lib().orc_fst_build(_p(fi, C.c_uint8), _p(oi, C.c_int64), _p(fo, C.c_uint8), _p(oo, C.c_int64), len(pairs),
                               int(share_non_singleton), _p(out, C.c_uint8), C.byref(n)))
    return out[:n.value].tobytes()


def fst_get(fst_bytes, key):
    """FST::get -> output bytes, or None when `key` is not accepted."""
    b = np.frombuffer(fst_bytes, dtype=np.uint8).copy()
    k = np.frombuffer(key, dtype=np.uint8).copy() if key else np.zeros(1, dtype=np.uint8)
    out = np.zeros(4096, dtype=np.uint8)
    r = lib().orc_fst_get(_p(b, C.c_uint8), b.size, _p(k, C.c_uint8), len(key), _p(out, C.c_uint8), out.size)
    if r == -1000:
        return None
    return out[:_check(r)].tobytes()


def fst_enumerate(fst_bytes):
    """BytesRefFSTIterator: [(input, output)] in input byte order (inputs/outputs < 256 bytes)."""
    b = np.frombuffer(fst_bytes, dtype=np.uint8).copy()
    n = C.c_int64(0)
    _check(lib().orc_fst_enumerate(_p(b, C.c_uint8), b.size, None, 0, C.byref(n)))
    flat = np.zeros(max(1, n.value), dtype=np.uint8)
    count = _check(lib().orc_fst_enumerate(_p(b, C.c_uint8), b.size, _p(flat, C.c_uint8), flat.size, C.byref(n)))
    raw, pos, out = flat.tobytes(), 0, []
    for _ in range(count):
        li = raw[pos]; inp = raw[pos + 1:pos + 1 + li]; pos += 1 + li
        lo = raw[pos]; outp = raw[pos + 1:pos + 1 + lo]; pos += 1 + lo
        out.append((inp, outp))
    return out


def fst_reverse_read(data, pos, skip_after_first, n):
    """Reverse bytes reader: one byte at `pos`, skip, then n-1 more; -> (bytes read, final position)."""
    b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.uint8)
    end = _check(lib().orc_fst_reverse_read(_p(b, C.c_uint8), b.size, pos, skip_after_first, _p(out, C.c_uint8), n))
    return out.tolist(), end - 1


def blocktree_write(fields, min_items=25, max_items=48, segment_id=None, suffix=""):
    """BlockTreeTermsWriter. fields = [{number, index_options, has_payloads, doc_count, terms: [bytes] sorted,
    states: FULL_TERM_STATE_DTYPE[len(terms)]}] in field-name order -> (tim bytes, tip bytes)."""
    numbers = np.array([f["number"] for f in fields], dtype=np.int32)
    opts = np.array([f.get("index_options", IO_DOCS_FREQS) for f in fields], dtype=np.int32)
    pay = np.array([1 if f.get("has_payloads") else 0 for f in fields], dtype=np.uint8)
    dcs = np.array([f["doc_count"] for f in fields], dtype=np.int32)
    foffs = np.zeros(len(fields) + 1, dtype=np.int64)
    all_terms, all_states = [], []
    for i, f in enumerate(fields):
        foffs[i + 1] = foffs[i] + len(f["terms"])
        all_terms += list(f["terms"])
        all_states.append(np.ascontiguousarray(f["states"], dtype=FULL_TERM_STATE_DTYPE))
    flat, toffs = _flatten_bytes(all_terms)
    states = np.concatenate(all_states) if all_states else np.zeros(0, dtype=FULL_TERM_STATE_DTYPE)
    sid = np.frombuffer(segment_id if segment_id is not None else bytes(range(16)), dtype=np.uint8).copy()
    tl, il = C.c_int64(0), C.c_int64(0)
    args = (len(fields), _p(numbers, C.c_int32), _p(opts, C.c_int32), _p(pay, C.c_uint8), _p(dcs, C.c_int32), _p(foffs, C.c_int64),
            _p(flat, C.c_uint8), _p(toffs, C.c_int64), states.ctypes.data_as(C.c_void_p), min_items, max_items, _p(sid, C.c_uint8),
            suffix.encode())
    _check(lib().orc_blocktree_write(*args, None, C.byref(tl), None, C.byref(il)))
    tim = np.zeros(tl.value, dtype=np.uint8)
    tip = np.zeros(il.value, dtype=np.uint8)
    _check(lib().orc_blocktree_write(*args, _p(tim, C.c_uint8), C.byref(tl), _p(tip, C.c_uint8), C.byref(il)))
    return tim.tobytes(), tip.tobytes()


class BlockTreeReader:
    """BlockTreeTermsReader + SegmentTermIterator::seek_exact / term_state."""

    def __init__(self, tim, tip, field_infos, max_doc):
        self._tim = np.frombuffer(tim, dtype=np.uint8).copy()
        self._tip = np.frombuffer(tip, dtype=np.uint8).copy()
        numbers = np.array([f["number"] for f in field_infos], dtype=np.int32)
        opts = np.array([f.get("index_options", IO_DOCS_FREQS) for f in field_infos], dtype=np.int32)
        pay = np.array([1 if f.get("has_payloads") else 0 for f in field_infos], dtype=np.uint8)
        self._h = lib().orc_blocktree_open(_p(self._tim, C.c_uint8), self._tim.size, _p(self._tip, C.c_uint8), self._tip.size,
                                           len(field_infos), _p(numbers, C.c_int32), _p(opts, C.c_int32), _p(pay, C.c_uint8), max_doc)
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())

    def field_stats(self, field):
        out = np.zeros(6, dtype=np.int64)
        r = _check(lib().orc_blocktree_field_stats(self._h, field, _p(out, C.c_int64)))
        if r == 1:
            return None
        return dict(zip(("num_terms", "sum_total_term_freq", "sum_doc_freq", "doc_count", "longs_size", "root_block_fp"),
                        out.tolist()))

    def seek_exact(self, field, terms):
        flat, offs = _flatten_bytes(list(terms))
        states = np.zeros(len(terms), dtype=FULL_TERM_STATE_DTYPE)
        found = np.zeros(max(1, len(terms)), dtype=np.uint8)
        _check(lib().orc_blocktree_seek_exact(self._h, field, _p(flat, C.c_uint8), _p(offs, C.c_int64), len(terms),
                                              states.ctypes.data_as(C.c_void_p), _p(found, C.c_uint8)))
        return states, found[:len(terms)].astype(bool)

    def close(self):
        if self._h:
            lib().orc_blocktree_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


# ---- Lucene60 field infos (".fnm") ----------------------------------------------------------------------------------
_FI_DEFAULTS = dict(index_options=0, doc_values_type=0, store_term_vector=False, omit_norms=False, store_payloads=False,
                    dv_gen=-1, attributes=None, point_dimension_count=0, point_num_bytes=0)


def _lp(s):
    b = s.encode("utf-8")
    return len(b).to_bytes(4, "little") + b


def field_infos_write(fields, segment_id=None, suffix=""):
    """Lucene60FieldInfosFormat::write. fields: [dict(name, number, index_options, ...)] (see _FI_DEFAULTS) -> ".fnm" bytes."""
    fields = [dict(_FI_DEFAULTS, **f) for f in fields]
    recs = np.zeros((max(len(fields), 1), 6), dtype=np.int32)
    gens = np.zeros(max(len(fields), 1), dtype=np.int64)
    flat = b""
    for i, f in enumerate(fields):
        bits = (1 if f["store_term_vector"] else 0) | (2 if f["omit_norms"] else 0) | (4 if f["store_payloads"] else 0)
        recs[i] = (f["number"], f["index_options"], f["doc_values_type"], bits, f["point_dimension_count"], f["point_num_bytes"])
        gens[i] = f["dv_gen"]
        attrs = f["attributes"] or {}
        flat += _lp(f["name"]) + len(attrs).to_bytes(4, "little") + b"".join(_lp(k) + _lp(v) for k, v in attrs.items())
    strings = np.frombuffer(flat + b"\0", dtype=np.uint8).copy()
    sid = np.frombuffer(segment_id if segment_id is not None else bytes(range(16)), dtype=np.uint8).copy()
    n = C.c_int64(0)
    args = (len(fields), _p(recs, C.c_int32), _p(gens, C.c_int64), _p(strings, C.c_uint8), _p(sid, C.c_uint8), suffix.encode())
    _check(lib().orc_field_infos_write(*args, None, C.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    _check(lib().orc_field_infos_write(*args, _p(out, C.c_uint8), C.byref(n)))
    return out.tobytes()


def field_infos_read(fnm):
    """Lucene60FieldInfosFormat::read -> [dict] with every FieldInfo member."""
    b = np.frombuffer(fnm, dtype=np.uint8).copy()
    n = C.c_int64(0)
    count = _check(lib().orc_field_infos_read(_p(b, C.c_uint8), b.size, 0, None, None, None, 0, C.byref(n)))
    recs = np.zeros((max(count, 1), 6), dtype=np.int32)
    gens = np.zeros(max(count, 1), dtype=np.int64)
    buf = np.zeros(max(n.value, 1), dtype=np.uint8)
    _check(lib().orc_field_infos_read(_p(b, C.c_uint8), b.size, count, _p(recs, C.c_int32), _p(gens, C.c_int64), _p(buf, C.c_uint8),
                                      buf.size, C.byref(n)))
    raw, pos, out = buf.tobytes(), 0, []

    def take():
        nonlocal pos
        ln = int.from_bytes(raw[pos:pos + 4], "little")
        s = raw[pos + 4:pos + 4 + ln].decode("utf-8")
        pos += 4 + ln
        return s
    for i in range(count):
        name = take()
        n_attr = int.from_bytes(raw[pos:pos + 4], "little")
        pos += 4
        attrs = {}
        for _ in range(n_attr):
            k = take()
            attrs[k] = take()
        out.append(dict(name=name, number=int(recs[i, 0]), index_options=int(recs[i, 1]), doc_values_type=int(recs[i, 2]),
                        store_term_vector=bool(recs[i, 3] & 1), omit_norms=bool(recs[i, 3] & 2), store_payloads=bool(recs[i, 3] & 4),
                        dv_gen=int(gens[i]), attributes=attrs, point_dimension_count=int(recs[i, 4]), point_num_bytes=int(recs[i, 5])))
    return out


# ---- ".si" and "segments_N" ------------------------------------------------------------------------------------------
def _u8(b):
    return np.frombuffer(bytes(b) + b"\0", dtype=np.uint8).copy()


def segment_info_write(name, max_doc, segment_id=None, files=(), is_compound_file=False, version=(6, 4, 18), diagnostics=None,
                       attributes=None):
    """Lucene62SegmentInfoFormat::write (no index sort) -> ".si" bytes."""
    diagnostics, attributes = diagnostics or {}, attributes or {}
    flat = (_lp(name) + len(diagnostics).to_bytes(4, "little") + b"".join(_lp(k) + _lp(v) for k, v in diagnostics.items()) +
            len(files).to_bytes(4, "little") + b"".join(_lp(f) for f in files) +
            len(attributes).to_bytes(4, "little") + b"".join(_lp(k) + _lp(v) for k, v in attributes.items()))
    sid = _u8(segment_id if segment_id is not None else bytes(range(16)))
    ver = np.asarray(version, dtype=np.int32)
    strings = _u8(flat)
    n = C.c_int64(0)
    args = (_p(strings, C.c_uint8), _p(sid, C.c_uint8), _p(ver, C.c_int32), int(max_doc), int(is_compound_file))
    _check(lib().orc_segment_info_write(*args, None, C.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    _check(lib().orc_segment_info_write(*args, _p(out, C.c_uint8), C.byref(n)))
    return out.tobytes()


def segment_info_read(si, expected_id=None):
    b = np.frombuffer(si, dtype=np.uint8).copy()
    out6, counts2, sid = np.zeros(6, np.int32), np.zeros(2, np.int32), np.zeros(16, np.uint8)
    eid = _u8(expected_id) if expected_id is not None else None
    _check(lib().orc_segment_info_read(_p(b, C.c_uint8), b.size, _p(eid, C.c_uint8), _p(out6, C.c_int32), _p(counts2, C.c_int32), _p(sid, C.c_uint8)))
    return dict(max_doc=int(out6[0]), is_compound_file=bool(out6[1]), version=tuple(int(v) for v in out6[2:5]), n_files=int(out6[5]),
                n_diagnostics=int(counts2[0]), n_attributes=int(counts2[1]), id=sid.tobytes())


def segments_file_write(segments, generation=1, commit_id=None, version=1, counter=None):
    """SegmentInfos::write_output. segments: [dict(name, id, max_doc, del_gen=-1, del_count=0, field_infos_gen=-1, dv_gen=-1,
    version=(6, 4, 18))] -> "segments_N" bytes."""
    n = len(segments)
    names = _u8(b"".join(_lp(s["name"]) for s in segments))
    ids = _u8(b"".join(bytes(s["id"]) for s in segments))
    longs = np.zeros((max(n, 1), 4), np.int64)
    ints = np.zeros((max(n, 1), 5), np.int32)
    for i, s in enumerate(segments):
        longs[i, :3] = (s.get("del_gen", -1), s.get("field_infos_gen", -1), s.get("dv_gen", -1))
        ints[i] = (s.get("del_count", 0), s["max_doc"]) + tuple(s.get("version", (6, 4, 18)))
    cid = _u8(commit_id if commit_id is not None else bytes(range(100, 116)))
    ln = C.c_int64(0)
    args = (int(generation), _p(cid, C.c_uint8), int(version), int(n if counter is None else counter), n, _p(names, C.c_uint8),
            _p(ids, C.c_uint8), _p(longs, C.c_int64), _p(ints, C.c_int32))
    _check(lib().orc_segments_file_write(*args, None, C.byref(ln)))
    out = np.zeros(ln.value, dtype=np.uint8)
    _check(lib().orc_segments_file_write(*args, _p(out, C.c_uint8), C.byref(ln)))
    return out.tobytes()


def segments_file_read(data, generation, max_docs=None):
    b = np.frombuffer(data, dtype=np.uint8).copy()
    md = np.asarray(max_docs, dtype=np.int32) if max_docs is not None else None
    ln = C.c_int64(0)
    count = _check(lib().orc_segments_file_read(_p(b, C.c_uint8), b.size, int(generation), _p(md, C.c_int32), 0 if md is None else md.size,
                                                0, None, 0, C.byref(ln), None, None, None))
    names = np.zeros(max(ln.value, 1), np.uint8)
    ids = np.zeros(max(count, 1) * 16, np.uint8)
    longs = np.zeros((max(count, 1), 3), np.int64)
    dels = np.zeros(max(count, 1), np.int32)
    _check(lib().orc_segments_file_read(_p(b, C.c_uint8), b.size, int(generation), _p(md, C.c_int32), 0 if md is None else md.size, count,
                                        _p(names, C.c_uint8), names.size, C.byref(ln), _p(ids, C.c_uint8), _p(longs, C.c_int64),
                                        _p(dels, C.c_int32)))
    raw, pos, out = names.tobytes(), 0, []
    for i in range(count):
        n = int.from_bytes(raw[pos:pos + 4], "little")
        out.append(dict(name=raw[pos + 4:pos + 4 + n].decode(), id=ids[16 * i:16 * i + 16].tobytes(), del_gen=int(longs[i, 0]),
                        field_infos_gen=int(longs[i, 1]), dv_gen=int(longs[i, 2]), del_count=int(dels[i])))
        pos += 4 + n
    return out


# ---- positions (".pos" + BlockPostingIterator) -------------------------------------------------------------------------
class PositionsIndex:
    """A docs+freqs+positions field written by the restated Lucene50PostingsWriter. postings: per term a list of
    (doc, [positions...]) in doc order. offsets / payloads: the field also stores them (IndexOptions::DocsAndFreqsAndPositionsAndOffsets
    / FieldInfo::has_store_payloads -> a third file, ".pay"); a doc's entry is then (doc, [positions], [(start, end), ...], [payload bytes, ...])
    — the two extra lists may be None / shorter tuples when their feature is off."""

    def __init__(self, max_doc, postings, version=1, offsets=False, payloads=False):
        docs, freqs, positions, doc_offs, pos_offs = [], [], [], [0], [0]
        starts, ends, pay_offs, pay_bytes = [], [], [0], bytearray()
        for plist in postings:
            for entry in plist:
                d, ps = entry[0], entry[1]
                docs.append(d)
                freqs.append(len(ps))
                positions.extend(ps)
                pos_offs.append(len(positions))
                if offsets:
                    assert len(entry[2]) == len(ps)
                    starts.extend(o[0] for o in entry[2])
                    ends.extend(o[1] for o in entry[2])
                if payloads:
                    assert len(entry[3]) == len(ps)
                    for b in entry[3]:
                        pay_bytes += bytes(b)
                        pay_offs.append(len(pay_bytes))
            doc_offs.append(len(docs))
        a = lambda v, t: np.ascontiguousarray(v if len(v) else [0], dtype=t)
        self.offsets, self.payloads = bool(offsets), bool(payloads)
        self._args = (a(doc_offs, np.int64), a(docs, np.int32), a(freqs, np.int32), a(pos_offs, np.int64), a(positions, np.int32),
                      a(starts, np.int32), a(ends, np.int32), a(pay_offs, np.int64), a(np.frombuffer(bytes(pay_bytes), np.uint8), np.uint8))
        self.n_terms = len(postings)
        self._h = lib().orc_pos_index_build_ex(int(max_doc), int(version), self.n_terms, _p(self._args[0], C.c_int64), _p(self._args[1], C.c_int32),
                                           _p(self._args[2], C.c_int32), _p(self._args[3], C.c_int64), _p(self._args[4], C.c_int32),
                                           (1 if offsets else 0) | (2 if payloads else 0), _p(self._args[5], C.c_int32), _p(self._args[6], C.c_int32),
                                           _p(self._args[7], C.c_int64), _p(self._args[8], C.c_uint8))
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())

    @classmethod
    def from_files(cls, doc_bytes, pos_bytes, states, positions, pay_bytes=None, offsets=False, payloads=False):
        """The readers over files somebody else wrote (the product's writer): states = rgpu_term_state records, positions =
        rgpu_term_positions records (numpy structured arrays as rucene_amd hands them out)."""
        self = cls.__new__(cls)
        n = len(states)
        s8 = np.zeros((n, 8), dtype=np.int64)
        for j, k in enumerate(("doc_start_fp", "skip_offset", "total_term_freq", "doc_freq", "singleton_doc_id")):
            s8[:, j] = states[k]
        s8[:, 5], s8[:, 6], s8[:, 7] = positions["pos_start_fp"], positions["last_pos_block_offset"], positions["pay_start_fp"]
        as_u8 = lambda b: np.ascontiguousarray(np.frombuffer(bytes(b), dtype=np.uint8) if isinstance(b, (bytes, bytearray)) else b, dtype=np.uint8)
        d, p = as_u8(doc_bytes), as_u8(pos_bytes)
        y = None if pay_bytes is None or len(pay_bytes) == 0 else as_u8(pay_bytes)
        self.offsets, self.payloads, self.n_terms = bool(offsets), bool(payloads), n
        self._args = (d, p, y, s8)
        self._h = lib().orc_pos_index_from_files(_p(d, C.c_uint8), d.size, _p(p, C.c_uint8), p.size, _p(y, C.c_uint8), 0 if y is None else y.size, n,
                                                 _p(s8, C.c_int64), (1 if offsets else 0) | (2 if payloads else 0))
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())
        return self

    def sizes(self):
        d, p = C.c_int64(0), C.c_int64(0)
        lib().orc_pos_index_sizes(self._h, C.byref(d), C.byref(p))
        return d.value, p.value

    def files(self):
        """-> (.doc bytes, .pos bytes)"""
        dl, pl = self.sizes()
        d, p = np.zeros(dl, np.uint8), np.zeros(pl, np.uint8)
        lib().orc_pos_index_copy(self._h, _p(d, C.c_uint8), _p(p, C.c_uint8))
        return d.tobytes(), p.tobytes()

    def pay_file(self):
        """-> the .pay bytes (b"" when the field stores neither payloads nor offsets)"""
        n = lib().orc_pos_index_pay(self._h, None)
        out = np.zeros(max(n, 1), np.uint8)
        lib().orc_pos_index_pay(self._h, _p(out, C.c_uint8))
        return out[:n].tobytes()

    def term_state(self, term):
        out = np.zeros(8, dtype=np.int64)
        _check(lib().orc_pos_term_state(self._h, term, _p(out, C.c_int64)))
        return dict(zip(("doc_start_fp", "skip_offset", "total_term_freq", "doc_freq", "singleton_doc_id", "pos_start_fp",
                         "last_pos_block_offset", "pay_start_fp"), out.tolist()))

    def iterate_everything(self, term, flags=0x78, targets=None, read_every=1, max_positions=-1, cap_visits=1 << 20, cap_positions=1 << 22,
                           cap_payload_bytes=1 << 24):
        """EverythingIterator (posting_reader.rs:1595-2337) -> [(doc, freq, [(position, start_offset, end_offset, payload bytes)])].
        flags: PostingIteratorFlags (0x58 PAYLOADS, 0x38 OFFSETS, 0x78 ALL)."""
        tg = None if targets is None else np.ascontiguousarray(targets, dtype=np.int32)
        docs, freqs, npos = (np.zeros(cap_visits, np.int32) for _ in range(3))
        pos, st, en, pl = (np.zeros(cap_positions, np.int32) for _ in range(4))
        pb = np.zeros(cap_payload_bytes, np.uint8)
        total = C.c_int64(0)
        n = _check(lib().orc_pos_iterate_everything(self._h, term, int(flags), _p(tg, C.c_int32), 0 if tg is None else tg.size, read_every,
                                                max_positions, _p(docs, C.c_int32), _p(freqs, C.c_int32), _p(npos, C.c_int32), cap_visits,
                                                _p(pos, C.c_int32), _p(st, C.c_int32), _p(en, C.c_int32), _p(pl, C.c_int32), cap_positions,
                                                _p(pb, C.c_uint8), cap_payload_bytes, C.byref(total)))
        out, at, bat = [], 0, 0
        raw = pb[:total.value].tobytes()
        for i in range(n):
            ps = []
            for j in range(at, at + int(npos[i])):
                ps.append((int(pos[j]), int(st[j]), int(en[j]), raw[bat:bat + int(pl[j])]))
                bat += int(pl[j])
            out.append((int(docs[i]), int(freqs[i]), ps))
            at += int(npos[i])
        return out

    def iterate(self, term, targets=None, read_every=1, max_positions=-1, cap_visits=1 << 20, cap_positions=1 << 22):
        """-> [(doc, freq, [positions read])]; targets None: next() to the end, else advance(t) for each t."""
        tg = None if targets is None else np.ascontiguousarray(targets, dtype=np.int32)
        docs = np.zeros(cap_visits, np.int32)
        freqs = np.zeros(cap_visits, np.int32)
        npos = np.zeros(cap_visits, np.int32)
        pos = np.zeros(cap_positions, np.int32)
        n = _check(lib().orc_pos_iterate(self._h, term, _p(tg, C.c_int32), 0 if tg is None else tg.size, read_every, max_positions,
                                         _p(docs, C.c_int32), _p(freqs, C.c_int32), _p(npos, C.c_int32), cap_visits, _p(pos, C.c_int32),
                                         cap_positions))
        out, at = [], 0
        for i in range(n):
            out.append((int(docs[i]), int(freqs[i]), pos[at:at + npos[i]].tolist()))
            at += int(npos[i])
        return out

    def phrase_freqs(self, term_ids, offsets=None, cap=1 << 20):
        """ExactPhraseScorer over the phrase term_ids (positions 0, 1, 2, ... unless `offsets`): [(doc, phrase freq)]."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        o = np.ascontiguousarray(range(t.size) if offsets is None else offsets, dtype=np.int32)
        docs, freqs = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = _check(lib().orc_pos_phrase_freqs(self._h, _p(t, C.c_int32), _p(o, C.c_int32), t.size, _p(docs, C.c_int32), _p(freqs, C.c_int32), cap))
        return list(zip(docs[:n].tolist(), freqs[:n].tolist()))

    def sloppy_freqs(self, term_ids, slop, offsets=None, cap=1 << 20):
        """SloppyPhraseScorer over the phrase term_ids (query order; positions 0, 1, 2, ... unless `offsets`): [(doc, sloppy freq f32)]."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        o = np.ascontiguousarray(range(t.size) if offsets is None else offsets, dtype=np.int32)
        docs, freqs = np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        n = _check(lib().orc_pos_sloppy_freqs(self._h, _p(t, C.c_int32), _p(o, C.c_int32), t.size, int(slop), _p(docs, C.c_int32), _p(freqs, C.c_float), cap))
        return docs[:n].copy(), freqs[:n].copy()

    def phrase_search(self, term_ids, k, norms, max_doc, doc_count, sum_total_term_freq, offsets=None, tie_mode=TIE_CANONICAL, slop=0,
                      live_docs=None, next_limit=None):
        """IndexSearcher::search(PhraseQuery(slop), TopDocsCollector(k)) -> (docs, scores, total_hits). live_docs: u64 words
        (FixedBitSet) or None; next_limit: DefaultIndexSearcher::new(reader, next_limit) — None = its default of 500 000 (only the
        two-phase sloppy scorer is subject to it)."""
        t = np.ascontiguousarray(term_ids, dtype=np.int32)
        o = np.ascontiguousarray(range(t.size) if offsets is None else offsets, dtype=np.int32)
        nm = None if norms is None else np.ascontiguousarray(norms, dtype=np.uint8)
        lv = None if live_docs is None else np.ascontiguousarray(live_docs, dtype=np.uint64)
        docs, scores = np.zeros(max(k, 1), np.int32), np.zeros(max(k, 1), np.float32)
        n, total = C.c_int32(), C.c_int64()
        _check(lib().orc_pos_phrase_search_ex(self._h, _p(t, C.c_int32), _p(o, C.c_int32), t.size, int(slop), _p(nm, C.c_uint8), int(max_doc),
                                              int(doc_count), int(sum_total_term_freq), k, tie_mode, None if lv is None else lv.ctypes.data,
                                              -1 if next_limit is None else int(next_limit), _p(docs, C.c_int32), _p(scores, C.c_float),
                                              C.byref(n), C.byref(total)))
        return docs[:n.value].copy(), scores[:n.value].copy(), total.value

    def close(self):
        if self._h:
            lib().orc_pos_index_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


# ---- compound files (".cfs" + ".cfe") ----------------------------------------------------------------------------------
def compound_write(files, segment_id):
    """Lucene50CompoundFormat::write. files: {file name: bytes} (every file must carry segment_id in its header) -> (cfs, cfe)."""
    names = sorted(files)
    flat_names = _u8(b"".join(_lp(n) for n in names))
    blob = _u8(b"".join(files[n] for n in names))
    offs = np.zeros(len(names) + 1, np.int64)
    np.cumsum([len(files[n]) for n in names], out=offs[1:])
    sid = _u8(segment_id)
    sl, el = C.c_int64(0), C.c_int64(0)
    args = (len(names), _p(flat_names, C.c_uint8), _p(blob, C.c_uint8), _p(offs, C.c_int64), _p(sid, C.c_uint8))
    _check(lib().orc_compound_write(*args, None, C.byref(sl), None, C.byref(el)))
    cfs, cfe = np.zeros(sl.value, np.uint8), np.zeros(el.value, np.uint8)
    _check(lib().orc_compound_write(*args, _p(cfs, C.c_uint8), C.byref(sl), _p(cfe, C.c_uint8), C.byref(el)))
    return cfs.tobytes(), cfe.tobytes()


def compound_read(cfe, cfs, expected_id=None):
    """Lucene50CompoundReader -> {entry id: (offset, length)}"""
    e, d = _u8(cfe)[:-1], _u8(cfs)[:-1]
    eid = _u8(expected_id) if expected_id is not None else None
    ln = C.c_int64(0)
    n = _check(lib().orc_compound_read(_p(e, C.c_uint8), e.size, _p(d, C.c_uint8), d.size, _p(eid, C.c_uint8), 0, None, 0, C.byref(ln), None, None))
    ids = np.zeros(max(ln.value, 1), np.uint8)
    offs, lens = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64)
    _check(lib().orc_compound_read(_p(e, C.c_uint8), e.size, _p(d, C.c_uint8), d.size, _p(eid, C.c_uint8), n, _p(ids, C.c_uint8), ids.size,
                                   C.byref(ln), _p(offs, C.c_int64), _p(lens, C.c_int64)))
    raw, pos, out = ids.tobytes(), 0, {}
    for i in range(n):
        k = int.from_bytes(raw[pos:pos + 4], "little")
        out[raw[pos + 4:pos + 4 + k].decode()] = (int(offs[i]), int(lens[i]))
        pos += 4 + k
    return out
