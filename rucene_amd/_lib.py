"""ctypes binding of include/rucene_gpu.h (librucene_gpu.so). No torch types cross this boundary: numpy arrays
for host buffers, raw integers for device pointers / hipStream_t handles."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "librucene_gpu.so"

ABI_VERSION = 6
OP_TERM, OP_AND, OP_OR = 0, 1, 2
OP_SHOULD_REQUIRED = 1 << 24   # RGPU_OP_SHOULD_REQUIRED: the optional SHOULD clauses are a nested disjunction under MUST
OP_NESTED_MUST = 1 << 25       # RGPU_OP_NESTED_MUST: ... are a nested conjunction under MUST (its own sum, formed first)
MAX_K = 1024
MAX_QUERY_TERMS = 64
MAX_PHRASE_TERMS = 16
NO_MORE_DOCS = 0x7FFFFFFF

TERM_STATE_DTYPE = np.dtype(
    [("doc_start_fp", "<i8"), ("skip_offset", "<i8"), ("total_term_freq", "<i8"), ("doc_freq", "<i4"),
     ("singleton_doc_id", "<i4")], align=True)
QUERY_TERM_DTYPE = np.dtype([("state", TERM_STATE_DTYPE), ("weight", "<f4"), ("sim_table", "<i4")], align=True)
QUERY_DTYPE = np.dtype([("op", "<i4"), ("n_terms", "<i4"), ("first_term", "<i4"), ("n_must_not", "<i4")], align=True)
HIT_DTYPE = np.dtype([("doc", "<i4"), ("score", "<f4")], align=True)
TERM_POSITIONS_DTYPE_ = np.dtype([("pos_start_fp", "<i8"), ("pay_start_fp", "<i8"), ("last_pos_block_offset", "<i8")], align=True)
PHRASE_TERM_DTYPE = np.dtype([("state", TERM_STATE_DTYPE), ("positions", TERM_POSITIONS_DTYPE_), ("position", "<i4"), ("reserved", "<i4")], align=True)
RESCORE_REQUEST_DTYPE = np.dtype([("query_weight", "<f4"), ("rescore_weight", "<f4"), ("mode", "<i4"), ("window_size", "<i4")], align=True)
RESCORE_AVG, RESCORE_MAX, RESCORE_MIN, RESCORE_TOTAL, RESCORE_MULTIPLY = range(5)
PHRASE_QUERY_DTYPE = np.dtype([("n_terms", "<i4"), ("first_term", "<i4"), ("weight", "<f4"), ("sim_table", "<i4"), ("slop", "<i4"), ("next_limit", "<i4")],
                              align=True)
assert PHRASE_TERM_DTYPE.itemsize == 64 and PHRASE_QUERY_DTYPE.itemsize == 24
FIELD_INFO_DTYPE = np.dtype([("number", "<i4"), ("index_options", "<i4"), ("has_payloads", "<i4"), ("flags", "<i4")], align=True)
FIELD_STATS_DTYPE = np.dtype([("num_terms", "<i8"), ("sum_total_term_freq", "<i8"), ("sum_doc_freq", "<i8"), ("doc_count", "<i4"),
                              ("longs_size", "<i4")], align=True)
INDEX_OPTIONS_DOCS, INDEX_OPTIONS_DOCS_AND_FREQS, INDEX_OPTIONS_POSITIONS, INDEX_OPTIONS_OFFSETS = 1, 2, 3, 4
FIELD_STORES_PAYLOADS = 0x100  # RGPU_FIELD_STORES_PAYLOADS: or-ed into index_options (>= 3) at upload
TERM_POSITIONS_DTYPE = np.dtype([("pos_start_fp", "<i8"), ("pay_start_fp", "<i8"), ("last_pos_block_offset", "<i8")], align=True)
SEGMENT_INFO_DTYPE = np.dtype([("max_doc", "<i4"), ("is_compound_file", "<i4"), ("version", "<i4", (3,)), ("n_files", "<i4"),
                               ("n_sort_fields", "<i4"), ("reserved", "<i4"), ("id", "u1", (16,))], align=True)
COMMIT_SEGMENT_DTYPE = np.dtype([("name", "S48"), ("codec", "S16"), ("id", "u1", (16,)), ("del_gen", "<i8"), ("field_infos_gen", "<i8"),
                                 ("dv_gen", "<i8"), ("del_count", "<i4"), ("reserved", "<i4")], align=True)
assert FIELD_INFO_DTYPE.itemsize == 16 and FIELD_STATS_DTYPE.itemsize == 32
COMPOUND_ENTRY_DTYPE = np.dtype([("id", "S112"), ("offset", "<i8"), ("length", "<i8")], align=True)
assert SEGMENT_INFO_DTYPE.itemsize == 48 and COMMIT_SEGMENT_DTYPE.itemsize == 112 and COMPOUND_ENTRY_DTYPE.itemsize == 128
assert TERM_STATE_DTYPE.itemsize == 32 and QUERY_TERM_DTYPE.itemsize == 40 and QUERY_DTYPE.itemsize == 16 and HIT_DTYPE.itemsize == 8

STATUS_NAMES = {0: "OK", -1: "IllegalState", -2: "IllegalArgument", -3: "UnexpectedEOF", -4: "CorruptIndex",
                -5: "UnsupportedOperation", -6: "IOError", -7: "RuntimeError"}

# every symbol include/rucene_gpu.h declares (tests/test_abi.py checks the header and this list agree)
EXPORTS = [
    "rgpu_init", "rgpu_shutdown", "rgpu_last_error", "rgpu_abi_version", "rgpu_device_name", "rgpu_segment_upload",
    "rgpu_segment_upload_field", "rgpu_segment_release_prepared_terms", "rgpu_segment_get_footprint", "rgpu_segment_attach_positions", "rgpu_segment_attach_payloads", "rgpu_decode_positions", "rgpu_decode_positions_device", "rgpu_search_phrase_batch", "rgpu_rescore_batch",
    "rgpu_segment_free", "rgpu_segment_version", "rgpu_segment_prepare_terms", "rgpu_decode_terms",
    "rgpu_decode_terms_device", "rgpu_advance_batch", "rgpu_sim_table_upload", "rgpu_search_batch",
    "rgpu_search_batch_device", "rgpu_merge_topk_device", "rgpu_bm25_compute_weight", "rgpu_bm25_encode_norm", "rgpu_bm25_term_weights",
    "rgpu_norms_from_lucene53", "rgpu_live_docs_from_lucene50", "rgpu_field_infos_from_lucene60", "rgpu_segment_info_from_lucene62", "rgpu_commit_from_segments_file", "rgpu_compound_entries_from_lucene50", "rgpu_terms_open", "rgpu_terms_close", "rgpu_terms_field_stats",
    "rgpu_terms_lookup", "rgpu_terms_lookup_positions", "rgpu_kernel_stats", "rgpu_kernel_stats_reset", "rgpu_synchronize",
    "rgpu_set_profiling", "rgpu_and_touched_bytes", "rgpu_comm_unique_id", "rgpu_comm_init", "rgpu_comm_destroy",
    "rgpu_search_batch_sharded", "rgpu_comm_status", "rgpu_comm_reserve", "rgpu_comm_gathers_issued", "rgpu_comm_init_all", "rgpu_search_batch_sharded_all", "rgpu_record_bytes",
    "rgpu_search_batch_record_device", "rgpu_merge_records_device", "rgpu_last_search_counters",
    "rgpu_planner_create", "rgpu_planner_create_flat", "rgpu_planner_destroy", "rgpu_planner_sim_table", "rgpu_planner_set_sim_table", "rgpu_plan_uniform_ids",
    "rgpu_plan_uniform_bytes", "rgpu_plan_batch_ids", "rgpu_plan_batch_bytes", "rgpu_planner_search_uniform_ids_device", "rgpu_planner_search_uniform_ids_sharded",
]


class RgpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s (%d): %s" % (STATUS_NAMES.get(status, "?"), status, message))
        self.status = status


class _Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("blocks_per_item", C.c_int32), ("and_blocks_per_item", C.c_int32),
                ("profile_kernels", C.c_int32), ("or_window_docs", C.c_int32), ("or_dense_clauses", C.c_int32),
                ("raw_norms", C.c_int32), ("or_wide", C.c_int32), ("or_wide_window_docs", C.c_int32),
                ("req_opt_rule", C.c_int32), ("or_bitmaps", C.c_int32), ("or_lazy_cells", C.c_int32), ("and_bitmaps", C.c_int32), ("bitmap_budget_mib", C.c_int32), ("prepared_budget_mib", C.c_int32), ("or_deferred", C.c_int32),
                ("comm_force_gather", C.c_int32)]


SEARCH_COUNTERS_DTYPE = np.dtype([("op", "<i4"), ("reserved", "<i4"), ("postings_covered", "<i8"), ("postings_decoded", "<i8"),
                                  ("blocks_decoded", "<i8"), ("touched_bytes", "<i8")], align=True)
FOOTPRINT_DTYPE = np.dtype([(n, "<i8") for n in ("doc_file_bytes", "norms_bytes", "live_docs_bytes", "positions_file_bytes", "directory_bytes",
                                                 "block_store_bytes", "posting_norms_bytes", "prepared_terms", "doc_bitmap_bytes", "doc_bitmap_terms", "doc_bitmap_refused")], align=True)
PLAN_STATS_DTYPE = np.dtype([("max_doc", "<i8"), ("doc_count", "<i8"), ("sum_total_term_freq", "<i8"), ("k1", "<f4"), ("b", "<f4")], align=True)
assert SEARCH_COUNTERS_DTYPE.itemsize == 40 and PLAN_STATS_DTYPE.itemsize == 32


class _KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("postings", C.c_int64),
                ("timed_launches", C.c_int64), ("min_ms", C.c_double), ("median_ms", C.c_double), ("max_ms", C.c_double)]


def lib_path():
    # RUCENE_GPU_LIB: developer knob for A/B-ing kernel build variants (same C ABI, same exports)
    return os.environ.get("RUCENE_GPU_LIB") or os.path.join(_HERE, _LIB_NAME)


_lib = None


def _share_torch_hip_runtime():
    """One HIP/HSA runtime per process. The PyTorch-ROCm wheel bundles its own libamdhip64.so /
    libhsa-runtime64.so (same SONAMEs as /opt/rocm's); if librucene_gpu.so pulled in the system copies first, a
    later `import torch` would bring up a second runtime that cannot see the GPU. Loading torch's copies first
    (by path, without importing torch) makes our NEEDED entries resolve to them, whatever the import order."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        p = os.path.join(libdir, name)
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """Load librucene_gpu.so. Fails loudly when the HIP extension has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). rucene_amd has no CPU fallback." % path)
    _share_torch_hip_runtime()
    L = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "rgpu_init": (i32, [i32, C.POINTER(_Config), C.POINTER(vp)]),
        "rgpu_shutdown": (None, [vp]),
        "rgpu_last_error": (C.c_char_p, [vp]),
        "rgpu_abi_version": (i32, []),
        "rgpu_device_name": (i32, [vp, C.c_char_p, C.c_size_t]),
        "rgpu_segment_upload": (i32, [vp, vp, C.c_size_t, vp, i32, i32, vp, C.POINTER(vp)]),
        "rgpu_segment_upload_field": (i32, [vp, vp, C.c_size_t, vp, i32, i32, vp, i32, C.POINTER(vp)]),
        "rgpu_segment_free": (None, [vp]),
        "rgpu_segment_version": (i32, [vp]),
        "rgpu_segment_prepare_terms": (i32, [vp, vp, i64]),
        "rgpu_segment_release_prepared_terms": (i32, [vp]),
        "rgpu_segment_get_footprint": (i32, [vp, vp]),
        "rgpu_segment_attach_positions": (i32, [vp, vp, C.c_size_t]),
        "rgpu_segment_attach_payloads": (i32, [vp, vp, C.c_size_t]),
        "rgpu_decode_positions": (i32, [vp, vp, vp, C.c_int64, vp]),
        "rgpu_decode_positions_device": (i32, [vp, vp, vp, C.c_int64, vp, vp]),
        "rgpu_search_phrase_batch": (i32, [vp, vp, i32, vp, i32, i32, vp, vp]),
        "rgpu_rescore_batch": (i32, [vp, vp, i32, vp, i32, vp, i32, vp, i32]),
        "rgpu_decode_terms": (i32, [vp, vp, i64, vp, vp]),
        "rgpu_decode_terms_device": (i32, [vp, vp, i64, vp, vp, vp]),
        "rgpu_advance_batch": (i32, [vp, vp, vp, i64, vp, vp]),
        "rgpu_sim_table_upload": (i32, [vp, vp, f32]),
        "rgpu_search_batch": (i32, [vp, vp, i32, vp, i32, i32, vp, vp]),
        "rgpu_search_batch_device": (i32, [vp, vp, i32, vp, i32, i32, vp, vp, vp]),
        "rgpu_merge_topk_device": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp]),
        "rgpu_bm25_compute_weight": (i32, [f32, f32, i64, i64, i64, vp, i32, f32, vp, vp, vp]),
        "rgpu_bm25_encode_norm": (C.c_uint8, [f32, i32]),
        "rgpu_bm25_term_weights": (i32, [i64, i64, vp, i64, f32, vp]),
        "rgpu_norms_from_lucene53": (i32, [vp, C.c_size_t, vp, C.c_size_t, i32, i32, vp]),
        "rgpu_live_docs_from_lucene50": (i32, [vp, C.c_size_t, i32, i32, vp]),
        "rgpu_segment_info_from_lucene62": (i32, [vp, C.c_size_t, vp, vp]),
        "rgpu_commit_from_segments_file": (i32, [vp, C.c_size_t, i64, vp, i32]),
        "rgpu_compound_entries_from_lucene50": (i32, [vp, C.c_size_t, vp, C.c_size_t, vp, vp, i32]),
        "rgpu_field_infos_from_lucene60": (i32, [vp, C.c_size_t, vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]),
        "rgpu_terms_open": (i32, [vp, C.c_size_t, vp, C.c_size_t, vp, i32, i32, C.POINTER(vp)]),
        "rgpu_terms_close": (None, [vp]),
        "rgpu_terms_field_stats": (i32, [vp, i32, vp]),
        "rgpu_terms_lookup": (i32, [vp, i32, vp, vp, i32, vp, vp]),
        "rgpu_terms_lookup_positions": (i32, [vp, i32, vp, vp, i32, vp, vp, vp]),
        "rgpu_kernel_stats": (i32, [vp, C.POINTER(_KernelStat), i32]),
        "rgpu_kernel_stats_reset": (None, [vp]),
        "rgpu_synchronize": (i32, [vp]),
        "rgpu_set_profiling": (i32, [vp, i32]),
        "rgpu_and_touched_bytes": (i32, [vp, C.POINTER(i64)]),
        "rgpu_comm_unique_id": (i32, [vp]),
        "rgpu_comm_init": (i32, [vp, i32, i32, vp, C.POINTER(vp)]),
        "rgpu_comm_destroy": (None, [vp]),
        "rgpu_search_batch_sharded": (i32, [vp, vp, vp, i32, vp, i32, i32, vp, vp, vp]),
        "rgpu_comm_status": (i32, [vp, vp]),
        "rgpu_comm_reserve": (i32, [vp, i32, i32]),
        "rgpu_comm_gathers_issued": (i64, [vp]),
        "rgpu_comm_init_all": (i32, [vp, i32, vp]),
        "rgpu_search_batch_sharded_all": (i32, [vp, vp, i32, vp, i32, vp, i32, i32, vp, vp, vp]),
        "rgpu_record_bytes": (i64, [i32, i32]),
        "rgpu_search_batch_record_device": (i32, [vp, vp, i32, vp, i32, i32, vp, vp]),
        "rgpu_merge_records_device": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
        "rgpu_last_search_counters": (i32, [vp, vp]),
        "rgpu_planner_create": (i32, [vp, vp, vp, vp, i32, C.POINTER(vp)]),
        "rgpu_planner_create_flat": (i32, [vp, vp, vp, i64, vp, i64, C.POINTER(vp)]),
        "rgpu_planner_destroy": (None, [vp]),
        "rgpu_planner_sim_table": (i32, [vp]),
        "rgpu_planner_set_sim_table": (i32, [vp, i32]),
        "rgpu_plan_uniform_ids": (i32, [vp, i32, i32, i32, vp, vp, vp]),
        "rgpu_plan_uniform_bytes": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
        "rgpu_planner_search_uniform_ids_device": (i32, [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]),
        "rgpu_planner_search_uniform_ids_sharded": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]),
        "rgpu_plan_batch_ids": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, i64]),
        "rgpu_plan_batch_bytes": (i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise RgpuError(rc, lib().rgpu_last_error(None).decode(errors="replace"))
    return rc


def bm25_compute_weight(k1, b, max_doc, doc_count, sum_total_term_freq, doc_freqs, boost=1.0):
    """BM25Similarity::compute_weight -> (weight, idf, cache[256])."""
    dfs = np.ascontiguousarray(np.atleast_1d(doc_freqs), dtype=np.int64)
    w, idf = C.c_float(), C.c_float()
    cache = np.zeros(256, dtype=np.float32)
    _check(lib().rgpu_bm25_compute_weight(k1, b, max_doc, doc_count, sum_total_term_freq, dfs.ctypes.data, dfs.size, boost,
                                          C.addressof(w), C.addressof(idf), cache.ctypes.data))
    return w.value, idf.value, cache


def bm25_term_weights(max_doc, doc_count, doc_freqs, boost=1.0):
    """idf(df) * boost for a whole array of single-term queries (rgpu_bm25_term_weights)."""
    dfs = np.ascontiguousarray(np.atleast_1d(doc_freqs), dtype=np.int64)
    out = np.zeros(dfs.size, dtype=np.float32)
    _check(lib().rgpu_bm25_term_weights(max_doc, doc_count, dfs.ctypes.data, dfs.size, boost, out.ctypes.data))
    return out


def bm25_encode_norm(boost, field_length):
    return int(lib().rgpu_bm25_encode_norm(boost, field_length))


def norms_from_lucene53(nvm, nvd, field_number, max_doc):
    """Lucene53NormsProducer: a segment's ".nvm" + ".nvd" bytes -> the u8[max_doc] norms array of one field
    (norms.get(doc) & 0xFF, what BM25 reads). Host-side parse, no GPU involved."""
    m = np.frombuffer(bytes(nvm), dtype=np.uint8)
    d = np.frombuffer(bytes(nvd), dtype=np.uint8)
    out = np.zeros(max(int(max_doc), 0), dtype=np.uint8)
    _check(lib().rgpu_norms_from_lucene53(m.ctypes.data, m.size, d.ctypes.data, d.size, int(field_number), int(max_doc),
                                          out.ctypes.data if out.size else None))
    return out


def live_docs_from_lucene50(liv, max_doc, del_count=-1):
    """Lucene50LiveDocsFormat::read_live_docs: a segment's ".liv" bytes -> u64[ceil(max_doc / 64)] FixedBitSet words.
    Host-side parse, no GPU involved."""
    b = np.frombuffer(bytes(liv), dtype=np.uint8)
    out = np.zeros((max(int(max_doc), 1) + 63) // 64, dtype=np.uint64)
    _check(lib().rgpu_live_docs_from_lucene50(b.ctypes.data, b.size, int(max_doc), int(del_count), out.ctypes.data))
    return out


def segment_info_from_lucene62(si, expected_id=None):
    """Lucene62SegmentInfoFormat::read: ".si" bytes -> dict(max_doc, is_compound_file, version, n_files, n_sort_fields, id)."""
    b = np.frombuffer(bytes(si), dtype=np.uint8)
    out = np.zeros(1, dtype=SEGMENT_INFO_DTYPE)
    eid = np.frombuffer(bytes(expected_id), dtype=np.uint8) if expected_id is not None else None
    _check(lib().rgpu_segment_info_from_lucene62(b.ctypes.data, b.size, eid.ctypes.data if eid is not None else None, out.ctypes.data))
    r = out[0]
    return dict(max_doc=int(r["max_doc"]), is_compound_file=bool(r["is_compound_file"]), version=tuple(int(v) for v in r["version"]),
                n_files=int(r["n_files"]), n_sort_fields=int(r["n_sort_fields"]), id=r["id"].tobytes())


def commit_from_segments_file(data, generation=-1):
    """SegmentInfos::read_commit: "segments_N" bytes -> [dict(name, codec, id, del_gen, del_count, field_infos_gen, dv_gen)]."""
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    n = _check(lib().rgpu_commit_from_segments_file(b.ctypes.data, b.size, int(generation), None, 0))
    out = np.zeros(max(n, 1), dtype=COMMIT_SEGMENT_DTYPE)
    _check(lib().rgpu_commit_from_segments_file(b.ctypes.data, b.size, int(generation), out.ctypes.data, n))
    return [dict(name=r["name"].decode(), codec=r["codec"].decode(), id=r["id"].tobytes(), del_gen=int(r["del_gen"]),
                 del_count=int(r["del_count"]), field_infos_gen=int(r["field_infos_gen"]), dv_gen=int(r["dv_gen"])) for r in out[:n]]


def compound_files_from_lucene50(cfe, cfs, expected_id=None):
    """Lucene50CompoundReader: the ".cfe" + ".cfs" pair of a compound segment -> {entry id: bytes of that file}, ids being
    file names without the segment name (".fnm", "_Lucene50_0.doc", ...)."""
    e = np.frombuffer(bytes(cfe), dtype=np.uint8)
    raw = bytes(cfs)
    d = np.frombuffer(raw, dtype=np.uint8)
    eid = np.frombuffer(bytes(expected_id), dtype=np.uint8) if expected_id is not None else None
    args = (e.ctypes.data, e.size, d.ctypes.data, d.size, eid.ctypes.data if eid is not None else None)
    n = _check(lib().rgpu_compound_entries_from_lucene50(*args, None, 0))
    out = np.zeros(max(n, 1), dtype=COMPOUND_ENTRY_DTYPE)
    _check(lib().rgpu_compound_entries_from_lucene50(*args, out.ctypes.data, n))
    return {r["id"].decode(): raw[int(r["offset"]):int(r["offset"]) + int(r["length"])] for r in out[:n]}


def field_infos_from_lucene60(fnm):
    """Lucene60FieldInfosFormat::read: ".fnm" bytes -> [dict(name, number, index_options, has_payloads, omit_norms,
    store_term_vector, doc_values_type)] in file order (ascending field number). Host-side parse."""
    b = np.frombuffer(bytes(fnm), dtype=np.uint8)
    names_len = C.c_size_t(0)
    n = _check(lib().rgpu_field_infos_from_lucene60(b.ctypes.data, b.size, None, 0, None, 0, C.byref(names_len)))
    infos = np.zeros(max(n, 1), dtype=FIELD_INFO_DTYPE)
    names = C.create_string_buffer(max(names_len.value, 1))
    _check(lib().rgpu_field_infos_from_lucene60(b.ctypes.data, b.size, infos.ctypes.data, n, names, names_len.value, C.byref(names_len)))
    out = []
    for rec, name in zip(infos[:n], names.raw[:names_len.value].split(b"\0")):
        flags = int(rec["flags"])
        out.append(dict(name=name.decode("utf-8"), number=int(rec["number"]), index_options=int(rec["index_options"]),
                        has_payloads=bool(rec["has_payloads"]), omit_norms=bool(flags & 1), store_term_vector=bool(flags & 2),
                        doc_values_type=(flags >> 8) & 0xFF))
    return out


class TermDictionary:
    """rgpu_terms: a segment's block-tree term dictionary (.tim + .tip), resolved term bytes -> rgpu_term_state on the
    host. `field_infos`: iterable of (number, index_options[, has_payloads]) for the segment's indexed fields."""

    def __init__(self, tim, tip, field_infos, max_doc):
        self._tim = np.frombuffer(bytes(tim), dtype=np.uint8)
        self._tip = np.frombuffer(bytes(tip), dtype=np.uint8)
        infos = np.zeros(len(field_infos), dtype=FIELD_INFO_DTYPE)
        for i, fi in enumerate(field_infos):
            infos[i]["number"], infos[i]["index_options"] = int(fi[0]), int(fi[1])
            infos[i]["has_payloads"] = int(fi[2]) if len(fi) > 2 else 0
        h = C.c_void_p()
        _check(lib().rgpu_terms_open(self._tim.ctypes.data, self._tim.size, self._tip.ctypes.data, self._tip.size,
                                     infos.ctypes.data if infos.size else None, infos.size, int(max_doc), C.byref(h)))
        self._h = h

    def field_stats(self, field_number):
        """Terms::{size, sum_total_term_freq, sum_doc_freq, doc_count}; None when the field is not in this segment."""
        out = np.zeros(1, dtype=FIELD_STATS_DTYPE)
        rc = lib().rgpu_terms_field_stats(self._h, int(field_number), out.ctypes.data)
        if rc == -2:
            return None
        _check(rc)
        return {k: int(out[0][k]) for k in FIELD_STATS_DTYPE.names}

    def lookup(self, field_number, terms, with_positions=False):
        """seek_exact + term_state for each of `terms` (bytes) -> (TERM_STATE_DTYPE[n], found bool[n]); with_positions adds
        the TERM_POSITIONS_DTYPE[n] pointers of a positions field in between."""
        terms = [bytes(t) for t in terms]
        offs = np.zeros(len(terms) + 1, dtype=np.int64)
        np.cumsum([len(t) for t in terms], out=offs[1:])
        flat = np.frombuffer(b"".join(terms) or b"\0", dtype=np.uint8)
        states = np.zeros(len(terms), dtype=TERM_STATE_DTYPE)
        found = np.zeros(max(len(terms), 1), dtype=np.uint8)
        if with_positions:
            pos = np.zeros(max(len(terms), 1), dtype=TERM_POSITIONS_DTYPE)
            _check(lib().rgpu_terms_lookup_positions(self._h, int(field_number), flat.ctypes.data, offs.ctypes.data, len(terms),
                                                     states.ctypes.data if len(terms) else None, pos.ctypes.data, found.ctypes.data))
            return states, pos[:len(terms)], found[:len(terms)].astype(bool)
        _check(lib().rgpu_terms_lookup(self._h, int(field_number), flat.ctypes.data, offs.ctypes.data, len(terms),
                                       states.ctypes.data if len(terms) else None, found.ctypes.data))
        return states, found[:len(terms)].astype(bool)

    def close(self):
        if getattr(self, "_h", None):
            lib().rgpu_terms_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """rgpu_ctx: one per process per GPU."""

    def __init__(self, device=0, profile_kernels=False, blocks_per_item=0, and_blocks_per_item=0, or_window_docs=0,
                 raw_norms=False, or_dense_clauses=0, or_wide=0, or_wide_window_docs=0, req_opt_rule=0, or_bitmaps=0, or_lazy_cells=0, and_bitmaps=0, bitmap_budget_mib=0,
                 prepared_budget_mib=0, or_deferred=False, comm_force_gather=False):
        cfg = _Config()
        cfg.abi_version = ABI_VERSION
        cfg.blocks_per_item = blocks_per_item
        cfg.profile_kernels = int(profile_kernels)
        cfg.and_blocks_per_item = and_blocks_per_item
        cfg.or_window_docs = or_window_docs
        cfg.or_dense_clauses = or_dense_clauses
        cfg.raw_norms = int(raw_norms)
        cfg.or_wide = or_wide
        cfg.or_wide_window_docs = or_wide_window_docs
        cfg.req_opt_rule = req_opt_rule
        cfg.or_bitmaps = or_bitmaps
        cfg.or_lazy_cells = or_lazy_cells
        cfg.and_bitmaps = and_bitmaps
        cfg.bitmap_budget_mib = bitmap_budget_mib
        cfg.prepared_budget_mib = prepared_budget_mib
        cfg.or_deferred = int(or_deferred)
        cfg.comm_force_gather = int(comm_force_gather)
        h = C.c_void_p()
        _check(lib().rgpu_init(device, C.byref(cfg), C.byref(h)))
        self._h = h
        self._tables = {}
        self._segments = []  # weakrefs: segments must be freed before the ctx they belong to

    @property
    def device_name(self):
        buf = C.create_string_buffer(128)
        _check(lib().rgpu_device_name(self._h, buf, 128))
        return buf.value.decode()

    def sim_table(self, cache, k1):
        """Upload (or reuse) a BM25 norm cache; returns the handle for rgpu_query_term.sim_table."""
        cache = np.ascontiguousarray(cache, dtype=np.float32)
        key = (cache.tobytes(), float(np.float32(k1)))
        if key not in self._tables:
            self._tables[key] = _check(lib().rgpu_sim_table_upload(self._h, cache.ctypes.data, k1))
        return self._tables[key]

    def synchronize(self):
        _check(lib().rgpu_synchronize(self._h))

    def set_profiling(self, on):
        _check(lib().rgpu_set_profiling(self._h, int(bool(on))))

    def and_touched_bytes(self):
        out = C.c_int64(0)
        _check(lib().rgpu_and_touched_bytes(self._h, C.byref(out)))
        return int(out.value)

    def last_search_counters(self):
        """rgpu_last_search_counters: what the most recent TERM / AND / wide-OR launch really decoded (waits for it)."""
        out = np.zeros(1, dtype=SEARCH_COUNTERS_DTYPE)
        _check(lib().rgpu_last_search_counters(self._h, out.ctypes.data))
        return {k: int(out[0][k]) for k in SEARCH_COUNTERS_DTYPE.names if k != "reserved"}

    def merge_records_device(self, records_ptr, n_ranks, n_queries, k, out_hits_ptr, out_totals_ptr, stream=0):
        """finish_parallel over n_ranks shard records laid out back to back (an all-gather's receive buffer)."""
        _check(lib().rgpu_merge_records_device(self._h, records_ptr, n_ranks, n_queries, k, out_hits_ptr, out_totals_ptr, stream or None))

    def kernel_stats(self):
        arr = (_KernelStat * 32)()
        n = _check(lib().rgpu_kernel_stats(self._h, arr, 32))
        return {arr[i].name.decode(): {"launches": arr[i].launches, "total_ms": arr[i].total_ms, "postings": arr[i].postings,
                                       "timed_launches": arr[i].timed_launches, "min_ms": arr[i].min_ms, "median_ms": arr[i].median_ms,
                                       "max_ms": arr[i].max_ms}
                for i in range(n)}

    def kernel_stats_reset(self):
        lib().rgpu_kernel_stats_reset(self._h)

    def merge_topk_device(self, hits_ptr, totals_ptr, n_lists, n_queries, k, out_hits_ptr, out_totals_ptr, stream=0):
        _check(lib().rgpu_merge_topk_device(self._h, hits_ptr, totals_ptr, n_lists, n_queries, k, out_hits_ptr, out_totals_ptr,
                                            stream or None))

    def close(self):
        if getattr(self, "_h", None):
            for ref in self._segments:
                seg = ref()
                if seg is not None:
                    seg.close()
            self._segments = []
            lib().rgpu_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def record_bytes(n_queries, k):
    """Bytes of one shard's record: [n_queries x k hits][n_queries counts][status]."""
    return int(lib().rgpu_record_bytes(n_queries, k))


class Planner:
    """rgpu_planner: a batch handed over as arrays -> (QUERY_DTYPE[n_queries], QUERY_TERM_DTYPE[sum of clauses]), natively.
    `leaf_terms` / `stats_terms`: TermDictionary handles (terms named by bytes) or TERM_STATE_DTYPE tables (named by id);
    stats_terms None = the searched leaf is the statistics leaf."""

    def __init__(self, ctx, max_doc, doc_count, sum_total_term_freq, leaf_terms, stats_terms=None, k1=1.2, b=0.75, field_number=0,
                 sim_table=None):
        """ctx None + sim_table: the caller uploaded the field's norm cache itself (nothing here touches a GPU)."""
        ps = np.zeros(1, dtype=PLAN_STATS_DTYPE)
        ch = None if ctx is None else ctx._h
        ps[0]["max_doc"], ps[0]["doc_count"], ps[0]["sum_total_term_freq"], ps[0]["k1"], ps[0]["b"] = max_doc, doc_count, sum_total_term_freq, k1, b
        h = C.c_void_p()
        self.flat = not isinstance(leaf_terms, TermDictionary)
        if self.flat:
            lt = np.ascontiguousarray(leaf_terms, dtype=TERM_STATE_DTYPE)
            stt = None if stats_terms is None else np.ascontiguousarray(stats_terms, dtype=TERM_STATE_DTYPE)
            _check(lib().rgpu_planner_create_flat(ch, ps.ctypes.data, lt.ctypes.data if lt.size else None, lt.size,
                                                  None if stt is None or not stt.size else stt.ctypes.data, 0 if stt is None else stt.size, C.byref(h)))
        else:
            self._keep = (leaf_terms, stats_terms)
            _check(lib().rgpu_planner_create(ch, ps.ctypes.data, leaf_terms._h, None if stats_terms is None else stats_terms._h,
                                             int(field_number), C.byref(h)))
        self._h = h
        self.ctx = ctx
        if sim_table is not None:
            _check(lib().rgpu_planner_set_sim_table(self._h, int(sim_table)))

    @property
    def sim_table(self):
        return int(lib().rgpu_planner_sim_table(self._h))

    @staticmethod
    def _bytes(terms):
        terms = [bytes(t) for t in terms]
        offs = np.zeros(len(terms) + 1, dtype=np.int64)
        np.cumsum([len(t) for t in terms], out=offs[1:])
        return np.frombuffer(b"".join(terms) or b"\0", dtype=np.uint8), offs

    def plan_uniform(self, op, terms, n_queries=None, n_clauses=None):
        """`terms`: [n_queries, n_clauses] ids, or (flat bytes planner) a list of n_queries * n_clauses byte strings with
        n_queries / n_clauses given."""
        if self.flat:
            ids = np.ascontiguousarray(terms, dtype=np.int64)
            if ids.ndim == 1:
                ids = ids.reshape(-1, 1)
            nq, nc = ids.shape
            qs = np.empty(nq, dtype=QUERY_DTYPE)
            ts = np.empty(nq * nc, dtype=QUERY_TERM_DTYPE)
            _check(lib().rgpu_plan_uniform_ids(self._h, int(op), nq, nc, ids.ctypes.data, qs.ctypes.data, ts.ctypes.data))
            return qs, ts
        flat, offs = self._bytes(terms)
        nq, nc = int(n_queries), int(n_clauses)
        if nq * nc != offs.size - 1:
            raise ValueError("n_queries * n_clauses terms expected")
        qs = np.empty(nq, dtype=QUERY_DTYPE)
        ts = np.empty(nq * nc, dtype=QUERY_TERM_DTYPE)
        _check(lib().rgpu_plan_uniform_bytes(self._h, int(op), nq, nc, flat.ctypes.data, offs.ctypes.data, qs.ctypes.data, ts.ctypes.data))
        return qs, ts

    def search_uniform_device(self, segment, op, term_ids, k, hits_ptr, totals_ptr, stream=0, comm=None):
        """rgpu_planner_search_uniform_ids_device (comm: ..._sharded): plan + search in one call, enqueue-only. `term_ids`
        [n_queries, n_clauses] flat-table ids (int64, C-contiguous: handed over as they are)."""
        ids = term_ids if (isinstance(term_ids, np.ndarray) and term_ids.dtype == np.int64 and term_ids.flags.c_contiguous) else np.ascontiguousarray(term_ids, dtype=np.int64)
        if ids.ndim == 1:
            ids = ids.reshape(-1, 1)
        nq, nc = ids.shape
        if comm is None:
            _check(lib().rgpu_planner_search_uniform_ids_device(self._h, segment._h, int(op), nq, nc, ids.ctypes.data, int(k), hits_ptr, totals_ptr, stream or None))
        else:
            _check(lib().rgpu_planner_search_uniform_ids_sharded(comm._h, self._h, segment._h, int(op), nq, nc, ids.ctypes.data, int(k), hits_ptr, totals_ptr,
                                                                 stream or None))

    def plan_batch(self, ops, n_terms, terms, n_must_not=None, boosts=None):
        """Any mix of trees: ops[q] as rgpu_query.op, n_terms[q] / n_must_not[q], `terms` in clause order (ids or byte strings)."""
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        nt = np.ascontiguousarray(n_terms, dtype=np.int32)
        nn = None if n_must_not is None else np.ascontiguousarray(n_must_not, dtype=np.int32)
        bo = None if boosts is None else np.ascontiguousarray(boosts, dtype=np.float32)
        qs = np.empty(ops.size, dtype=QUERY_DTYPE)
        if self.flat:
            ids = np.ascontiguousarray(terms, dtype=np.int64).ravel()
            ts = np.empty(max(ids.size, 1), dtype=QUERY_TERM_DTYPE)
            _check(lib().rgpu_plan_batch_ids(self._h, ops.size, ops.ctypes.data, nt.ctypes.data, None if nn is None else nn.ctypes.data,
                                             ids.ctypes.data, None if bo is None else bo.ctypes.data, qs.ctypes.data, ts.ctypes.data, ids.size))
            return qs, ts[:ids.size] if ids.size else ts
        flat, offs = self._bytes(terms)
        n = offs.size - 1
        ts = np.empty(max(n, 1), dtype=QUERY_TERM_DTYPE)
        _check(lib().rgpu_plan_batch_bytes(self._h, ops.size, ops.ctypes.data, nt.ctypes.data, None if nn is None else nn.ctypes.data,
                                           flat.ctypes.data, offs.ctypes.data, None if bo is None else bo.ctypes.data, qs.ctypes.data,
                                           ts.ctypes.data, n))
        return qs, ts[:n] if n else ts

    def close(self):
        if getattr(self, "_h", None):
            lib().rgpu_planner_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 bytes for rgpu_comm_init, created on one rank and handed to the others."""
    buf = np.zeros(128, np.uint8)
    _check(lib().rgpu_comm_unique_id(buf.ctypes.data))
    return buf


class Comm:
    """rgpu_comm: this rank's end of the RCCL communicator over which per-shard top-k is all-gathered."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        uid = np.ascontiguousarray(unique_id, dtype=np.uint8)
        if uid.size != 128:
            raise ValueError("unique_id must hold 128 bytes")
        h = C.c_void_p()
        _check(lib().rgpu_comm_init(ctx._h, n_ranks, rank, uid.ctypes.data, C.byref(h)))
        self._h = h
        self.ctx, self.n_ranks, self.rank = ctx, n_ranks, rank

    def search_batch_sharded(self, segment, queries, terms, k, hits_ptr, totals_ptr, stream=0):
        """local search -> one all-gather of {hits, counts} records -> canonical merge; enqueue-only, collective."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=QUERY_TERM_DTYPE)
        _check(lib().rgpu_search_batch_sharded(self._h, segment._h, q.ctypes.data, q.size, t.ctypes.data, t.size, k, hits_ptr, totals_ptr,
                                               stream or None))

    def reserve(self, n_queries, k):
        """Start-up sizing: every slot's gather buffer for batches of up to n_queries x k (the data path then never allocates)."""
        _check(lib().rgpu_comm_reserve(self._h, n_queries, k))

    def gathers_issued(self):
        """ncclAllGather calls enqueued on this communicator so far."""
        return int(lib().rgpu_comm_gathers_issued(self._h))

    def status(self):
        """Every rank's rgpu_status for the most recent sharded batch (waits for it)."""
        out = np.zeros(self.n_ranks, dtype=np.int32)
        _check(lib().rgpu_comm_status(self._h, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().rgpu_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Segment:
    """rgpu_segment: one uploaded leaf (.doc bytes, norms, live docs in HBM)."""

    def __init__(self, ctx, doc_bytes, norms, max_doc, doc_base=0, live_docs=None, index_options=INDEX_OPTIONS_DOCS_AND_FREQS):
        self.ctx = ctx
        doc = np.ascontiguousarray(doc_bytes, dtype=np.uint8)
        nb = None if norms is None else np.ascontiguousarray(norms, dtype=np.uint8)
        lv = None if live_docs is None else np.ascontiguousarray(live_docs, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib().rgpu_segment_upload_field(ctx._h, doc.ctypes.data, doc.size, None if nb is None else nb.ctypes.data, max_doc,
                                               doc_base, None if lv is None else lv.ctypes.data, index_options, C.byref(h)))
        self._h = h
        self.max_doc, self.doc_base = max_doc, doc_base
        import weakref
        ctx._segments.append(weakref.ref(self))

    @property
    def version(self):
        return lib().rgpu_segment_version(self._h)

    def prepare_terms(self, states):
        st = np.ascontiguousarray(states, dtype=TERM_STATE_DTYPE)
        _check(lib().rgpu_segment_prepare_terms(self._h, st.ctypes.data, st.size))

    def attach_positions(self, pos_bytes):
        """The segment's ".pos" file (a field uploaded with index_options = 3)."""
        pos = np.ascontiguousarray(pos_bytes, dtype=np.uint8)
        _check(lib().rgpu_segment_attach_positions(self._h, pos.ctypes.data, pos.size))

    def attach_payloads(self, pay_bytes):
        """The segment's ".pay" file (a field uploaded with index_options 4 and / or FIELD_STORES_PAYLOADS): validated, not kept."""
        pay = np.ascontiguousarray(pay_bytes, dtype=np.uint8)
        _check(lib().rgpu_segment_attach_payloads(self._h, pay.ctypes.data, pay.size))

    def decode_positions(self, states, positions):
        """BlockPostingIterator::next_position to exhaustion: every position of every doc of the given terms (rgpu_decode_positions)."""
        st = np.ascontiguousarray(np.atleast_1d(states), dtype=TERM_STATE_DTYPE)
        tp = np.ascontiguousarray(np.atleast_1d(positions), dtype=TERM_POSITIONS_DTYPE)
        assert st.size == tp.size
        out = np.zeros(max(1, int(st["total_term_freq"][st["doc_freq"] > 0].sum())), dtype=np.int32)
        _check(lib().rgpu_decode_positions(self._h, st.ctypes.data, tp.ctypes.data, st.size, out.ctypes.data))
        return out[:int(st["total_term_freq"][st["doc_freq"] > 0].sum())]

    def decode_positions_device(self, states, positions, positions_ptr, stream=0):
        st = np.ascontiguousarray(np.atleast_1d(states), dtype=TERM_STATE_DTYPE)
        tp = np.ascontiguousarray(np.atleast_1d(positions), dtype=TERM_POSITIONS_DTYPE)
        _check(lib().rgpu_decode_positions_device(self._h, st.ctypes.data, tp.ctypes.data, st.size, positions_ptr, stream or None))

    def search_phrase_batch(self, queries, terms, k):
        q = np.ascontiguousarray(queries, dtype=PHRASE_QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=PHRASE_TERM_DTYPE)
        hits = np.zeros((q.size, k), dtype=HIT_DTYPE)
        totals = np.zeros(q.size, dtype=np.int64)
        _check(lib().rgpu_search_phrase_batch(self._h, q.ctypes.data, q.size, t.ctypes.data, t.size, k, hits.ctypes.data, totals.ctypes.data))
        return hits, totals

    def rescore_batch(self, queries, terms, requests, hits, finish=True):
        """QueryRescorer over first-pass rows `hits` [n][k] (modified copy returned): rgpu_rescore_batch."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=QUERY_TERM_DTYPE)
        r = np.ascontiguousarray(requests, dtype=RESCORE_REQUEST_DTYPE)
        h = np.ascontiguousarray(hits, dtype=HIT_DTYPE).copy()
        _check(lib().rgpu_rescore_batch(self._h, q.ctypes.data, q.size, t.ctypes.data, t.size, r.ctypes.data, h.shape[1], h.ctypes.data, int(finish)))
        return h

    def release_prepared_terms(self):
        _check(lib().rgpu_segment_release_prepared_terms(self._h))

    def footprint(self):
        """rgpu_segment_footprint: HBM bytes held for this segment, by part."""
        out = np.zeros(1, dtype=FOOTPRINT_DTYPE)
        _check(lib().rgpu_segment_get_footprint(self._h, out.ctypes.data))
        return {k: int(out[0][k]) for k in FOOTPRINT_DTYPE.names}

    def decode_terms(self, states):
        st = np.ascontiguousarray(np.atleast_1d(states), dtype=TERM_STATE_DTYPE)
        total = int(st["doc_freq"].sum())
        docs = np.zeros(max(total, 1), dtype=np.int32)
        freqs = np.zeros(max(total, 1), dtype=np.int32)
        _check(lib().rgpu_decode_terms(self._h, st.ctypes.data, st.size, docs.ctypes.data, freqs.ctypes.data))
        return docs[:total], freqs[:total]

    def decode_terms_device(self, states, docs_ptr, freqs_ptr, stream=0):
        st = np.ascontiguousarray(np.atleast_1d(states), dtype=TERM_STATE_DTYPE)
        _check(lib().rgpu_decode_terms_device(self._h, st.ctypes.data, st.size, docs_ptr, freqs_ptr, stream or None))

    def advance(self, state, targets):
        st = np.ascontiguousarray(np.atleast_1d(state), dtype=TERM_STATE_DTYPE)
        t = np.ascontiguousarray(targets, dtype=np.int32)
        docs = np.zeros(t.size, dtype=np.int32)
        freqs = np.zeros(t.size, dtype=np.int32)
        _check(lib().rgpu_advance_batch(self._h, st.ctypes.data, t.ctypes.data, t.size, docs.ctypes.data, freqs.ctypes.data))
        return docs, freqs

    def search_batch(self, queries, terms, k):
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=QUERY_TERM_DTYPE)
        hits = np.zeros((q.size, max(k, 1)), dtype=HIT_DTYPE)
        totals = np.zeros(q.size, dtype=np.int64)
        _check(lib().rgpu_search_batch(self._h, q.ctypes.data, q.size, t.ctypes.data, t.size, k, hits.ctypes.data, totals.ctypes.data))
        return hits, totals

    def search_batch_record_device(self, queries, terms, k, record_ptr, stream=0):
        """This shard's record ([hits][counts][status], record_bytes(n_queries, k) bytes of device memory); enqueue-only."""
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=QUERY_TERM_DTYPE)
        _check(lib().rgpu_search_batch_record_device(self._h, q.ctypes.data, q.size, t.ctypes.data, t.size, k, record_ptr, stream or None))

    def search_batch_device(self, queries, terms, k, hits_ptr, totals_ptr, stream=0):
        q = np.ascontiguousarray(queries, dtype=QUERY_DTYPE)
        t = np.ascontiguousarray(terms, dtype=QUERY_TERM_DTYPE)
        _check(lib().rgpu_search_batch_device(self._h, q.ctypes.data, q.size, t.ctypes.data, t.size, k, hits_ptr, totals_ptr,
                                              stream or None))

    def close(self):
        if getattr(self, "_h", None):
            lib().rgpu_segment_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
