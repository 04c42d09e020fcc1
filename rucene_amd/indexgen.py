"""ctypes binding over librucene_indexgen.so — the deterministic synthetic Lucene50 segment generator
(rucene_amd/csrc/indexgen/indexgen.cpp). Host-only; produces the inputs of tests and bench.py
(SURVEY.md §8(d) corpus): raw ".doc" bytes, 1-byte norms and a flat BlockTermState table."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librucene_indexgen.so")

TERM_STATE_DTYPE = np.dtype(
    [("doc_start_fp", "<i8"), ("skip_offset", "<i8"), ("total_term_freq", "<i8"), ("doc_freq", "<i4"),
     ("singleton_doc_id", "<i4")], align=True)

DEFAULT_SEED = 0x527563656E65  # "Rucene"


class _Config(C.Structure):
    _fields_ = [("max_doc", C.c_int32), ("version", C.c_int32), ("n_terms", C.c_int64), ("zipf_scale", C.c_double),
                ("seed", C.c_uint64), ("shard", C.c_int32), ("positions", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("%s is missing — run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB_PATH)
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.rgen_build_zipf.restype = vp
        L.rgen_build_zipf.argtypes = [C.POINTER(_Config)]
        L.rgen_build_explicit.restype = vp
        L.rgen_build_explicit.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp]
        L.rgen_free.argtypes = [vp]
        L.rgen_doc_len.restype = C.c_int64
        L.rgen_doc_len.argtypes = [vp]
        for n in ("rgen_doc_bytes", "rgen_norms", "rgen_terms"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.rgen_n_terms.restype = C.c_int64
        L.rgen_n_terms.argtypes = [vp]
        L.rgen_max_doc.restype = C.c_int32
        L.rgen_max_doc.argtypes = [vp]
        L.rgen_stats.argtypes = [vp, vp]
        L.rgen_build_explicit_positions.restype = vp
        L.rgen_build_explicit_positions.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.rgen_pos_len.restype = C.c_int64
        L.rgen_pos_len.argtypes = [vp]
        for n in ("rgen_pos_bytes", "rgen_pos_start_fps", "rgen_last_pos_block_offsets"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.rgen_build_explicit_positions_ex.restype = vp
        L.rgen_build_explicit_positions_ex.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.rgen_pay_len.restype = C.c_int64
        L.rgen_pay_len.argtypes = [vp]
        for n in ("rgen_pay_bytes", "rgen_pay_start_fps"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.rgen_error.restype = C.c_char_p
        L.rgen_error.argtypes = [vp]
        _lib = L
    return _lib


class SyntheticSegment:
    """One generated segment. Arrays are numpy copies (the native object is freed in the constructor)."""

    def __init__(self, handle, doc_base=0):
        L = lib()
        n = L.rgen_doc_len(handle)
        self.max_doc = L.rgen_max_doc(handle)
        self.doc_bytes = np.ctypeslib.as_array(C.cast(L.rgen_doc_bytes(handle), C.POINTER(C.c_uint8)), shape=(n,)).copy()
        self.norms = np.ctypeslib.as_array(C.cast(L.rgen_norms(handle), C.POINTER(C.c_uint8)), shape=(self.max_doc,)).copy()
        nt = L.rgen_n_terms(handle)
        raw = np.ctypeslib.as_array(C.cast(L.rgen_terms(handle), C.POINTER(C.c_uint8)), shape=(nt * 32,)).copy()
        self.terms = raw.view(TERM_STATE_DTYPE)
        st = np.zeros(8, dtype=np.int64)
        L.rgen_stats(handle, st.ctypes.data)
        (self.sum_total_term_freq, self.sum_doc_freq, self.total_postings, self.full_blocks, self.block_bytes,
         self.tail_bytes, self.skip_bytes, _) = [int(x) for x in st]
        self.doc_count = self.max_doc
        self.doc_base = doc_base
        self.live_docs = None
        npos = L.rgen_pos_len(handle)
        self.pos_bytes = self.pos_start_fp = self.last_pos_block_offset = None
        if npos > 0:   # a positions field: the ".pos" file and the two extra BlockTermState pointers per term
            self.pos_bytes = np.ctypeslib.as_array(C.cast(L.rgen_pos_bytes(handle), C.POINTER(C.c_uint8)), shape=(npos,)).copy()
            self.pos_start_fp = np.ctypeslib.as_array(C.cast(L.rgen_pos_start_fps(handle), C.POINTER(C.c_int64)), shape=(nt,)).copy()
            self.last_pos_block_offset = np.ctypeslib.as_array(C.cast(L.rgen_last_pos_block_offsets(handle), C.POINTER(C.c_int64)),
                                                               shape=(nt,)).copy()
        npay = L.rgen_pay_len(handle)
        self.pay_bytes = self.pay_start_fp = None
        if npay > 0:   # the field stores payloads or offsets: the ".pay" file and each term's pointer into it
            self.pay_bytes = np.ctypeslib.as_array(C.cast(L.rgen_pay_bytes(handle), C.POINTER(C.c_uint8)), shape=(npay,)).copy()
            self.pay_start_fp = np.ctypeslib.as_array(C.cast(L.rgen_pay_start_fps(handle), C.POINTER(C.c_int64)), shape=(nt,)).copy()
        err = L.rgen_error(handle).decode()
        L.rgen_free(handle)
        if err:
            raise ValueError(err)


def build_zipf(max_doc, n_terms, zipf_scale=0.2, version=1, seed=DEFAULT_SEED, shard=0, doc_base=0, positions=False):
    """The Zipfian corpus of SURVEY.md 8(d). positions=True: the same postings as a DocsAndFreqsAndPositions field (".pos", position
    pointers in the skip entries; a posting's `freq` positions start at 0..63 and step by 1..16)."""
    cfg = _Config(max_doc, version, n_terms, zipf_scale, seed, shard, 1 if positions else 0)
    h = lib().rgen_build_zipf(C.byref(cfg))
    return SyntheticSegment(h, doc_base)


def build_explicit(max_doc, postings, norms=None, version=1, segment_id=None, doc_base=0):
    """postings: list of (docs, freqs) per term (empty docs -> absent term)."""
    offs = np.zeros(len(postings) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(p[0]) for p in postings])
    docs = np.concatenate([np.asarray(p[0], dtype=np.int32) for p in postings]) if postings else np.zeros(0, np.int32)
    freqs = np.concatenate([np.asarray(p[1], dtype=np.int32) for p in postings]) if postings else np.zeros(0, np.int32)
    docs = np.ascontiguousarray(docs, dtype=np.int32)
    freqs = np.ascontiguousarray(freqs, dtype=np.int32)
    nb = None if norms is None else np.ascontiguousarray(norms, dtype=np.uint8)
    sid = None if segment_id is None else np.frombuffer(segment_id, dtype=np.uint8).copy()
    h = lib().rgen_build_explicit(max_doc, version, len(postings), offs.ctypes.data, docs.ctypes.data, freqs.ctypes.data,
                                  None if nb is None else nb.ctypes.data, None if sid is None else sid.ctypes.data)
    seg = SyntheticSegment(h, doc_base)
    if norms is None:
        seg.norms = None
    return seg


def build_explicit_positions(max_doc, postings, norms=None, version=1, segment_id=None, doc_base=0, offsets=False, payloads=False):
    """A DocsAndFreqsAndPositions field. postings: per term a list of (doc, [positions ascending]) in doc order (an empty
    list -> absent term). The segment carries .doc bytes (skip entries with position pointers), .pos bytes and, per term,
    pos_start_fp / last_pos_block_offset next to the usual term states. offsets / payloads: the field also stores them
    (IndexOptions::DocsAndFreqsAndPositionsAndOffsets / FieldInfo::has_store_payloads) — a doc's entry is then
    (doc, [positions], [(start, end), ...], [payload bytes, ...]) and the segment also carries .pay bytes and pay_start_fp."""
    offs = np.zeros(len(postings) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(p) for p in postings])
    docs = np.ascontiguousarray([e[0] for p in postings for e in p] or [0], dtype=np.int32)
    freqs = np.ascontiguousarray([len(e[1]) for p in postings for e in p] or [0], dtype=np.int32)
    pos_offs = np.zeros(int(offs[-1]) + 1, dtype=np.int64)
    pos_offs[1:] = np.cumsum([len(e[1]) for p in postings for e in p]) if offs[-1] else 0
    positions = np.ascontiguousarray([x for p in postings for e in p for x in e[1]] or [0], dtype=np.int32)
    nb = None if norms is None else np.ascontiguousarray(norms, dtype=np.uint8)
    sid = None if segment_id is None else np.frombuffer(segment_id, dtype=np.uint8).copy()
    if not offsets and not payloads:
        h = lib().rgen_build_explicit_positions(max_doc, version, len(postings), offs.ctypes.data, docs.ctypes.data, freqs.ctypes.data,
                                                pos_offs.ctypes.data, positions.ctypes.data, None if nb is None else nb.ctypes.data,
                                                None if sid is None else sid.ctypes.data)
    else:
        starts = np.ascontiguousarray([o[0] for p in postings for e in p for o in e[2]] if offsets else [0], dtype=np.int32)
        ends = np.ascontiguousarray([o[1] for p in postings for e in p for o in e[2]] if offsets else [0], dtype=np.int32)
        blobs = [bytes(b) for p in postings for e in p for b in e[3]] if payloads else []
        pay_offs = np.zeros(len(blobs) + 1, dtype=np.int64)
        if blobs:
            pay_offs[1:] = np.cumsum([len(b) for b in blobs])
        pay = np.frombuffer(b"".join(blobs) + b"\0", dtype=np.uint8).copy()
        assert not offsets or starts.size == positions.size
        assert not payloads or len(blobs) == positions.size
        h = lib().rgen_build_explicit_positions_ex(max_doc, version, len(postings), offs.ctypes.data, docs.ctypes.data, freqs.ctypes.data,
                                                   pos_offs.ctypes.data, positions.ctypes.data, (1 if offsets else 0) | (2 if payloads else 0),
                                                   starts.ctypes.data, ends.ctypes.data, pay_offs.ctypes.data, pay.ctypes.data,
                                                   None if nb is None else nb.ctypes.data, None if sid is None else sid.ctypes.data)
    seg = SyntheticSegment(h, doc_base)
    if norms is None:
        seg.norms = None
    return seg


def log_uniform_ranks(n, lo, hi, seed):
    """Query term ranks drawn log-uniform in [lo, hi], deterministic (splitmix64)."""
    out = np.zeros(n, dtype=np.int64)
    s = np.uint64(seed)
    mask = (1 << 64) - 1
    x = int(s)
    for i in range(n):
        x = (x + 0x9E3779B97F4A7C15) & mask
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        u = (z >> 11) / float(1 << 53)
        out[i] = min(hi, max(lo, int(np.floor(np.exp(np.log(lo) + u * (np.log(hi + 1) - np.log(lo)))))))
    return out
