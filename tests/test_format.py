"""The ".doc" byte grammar: generator (bulk writer, product host code) vs the oracle's line-faithful
Lucene50PostingsWriter restatement, byte for byte; oracle reader round trips; skip-list advance vs brute force.
These paths are parity-unpinned in the reference (no write->read test exists there), so two
independent implementations of the source text are checked against each other. CPU only."""
import numpy as np
import pytest

from rucene_amd import indexgen

EDGE_DFS = [1, 2, 3, 127, 128, 129, 255, 256, 257, 383, 384, 1023, 1024, 1025, 1152, 1153, 2176, 2177, 2304, 8192, 8193, 9000]


def random_postings(rng, df, max_doc, max_freq=10):
    docs = np.sort(rng.choice(max_doc, size=df, replace=False)).astype(np.int32)
    freqs = np.minimum(max_freq, rng.geometric(0.5, size=df)).astype(np.int32)
    return docs, freqs


def make_lists(seed, max_doc, dfs):
    rng = np.random.default_rng(seed)
    out = [random_postings(rng, df, max_doc) for df in dfs]
    # special shapes: all deltas equal (b == 0 doc block), all freqs 1 (b == 0 freq block), huge gaps (wide b)
    out.append((np.arange(0, 300 * 3, 3, dtype=np.int32) + 5, np.ones(300, np.int32)))
    out.append((np.arange(260, dtype=np.int32), np.full(260, 7, np.int32)))
    wide = np.sort(rng.choice(max_doc, size=200, replace=False)).astype(np.int32)
    out.append((wide, rng.integers(1, 2**20, size=200).astype(np.int32)))
    return out


@pytest.mark.parametrize("version", [1, 0])
@pytest.mark.parametrize("max_doc", [20_000, 3_000_000])
def test_generator_matches_oracle_writer(oracle, version, max_doc):
    lists = make_lists(11 + version, max_doc, EDGE_DFS)
    sid = bytes(range(16))
    seg = indexgen.build_explicit(max_doc, lists, version=version, segment_id=sid)
    w = oracle.Writer(max_doc, version=version, segment_id=sid)
    states = [w.write_term(d, f) for d, f in lists]
    ref = w.close()
    assert seg.doc_bytes.size == ref.size
    assert seg.doc_bytes.tobytes() == ref.tobytes()
    for st, gen in zip(states, seg.terms):
        for name in st.dtype.names:
            assert st[name] == gen[name], name


@pytest.mark.parametrize("version", [1, 0])
def test_oracle_reader_roundtrip(oracle, version):
    max_doc = 500_000
    lists = make_lists(23, max_doc, EDGE_DFS + [70_000])
    seg = indexgen.build_explicit(max_doc, lists, version=version)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, max_doc, seg.terms)
    assert oseg.version == version
    for (docs, freqs), st in zip(lists, seg.terms):
        d, f = oseg.decode_term(st)
        assert (d == docs).all() and (f == freqs).all()


def test_oracle_advance_matches_brute_force(oracle):
    max_doc = 2_000_000
    rng = np.random.default_rng(5)
    lists = make_lists(29, max_doc, [129, 1025, 8193, 70_000, 200_000])
    seg = indexgen.build_explicit(max_doc, lists)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, max_doc, seg.terms)
    for (docs, freqs), st in zip(lists, seg.terms):
        it = oseg.postings(st)
        cur = -1
        # increasing targets with mixed strides so every skip level gets exercised
        for _ in range(300):
            stride = int(rng.choice([1, 3, 50, 2_000, 60_000]))
            target = cur + 1 + int(rng.integers(0, stride))
            got = it.advance(target)
            i = int(np.searchsorted(docs, target, side="left"))
            if i >= docs.size:
                assert got == oracle.NO_MORE_DOCS
                break
            assert got == docs[i] and it.freq() == freqs[i]
            cur = got


def test_footer_crc_and_header(oracle):
    seg = indexgen.build_explicit(1000, [(np.arange(0, 600, 2, dtype=np.int32), np.ones(300, np.int32))])
    raw = seg.doc_bytes
    import ctypes as C
    crc = oracle.lib().orc_crc32(raw.ctypes.data_as(C.POINTER(C.c_uint8)), raw.size - 8)
    assert int.from_bytes(raw[-8:].tobytes(), "big") == crc
    assert raw[:4].tobytes() == bytes.fromhex("3FD76C17")
    import zlib
    assert crc == zlib.crc32(raw[:-8].tobytes())


def test_zipf_corpus_is_deterministic_and_readable(oracle):
    a = indexgen.build_zipf(100_000, 20_000)
    b = indexgen.build_zipf(100_000, 20_000)
    assert a.doc_bytes.tobytes() == b.doc_bytes.tobytes() and a.norms.tobytes() == b.norms.tobytes()
    assert a.terms["doc_freq"][0] >= a.terms["doc_freq"][100] >= a.terms["doc_freq"][-1] >= 1
    # norms land in the byte range SURVEY.md §8(d) predicts (len 10000 -> 97 ... len 1 -> 124)
    assert a.norms.min() >= 97 and a.norms.max() <= 124
    oseg = oracle.Segment(a.doc_bytes, a.norms, a.max_doc, a.terms)
    for t in (0, 1, 7, 155, 156, 157, 5000, 19_999):
        d, f = oseg.decode_term(a.terms[t])
        assert d.size == a.terms["doc_freq"][t] and (np.diff(d) > 0).all() and d[-1] < a.max_doc
        assert f.min() >= 1 and f.max() <= 10 and int(f.sum()) == a.terms["total_term_freq"][t]
    other = indexgen.build_zipf(100_000, 20_000, shard=1)
    assert other.doc_bytes.tobytes() != a.doc_bytes.tobytes()
