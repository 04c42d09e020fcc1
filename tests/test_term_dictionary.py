"""Block-tree term dictionary (".tim" + ".tip") -> term bytes -> BlockTermState. Host-only code on both sides: the
product is rgpu_terms_* (rucene_amd/csrc/host/term_dict.hpp through the C ABI: one enumeration of every term block
into a hash table), the checker is the oracle's restatement of BlockTreeTermsWriter and of the reference's lookup
path — index FST walk, floor-block choice, block scan, decode_metadata (oracle/blocktree.hpp, oracle/fst.hpp). The two
share no code and no algorithm. The reference holds no test for these files (parity unpinned: the source text is the
only authority), so the implementations are checked against each other and against hand-assembled bytes; the FST
itself is pinned by the reference's own test (tests/test_oracle_kat.py)."""
import random
import struct
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def rgpu():
    import __graft_entry__ as g
    g.build()
    import rucene_amd
    return rucene_amd


SID = bytes(range(16))


def _header(codec, version, sid=SID, suffix=b""):
    return struct.pack(">I", 0x3FD76C17) + bytes([len(codec)]) + codec + struct.pack(">i", version) + sid + bytes([len(suffix)]) + suffix


def _footer(body):
    f = body + struct.pack(">Ii", 0xC02893E8, 0)
    return f + struct.pack(">q", zlib.crc32(f) & 0xFFFFFFFF)


def _hand_assembled():
    """field 0 (DocsAndFreqs): "a" -> df 1, ttf 3, singleton doc 7, doc_start_fp 40; "b" -> df 2, ttf 2, doc_start_fp 40."""
    head = _header(b"BlockTreeTermsDict", 3) + _header(b"Lucene50PostingsWriterTerms", 1) + b"\x80\x01"   # vint BLOCK_SIZE
    assert len(head) == 99
    block = bytes([
        0x05,                          # 2 entries << 1 | last block of its floor group
        0x09, 0x01, 0x61, 0x01, 0x62,  # 4 suffix bytes << 1 | leaf; "a", "b"
        0x04, 0x01, 0x02, 0x02, 0x00,  # stats: (df 1, ttf-df 2), (df 2, ttf-df 0)
        0x03, 0x28, 0x07, 0x00])       # meta: doc fp +40 absolute, singleton 7; doc fp +0
    root_code = bytes([0x8E, 0x03])    # vlong (99 << 2 | HAS_TERMS)
    summary = (bytes([0x01, 0x00, 0x02, 0x02]) + root_code +   # 1 field; number 0; 2 terms; root code
               bytes([0x05, 0x03, 0x02, 0x01]) +               # sumTTF 5, sumDF 3, docCount 2, longsSize 1
               b"\x01a\x01b")                                  # min term, max term
    tim = _footer(head + block + summary + struct.pack(">q", len(head) + len(block)))
    ihead = _header(b"BlockTreeTermsIndex", 3)
    fst = (struct.pack(">I", 0x3FD76C17) + b"\x03FST" + struct.pack(">i", 6) +
           bytes([0x01, 0x03, 0x03, 0x8E, 0x02]) +   # has empty output; 3 bytes; (vint 2, 8E 03) reversed
           bytes([0x00, 0x00, 0x01, 0x00]))          # BYTE1; start node 0; 1 byte of arcs: the builder's leading 0
    tip = _footer(ihead + fst + bytes([len(ihead)]) + struct.pack(">q", len(ihead) + len(fst)))
    return tim, tip


def _full_states(oracle, n):
    return np.zeros(n, dtype=oracle.FULL_TERM_STATE_DTYPE)


def test_hand_assembled_files(rgpu, oracle):
    tim, tip = _hand_assembled()
    st = _full_states(oracle, 2)
    st["base"]["doc_freq"] = [1, 2]
    st["base"]["total_term_freq"] = [3, 2]
    st["base"]["doc_start_fp"] = [40, 40]
    st["base"]["singleton_doc_id"] = [7, -1]
    st["base"]["skip_offset"] = -1
    st["last_pos_block_offset"] = -1
    # the writer restatement produces exactly these bytes
    wtim, wtip = oracle.blocktree_write([dict(number=0, doc_count=2, terms=[b"a", b"b"], states=st)], segment_id=SID)
    assert wtim == tim
    assert wtip == tip
    # both readers resolve them
    r = oracle.BlockTreeReader(tim, tip, [dict(number=0)], max_doc=10)
    got, found = r.seek_exact(0, [b"a", b"b", b"c", b"", b"ab"])
    assert found.tolist() == [True, True, False, False, False]
    d = rgpu.TermDictionary(tim, tip, [(0, 2)], max_doc=10)
    pgot, pfound = d.lookup(0, [b"a", b"b", b"c", b"", b"ab"])
    assert pfound.tolist() == [True, True, False, False, False]
    for name in pgot.dtype.names:
        assert pgot[name][:2].tolist() == st["base"][name].tolist() == got["base"][name][:2].tolist()
    assert pgot["doc_freq"][2:].tolist() == [0, 0, 0] and pgot["skip_offset"][2:].tolist() == [-1, -1, -1]
    assert d.field_stats(0) == dict(num_terms=2, sum_total_term_freq=5, sum_doc_freq=3, doc_count=2, longs_size=1)
    assert d.field_stats(1) is None
    assert r.field_stats(0)["root_block_fp"] == 99


def _vocab(rng, n, alphabet, maxlen):
    keys = set()
    while len(keys) < n:
        keys.add(bytes(rng.choice(alphabet) for _ in range(rng.randint(1, maxlen))))
    return sorted(keys)


def _states(oracle, rng, n, opts):
    st = _full_states(oracle, n)
    fp, pfp, payfp = 40, 0, 0
    for i in range(n):
        df = rng.choice([1, 1, 2, 5, 100, 128, 129, 300, 5000])
        b = st[i]["base"]
        b["doc_freq"] = df
        b["total_term_freq"] = df + rng.randint(0, 50) if opts != oracle.IO_DOCS else -1
        b["doc_start_fp"] = fp
        b["singleton_doc_id"] = rng.randint(0, 10**6) if df == 1 else -1
        if df > 1:
            fp += rng.randint(1, 500)
        b["skip_offset"] = rng.randint(1, 10**5) if df > 128 else -1
        st[i]["last_pos_block_offset"] = -1
        if opts >= oracle.IO_DOCS_FREQS_POS:
            st[i]["pos_start_fp"] = pfp
            pfp += rng.randint(0, 1000)
            if b["total_term_freq"] > 128:
                st[i]["last_pos_block_offset"] = rng.randint(0, 10**4)
            if opts >= oracle.IO_DOCS_FREQS_POS_OFFS:
                st[i]["pay_start_fp"] = payfp
                payfp += rng.randint(0, 100)
    return st


ALPHABETS = {"bytes": list(range(1, 256)), "abcdefg": list(b"abcdefg"), "ab": list(b"ab"), "lower": list(b"abcdefghijklmnopqrstuvwxyz")}


@pytest.mark.parametrize("alphabet", sorted(ALPHABETS))
@pytest.mark.parametrize("blocks", [(25, 48), (2, 2), (2, 4), (5, 10)])
def test_product_matches_oracle_lookup(rgpu, oracle, alphabet, blocks):
    """Every term of random vocabularies — deep shared prefixes ("ab"), wide fan-out ("bytes" -> array arcs in the FST),
    tiny block sizes (floor blocks everywhere) — resolves to the same BlockTermState through the product's hash table and
    through the oracle's FST walk + block scan; absent terms are absent in both."""
    rng = random.Random(zlib.crc32(repr((alphabet, blocks)).encode()))
    alpha = ALPHABETS[alphabet]
    for n in (1, 2, 24, 25, 26, 48, 49, 100, 700, 4000):
        maxlen = 14 if len(alpha) == 2 else rng.choice([3, 6, 12])
        if len(alpha) ** maxlen < 2 * n:
            n = min(n, 50)
        for opts in (oracle.IO_DOCS, oracle.IO_DOCS_FREQS, oracle.IO_DOCS_FREQS_POS, oracle.IO_DOCS_FREQS_POS_OFFS):
            terms = _vocab(rng, n, alpha, maxlen)
            st = _states(oracle, rng, n, opts)
            tim, tip = oracle.blocktree_write([dict(number=3, index_options=opts, doc_count=1, terms=terms, states=st)], *blocks)
            r = oracle.BlockTreeReader(tim, tip, [dict(number=3, index_options=opts)], max_doc=10**7)
            d = rgpu.TermDictionary(tim, tip, [(3, opts)], max_doc=10**7)
            known = set(terms)
            probes = terms + [t for t in ([x + b"\x01" for x in terms[:40]] + [x[:-1] for x in terms[:40]] + [b"", b"\xff" * 3])
                              if t not in known]
            ostates, ofound = r.seek_exact(3, probes)
            pstates, pfound = d.lookup(3, probes)
            assert ofound[:n].all() and not ofound[n:].any()
            assert (pfound == ofound).all()
            for name in pstates.dtype.names:
                assert (pstates[name][:n] == st["base"][name]).all(), name
                assert (ostates["base"][name][:n] == st["base"][name]).all(), name
            assert (pstates["doc_freq"][n:] == 0).all()
            # the rest of BlockTermState for positions fields: rgpu_terms_lookup_positions
            pstates2, ppos, pfound2 = d.lookup(3, probes, with_positions=True)
            assert pstates2.tobytes() == pstates.tobytes() and (pfound2 == pfound).all()
            for name in ("pos_start_fp", "pay_start_fp", "last_pos_block_offset"):
                assert (ostates[name][:n] == st[name]).all(), name
                if opts >= oracle.IO_DOCS_FREQS_POS:
                    assert (ppos[name][:n] == st[name]).all(), name
            if opts < oracle.IO_DOCS_FREQS_POS:
                assert (ppos["pos_start_fp"] == 0).all() and (ppos["last_pos_block_offset"] == -1).all()
            assert (ppos["last_pos_block_offset"][n:] == -1).all() and (ppos["pos_start_fp"][n:] == 0).all()
            stats = d.field_stats(3)
            assert stats["num_terms"] == n and stats["sum_doc_freq"] == int(st["base"]["doc_freq"].sum())
            assert stats == {k: v for k, v in r.field_stats(3).items() if k != "root_block_fp"}
            r.close()
            d.close()


def test_multiple_fields(rgpu, oracle):
    rng = random.Random(7)
    fields = []
    for number, opts, n in ((0, oracle.IO_DOCS_FREQS, 300), (2, oracle.IO_DOCS, 40), (5, oracle.IO_DOCS_FREQS_POS, 1200)):
        terms = _vocab(rng, n, ALPHABETS["lower"], 8)
        fields.append(dict(number=number, index_options=opts, doc_count=1, terms=terms, states=_states(oracle, rng, n, opts)))
    tim, tip = oracle.blocktree_write(fields)
    infos = [(f["number"], f["index_options"]) for f in fields] + [(9, 2)]   # a field without postings in this segment
    d = rgpu.TermDictionary(tim, tip, infos, max_doc=10**7)
    r = oracle.BlockTreeReader(tim, tip, [dict(number=a, index_options=b) for a, b in infos], max_doc=10**7)
    for f in fields:
        ps, pf = d.lookup(f["number"], f["terms"])
        os_, of = r.seek_exact(f["number"], f["terms"])
        assert pf.all() and of.all()
        for name in ps.dtype.names:
            assert (ps[name] == f["states"]["base"][name]).all() and (os_["base"][name] == ps[name]).all()
        # another field's terms are not found under this number (unless shared by chance)
        other = [t for g in fields if g is not f for t in g["terms"] if t not in set(f["terms"])][:100]
        assert not d.lookup(f["number"], other)[1].any()
    assert d.field_stats(9) is None and not d.lookup(9, [b"x"])[1].any()
    assert d.field_stats(2)["sum_total_term_freq"] == -1


def test_synthetic_segment_terms_resolve_to_the_states_the_postings_writer_produced(rgpu, oracle):
    """End to end on the format side: the synthetic index writer lays out a .doc file and reports one BlockTermState
    per term; the block-tree writer files them; the product dictionary must hand the very same states back."""
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(20000, 3000, seed=11)
    order = sorted(range(seg.terms.size), key=lambda r: b"t%06d" % r)
    present = [r for r in order if seg.terms[r]["doc_freq"] > 0]
    terms = [b"t%06d" % r for r in present]
    st = _full_states(oracle, len(present))
    st["base"] = seg.terms[present]
    st["last_pos_block_offset"] = -1
    tim, tip = oracle.blocktree_write([dict(number=0, doc_count=seg.doc_count, terms=terms, states=st)])
    d = rgpu.TermDictionary(tim, tip, [(0, 2)], max_doc=seg.max_doc)
    got, found = d.lookup(0, terms)
    assert found.all()
    assert got.tobytes() == np.ascontiguousarray(seg.terms[present]).tobytes()
    stats = d.field_stats(0)
    assert stats["sum_doc_freq"] == int(seg.terms["doc_freq"].sum())
    assert stats["sum_total_term_freq"] == int(seg.terms["total_term_freq"][present].sum())


def _small(oracle, n=200, blocks=(2, 4)):
    rng = random.Random(3)
    terms = _vocab(rng, n, ALPHABETS["abcdefg"], 6)
    st = _states(oracle, rng, n, oracle.IO_DOCS_FREQS)
    return oracle.blocktree_write([dict(number=1, doc_count=1, terms=terms, states=st)], *blocks), terms


def _refoot(body_with_old_footer):
    return _footer(body_with_old_footer[:-16])


def test_rejects_damaged_files(rgpu, oracle):
    (tim, tip), terms = _small(oracle)
    infos = [(1, 2)]
    ok = rgpu.TermDictionary(tim, tip, infos, max_doc=10**6)
    assert ok.lookup(1, terms)[1].all()

    def status(t, x, infos=infos, max_doc=10**6):
        with pytest.raises(rgpu.RgpuError) as e:
            rgpu.TermDictionary(t, x, infos, max_doc=max_doc)
        return e.value.status

    assert status(tim[:50], tip) in (-3, -4)                       # truncated header
    assert status(tim[:-1], tip) == -4                             # footer not where it must be
    assert status(tim, tip[:-3]) == -4
    assert status(b"\x00" + tim[1:], tip) == -4                    # bad magic
    assert status(tim, _refoot(tip[:30] + bytes([tip[30] ^ 1]) + tip[31:])) == -4   # .tip of another segment (id differs)
    assert status(tim, tip, infos=[(2, 2)]) == -4                  # the summary names a field the caller does not know
    assert status(tim, tip, infos=[(1, 3)]) == -4                  # longs_size contradicts the field's index options
    assert status(tim, tip, max_doc=0) == -4                       # doc_count > max_doc
    v1 = bytearray(tim); v1[4 + 1 + 18 + 3] = 1                    # BlockTreeTermsDict version 1: auto-prefix terms
    x1 = bytearray(tip); x1[4 + 1 + 19 + 3] = 1
    assert status(bytes(v1), bytes(x1)) == -5
    assert status(bytes(v1), tip) == -5
    # a directory pointer into the footer, and one before the first block
    body = tim[:-16]
    assert status(_footer(body[:-8] + struct.pack(">q", len(tim))), tip) == -4
    assert status(_footer(body[:-8] + struct.pack(">q", 10)), tip) == -4
    # flip bytes inside the term blocks: either the structure checks catch it or the dictionary still opens (a flipped
    # suffix byte is a legal different term) — it must never crash or hang
    rng = random.Random(5)
    start = 99
    end = struct.unpack(">q", body[-8:])[0]
    for _ in range(300):
        pos = rng.randrange(start, end)
        dmg = bytearray(tim)
        dmg[pos] ^= 1 << rng.randrange(8)
        try:
            rgpu.TermDictionary(bytes(dmg), tip, infos, max_doc=10**6).close()
        except rgpu.RgpuError as e:
            assert e.status in (-3, -4, -5)


def test_oracle_reader_rejects_what_the_reference_rejects(oracle):
    (tim, tip), _ = _small(oracle)
    with pytest.raises(oracle.OracleError):
        oracle.BlockTreeReader(tim[:-1], tip, [dict(number=1)], max_doc=10**6)
    with pytest.raises(oracle.OracleError):
        oracle.BlockTreeReader(tim, tip, [dict(number=2)], max_doc=10**6)      # invalid field number
    with pytest.raises(oracle.OracleError):
        oracle.BlockTreeReader(tim, tip, [dict(number=1)], max_doc=0)          # invalid doc_count
    with pytest.raises(oracle.OracleError):
        oracle.blocktree_write([dict(number=1, doc_count=1, terms=[b"a"], states=np.zeros(1, oracle.FULL_TERM_STATE_DTYPE))], 1, 48)
    with pytest.raises(oracle.OracleError):
        oracle.blocktree_write([dict(number=1, doc_count=1, terms=[b"a"], states=np.zeros(1, oracle.FULL_TERM_STATE_DTYPE))], 10, 12)


def test_lookup_throughput_against_the_reference_algorithm(rgpu, oracle, capsys):
    """Measurement (host CPU, one thread; DESIGN.md §2b quotes it): batched hash lookups of the product against the
    reference's per-term algorithm — FST walk, floor-block choice, block scan, metadata decode — as restated by the
    oracle. Same files, same probes, identical answers; the product must be clearly faster (it is ~20x here)."""
    import ctypes as C
    import time
    n = 300_000
    terms = [b"t%07d" % i for i in range(n)]
    st = _full_states(oracle, n)
    df = np.maximum(1, 600_000 // (np.arange(n) + 1)).astype(np.int32)
    st["base"]["doc_freq"] = df
    st["base"]["total_term_freq"] = df.astype(np.int64) * 2
    sizes = np.where(df > 1, df.astype(np.int64) * 2, 0)
    st["base"]["doc_start_fp"] = np.cumsum(sizes) - sizes + 100
    st["base"]["singleton_doc_id"] = np.where(df == 1, 5, -1)
    st["base"]["skip_offset"] = np.where(df > 128, 1000, -1)
    st["last_pos_block_offset"] = -1
    tim, tip = oracle.blocktree_write([dict(number=0, doc_count=1000, terms=terms, states=st)])
    t0 = time.perf_counter()
    d = rgpu.TermDictionary(tim, tip, [(0, 2)], max_doc=10**7)
    open_s = time.perf_counter() - t0
    r = oracle.BlockTreeReader(tim, tip, [dict(number=0)], max_doc=10**7)
    rng = np.random.default_rng(1)
    probes = [terms[i] for i in rng.integers(0, n, 100_000)]
    offs = np.zeros(len(probes) + 1, dtype=np.int64)
    np.cumsum([len(p) for p in probes], out=offs[1:])
    flat = np.frombuffer(b"".join(probes), dtype=np.uint8)
    pst = np.zeros(len(probes), dtype=rgpu.TERM_STATE_DTYPE)
    ost = np.zeros(len(probes), dtype=oracle.FULL_TERM_STATE_DTYPE)
    found = np.zeros(len(probes), dtype=np.uint8)

    def best(fn, reps):
        times = []
        for _ in range(reps):
            t = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t)
        return min(times)
    tp = best(lambda: rgpu.lib().rgpu_terms_lookup(d._h, 0, flat.ctypes.data, offs.ctypes.data, len(probes), pst.ctypes.data,
                                                   found.ctypes.data), 5)
    assert found.all()
    to = best(lambda: oracle.lib().orc_blocktree_seek_exact(r._h, 0, flat.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                            offs.ctypes.data_as(C.POINTER(C.c_int64)), len(probes),
                                                            ost.ctypes.data_as(C.c_void_p),
                                                            found.ctypes.data_as(C.POINTER(C.c_uint8))), 2)
    assert found.all() and pst.tobytes() == np.ascontiguousarray(ost["base"]).tobytes()
    with capsys.disabled():
        print("\n[term dictionary] %d terms: open %.0f ms (%.1f M terms/s); lookups: product %.1f M/s, reference algorithm %.2f M/s (%.0fx)"
              % (n, open_s * 1e3, n / open_s / 1e6, len(probes) / tp / 1e6, len(probes) / to / 1e6, to / tp))
    assert to > 3 * tp
