"""Lucene53 norms files (".nvm" + ".nvd") -> the norms array the scoring kernels take. Host-only code on both sides:
the product reader is rgpu_norms_from_lucene53 (rucene_amd/csrc/host/norms_format.hpp, through the C ABI), the checker
is the oracle's restatement of Lucene53NormsConsumer / Lucene53NormsProducer (oracle/norms.hpp). The reference holds no
test for these files (parity unpinned: the source text is the only authority), so the two independent
implementations are checked against each other and against hand-assembled bytes."""
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


def _header(codec, version=0, sid=bytes(range(16)), suffix=b""):
    return struct.pack(">I", 0x3FD76C17) + bytes([len(codec)]) + codec + struct.pack(">i", version) + sid + bytes([len(suffix)]) + suffix


def _footer(body):
    f = body + struct.pack(">Ii", 0xC02893E8, 0)
    return f + struct.pack(">q", zlib.crc32(f) & 0xFFFFFFFF)


def test_hand_assembled_files(rgpu, oracle):
    # one byte per doc, field 3, offset right after the data header (norms_consumer.rs:84-113, :117-147, :150-158)
    vals = bytes([110, 97, 124, 101, 110])
    nvd_head = _header(b"Lucene53NormsData")
    nvd = _footer(nvd_head + vals)
    nvm = _footer(_header(b"Lucene53NormsMetadata") + bytes([3, 1]) + struct.pack(">q", len(nvd_head)) + b"\xff\xff\xff\xff\x0f")
    assert rgpu.norms_from_lucene53(nvm, nvd, 3, 5).tolist() == list(vals)
    assert oracle.norms_read(nvm, nvd, 3, 5).tolist() == list(vals)
    # the writer restatement produces exactly these bytes
    wm, wd = oracle.norms_write(list(vals), field_number=3)
    assert wm == nvm and wd == nvd


@pytest.mark.parametrize("kind", ["constant", "i8", "u8_range", "i16", "i32", "i64"])
def test_round_trip_all_widths(rgpu, oracle, kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    n = 1000
    vals = {
        "constant": np.full(n, 117, np.int64),
        "i8": rng.integers(97, 125, n),                 # what BM25 norms of real lengths look like: 1 byte/doc
        "u8_range": rng.integers(0, 256, n),            # bytes >= 128 no longer fit i8 -> 2 bytes per value
        "i16": rng.integers(-30000, 30000, n),
        "i32": rng.integers(-2**31, 2**31 - 1, n),
        "i64": rng.integers(-2**62, 2**62, n),
    }[kind].astype(np.int64)
    nvm, nvd = oracle.norms_write(vals, field_number=7, segment_id=bytes(range(100, 116)), suffix="Lucene53_0")
    width = {"constant": 0, "i8": 1, "u8_range": 2, "i16": 2, "i32": 4, "i64": 8}[kind]
    assert len(nvd) == len(_header(b"Lucene53NormsData", sid=bytes(16), suffix=b"Lucene53_0")) + width * n + 16
    assert (oracle.norms_read(nvm, nvd, 7, n) == vals).all()
    got = rgpu.norms_from_lucene53(nvm, nvd, 7, n)
    assert (got == (vals & 0xFF).astype(np.uint8)).all()


def test_generated_segment_norms_survive_the_files(rgpu, oracle):
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(50_000, 5_000)
    nvm, nvd = oracle.norms_write(seg.norms.astype(np.int64), field_number=0)
    assert (rgpu.norms_from_lucene53(nvm, nvd, 0, seg.max_doc) == seg.norms).all()


def test_corrupt_files_are_rejected(rgpu, oracle):
    vals = np.arange(100, dtype=np.int64) % 50 + 60
    nvm, nvd = oracle.norms_write(vals, field_number=2)
    ok = rgpu.norms_from_lucene53(nvm, nvd, 2, 100)
    assert (ok == vals).all()

    def status(m, d, field=2, n=100):
        with pytest.raises(rgpu.RgpuError) as e:
            rgpu.norms_from_lucene53(m, d, field, n)
        return e.value.status

    flip = lambda b, i: b[:i] + bytes([b[i] ^ 0x40]) + b[i + 1:]
    assert status(flip(nvm, 0), nvd) == -4                    # magic
    assert status(flip(nvm, 8), nvd) == -4                    # codec name
    assert status(flip(nvm, len(nvm) - 20), nvd) == -4        # entry byte changed -> checksum (or framing) fails
    assert status(flip(nvm, len(nvm) - 1), nvd) == -4         # stored CRC
    assert status(nvm[:-3], nvd) in (-3, -4)                  # truncated metadata
    assert status(nvm, flip(nvd, 0)) == -4
    assert status(nvm, nvd[:40]) in (-3, -4)                  # truncated data: footer misplaced
    assert status(nvm, nvd, field=5) == -2                    # no such field
    assert status(nvm, nvd, n=101) == -3                      # slice runs past the data
    other_m, other_d = oracle.norms_write(vals, field_number=2, segment_id=bytes(range(50, 66)))
    assert status(nvm, other_d) == -4                         # files of different segments
    # bytes_per_value outside {0,1,2,4,8}
    head = _header(b"Lucene53NormsMetadata")
    bad = _footer(head + bytes([2, 3]) + struct.pack(">q", 41) + b"\xff\xff\xff\xff\x0f")
    assert status(bad, nvd) == -4


# ---- Lucene50LiveDocsFormat (".liv") ----------------------------------------------------------------------------------------
def test_live_docs_file_round_trip_and_hand_assembled(rgpu, oracle):
    rng = np.random.default_rng(12)
    for max_doc in (1, 63, 64, 65, 1000, 4096, 50_001):
        bits = rng.random(max_doc) < 0.9
        words = np.zeros((max_doc + 63) // 64, dtype=np.uint64)
        idx = np.nonzero(bits)[0]
        np.bitwise_or.at(words, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
        dels = int(max_doc - bits.sum())
        liv = oracle.live_docs_write(words, max_doc, dels, gen=37)
        # header: codec Lucene50LiveDocs, version 0, suffix = base36(gen) = "11"; then big-endian words; then the footer
        head = _header(b"Lucene50LiveDocs", suffix=b"11")
        assert liv == _footer(head + b"".join(struct.pack(">Q", int(w)) for w in words))
        assert (rgpu.live_docs_from_lucene50(liv, max_doc, dels) == words).all()
        assert (rgpu.live_docs_from_lucene50(liv, max_doc) == words).all()        # del_count unknown: not checked
        assert (oracle.live_docs_read(liv, max_doc, dels) == words).all()


def test_live_docs_file_corruption(rgpu, oracle):
    max_doc = 200
    words = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    words[3] = np.uint64((1 << (200 - 192)) - 1)
    words[1] &= ~np.uint64(0b1011)
    liv = oracle.live_docs_write(words, max_doc, 3)

    def status(b, n=max_doc, dc=3):
        with pytest.raises(rgpu.RgpuError) as e:
            rgpu.live_docs_from_lucene50(b, n, dc)
        return e.value.status

    assert status(liv, dc=2) == -4                                   # bits.deleted != info.delcount
    assert status(liv[:-1]) in (-3, -4)
    assert status(liv[:5] + b"X" + liv[6:]) == -4                    # codec name
    body = bytearray(liv)
    body[len(_header(b"Lucene50LiveDocs", suffix=b"1")) + 3] ^= 1    # a payload bit: checksum fails
    assert status(bytes(body)) == -4
    assert status(liv, n=500) == -3                                  # file too short for that many docs
    ghost = words.copy()
    ghost[3] |= np.uint64(1 << 20)                                   # a bit past max_doc
    head = _header(b"Lucene50LiveDocs", suffix=b"1")
    assert status(_footer(head + b"".join(struct.pack(">Q", int(w)) for w in ghost)), dc=-1) == -4
