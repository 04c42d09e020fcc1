"""Lucene60 field infos (".fnm") -> field name / number / index options, the keys the term dictionary and the norms
reader are addressed by. Host-only code on both sides: the product reader is rgpu_field_infos_from_lucene60
(rucene_amd/csrc/host/field_infos_format.hpp through the C ABI), the checker is the oracle's restatement of
Lucene60FieldInfosFormat::{write, read} (oracle/field_infos.hpp). The reference holds no test for this file (parity
unpinned: the source text is the only authority), so the two are checked against each other and against
hand-assembled bytes."""
import struct
import zlib

import pytest


@pytest.fixture(scope="module")
def rgpu():
    import __graft_entry__ as g
    g.build()
    import rucene_amd
    return rucene_amd


SID = bytes(range(16))


def _header(codec, version=0, sid=SID, suffix=b""):
    return struct.pack(">I", 0x3FD76C17) + bytes([len(codec)]) + codec + struct.pack(">i", version) + sid + bytes([len(suffix)]) + suffix


def _footer(body):
    f = body + struct.pack(">Ii", 0xC02893E8, 0)
    return f + struct.pack(">q", zlib.crc32(f) & 0xFFFFFFFF)


def _hand_assembled():
    # field_infos_format.rs:214-259: string name, vint number, bits, index options, doc values type, i64 dv_gen, attribute map,
    # vint point dimension count [, vint point num bytes]
    body = (_header(b"Lucene60FieldInfos") + bytes([2]) +
            b"\x04body" + bytes([0, 0x00, 2, 0]) + struct.pack(">q", -1) +
            bytes([2]) + b"\x1dPerFieldPostingsFormat.format" + b"\x08Lucene50" + b"\x1dPerFieldPostingsFormat.suffix" + b"\x010" +
            bytes([0]) +
            b"\x02id" + bytes([3, 0x02, 1, 1]) + struct.pack(">q", -1) + bytes([0]) + bytes([1, 8]))
    return _footer(body)


def test_hand_assembled_file(rgpu, oracle):
    fnm = _hand_assembled()
    fields = [dict(name="body", number=0, index_options=2,
                   attributes={"PerFieldPostingsFormat.format": "Lucene50", "PerFieldPostingsFormat.suffix": "0"}),
              dict(name="id", number=3, index_options=1, omit_norms=True, doc_values_type=1, point_dimension_count=1, point_num_bytes=8)]
    assert oracle.field_infos_write(fields, segment_id=SID) == fnm
    got = oracle.field_infos_read(fnm)
    assert [g["name"] for g in got] == ["body", "id"] and got[0]["attributes"] == fields[0]["attributes"]
    assert got[1]["omit_norms"] and got[1]["point_num_bytes"] == 8 and got[1]["dv_gen"] == -1
    assert rgpu.field_infos_from_lucene60(fnm) == [
        dict(name="body", number=0, index_options=2, has_payloads=False, omit_norms=False, store_term_vector=False, doc_values_type=0),
        dict(name="id", number=3, index_options=1, has_payloads=False, omit_norms=True, store_term_vector=False, doc_values_type=1)]


def test_product_matches_oracle(rgpu, oracle):
    fields = [dict(name="título", number=7, index_options=4, store_payloads=True, store_term_vector=True, attributes={"k": "", "": "v"}),
              dict(name="b", number=1, index_options=3),
              dict(name="stored-only", number=2),
              dict(name="dv", number=40_000, doc_values_type=5, dv_gen=12),
              dict(name="x" * 300, number=5, index_options=1, omit_norms=True)]
    fnm = oracle.field_infos_write(fields, segment_id=bytes(range(50, 66)), suffix="")
    want = oracle.field_infos_read(fnm)
    assert [w["number"] for w in want] == [1, 2, 5, 7, 40_000]          # by_number order
    got = rgpu.field_infos_from_lucene60(fnm)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert (g["name"], g["number"], g["index_options"], g["has_payloads"], g["omit_norms"], g["store_term_vector"], g["doc_values_type"]) == \
               (w["name"], w["number"], w["index_options"], w["store_payloads"], w["omit_norms"], w["store_term_vector"], w["doc_values_type"])
    assert rgpu.field_infos_from_lucene60(oracle.field_infos_write([])) == []


def test_rejects_what_the_reference_rejects(rgpu, oracle):
    good = _hand_assembled()

    def status(b):
        with pytest.raises(rgpu.RgpuError) as e:
            rgpu.field_infos_from_lucene60(b)
        with pytest.raises(oracle.OracleError):
            oracle.field_infos_read(b)
        return e.value.status

    assert status(good[:-1]) == -4                                      # footer not where it must be
    assert status(good[:-8] + bytes(8)) == -4                           # checksum
    flipped = bytearray(good); flipped[60] ^= 0x40
    assert status(bytes(flipped)) in (-1, -3, -4)
    body = good[:-16]
    head = len(_header(b"Lucene60FieldInfos"))
    bad_opts = bytearray(body); bad_opts[head + 1 + 5 + 2] = 9          # IndexOptions byte of "body"
    assert status(_footer(bytes(bad_opts))) == -4
    payload_without_positions = bytearray(body); payload_without_positions[head + 1 + 5 + 1] = 0x04
    assert status(_footer(bytes(payload_without_positions))) == -1      # FieldInfo::check_consistency -> IllegalState
    dup = _footer(_header(b"Lucene60FieldInfos") + bytes([2]) + (b"\x01a" + bytes([0, 0, 1, 0]) + struct.pack(">q", -1) + bytes([0, 0])) * 2)
    assert status(dup) == -2                                            # FieldInfos::new -> IllegalArgument
    assert status(_footer(_header(b"Lucene50FieldInfos") + bytes([0]))) == -4
    with pytest.raises(oracle.OracleError):                             # the writer checks consistency too
        oracle.field_infos_write([dict(name="p", number=0, index_options=2, store_payloads=True)])
