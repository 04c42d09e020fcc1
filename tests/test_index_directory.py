"""The files that say what a Rucene index directory holds — the commit point "segments_N" and each segment's ".si" —
and opening a whole directory through them. Host-only code on both sides: the product readers are
rgpu_commit_from_segments_file / rgpu_segment_info_from_lucene62 (rucene_amd/csrc/host/segment_infos_format.hpp through the
C ABI), the checker is the oracle's restatement of SegmentInfos::{write_output, read_commit} and
Lucene62SegmentInfoFormat::{write, read} (oracle/segment_infos.hpp). No reference test pins these formats (parity
unpinned: the source text is the only authority); the two are checked against each other and hand-assembled bytes."""
import os
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
CID = bytes(range(100, 116))


def _header(codec, version, sid, suffix=b""):
    return struct.pack(">I", 0x3FD76C17) + bytes([len(codec)]) + codec + struct.pack(">i", version) + sid + bytes([len(suffix)]) + suffix


def _footer(body):
    f = body + struct.pack(">Ii", 0xC02893E8, 0)
    return f + struct.pack(">q", zlib.crc32(f) & 0xFFFFFFFF)


def test_hand_assembled_segment_info(rgpu, oracle):
    # segment_infos_format.rs:249-380: version triple (i32 x3), i32 max_doc, u8 compound, diagnostics, files, attributes, sort
    si = _footer(_header(b"Lucene62SegmentInfo", 1, SID) + struct.pack(">iiii", 6, 4, 18, 1234) + b"\xff" +
                 bytes([1]) + b"\x06source" + b"\x05flush" +
                 bytes([2]) + b"\x06_0.fnm" + b"\x05_0.si" +
                 bytes([0]) + bytes([0]))
    assert oracle.segment_info_write("_0", 1234, segment_id=SID, files=["_0.si", "_0.fnm"], diagnostics={"source": "flush"}) == si
    want = dict(max_doc=1234, is_compound_file=False, version=(6, 4, 18), n_files=2, id=SID)
    got = rgpu.segment_info_from_lucene62(si, expected_id=SID)
    assert {k: got[k] for k in want} == want and got["n_sort_fields"] == 0
    assert {k: oracle.segment_info_read(si, SID)[k] for k in want} == want
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.segment_info_from_lucene62(si, expected_id=CID)             # check_index_header_id
    assert e.value.status == -4
    with pytest.raises(oracle.OracleError):
        oracle.segment_info_read(si, CID)
    with pytest.raises(oracle.OracleError):                                 # a file of another segment
        oracle.segment_info_write("_0", 1, files=["_1.si"])


def test_index_sorted_and_compound_segment_info(rgpu, oracle):
    # the index-sort grammar (segment_infos_format.rs:65-200) is parsed to reach the footer: Long with a missing value, a
    # SortedNumeric (Float, Max) and a reversed String
    body = (_header(b"Lucene62SegmentInfo", 1, SID) + struct.pack(">iiii", 6, 4, 18, 77) + b"\x01" + bytes([0, 0, 0]) + bytes([3]) +
            b"\x01a" + bytes([1]) + bytes([1]) + bytes([1]) + struct.pack(">q", -5) +
            b"\x01b" + bytes([6, 3, 1]) + bytes([0]) + bytes([1]) + struct.pack(">i", 7) +
            b"\x01c" + bytes([0]) + bytes([0]) + bytes([0]))
    si = _footer(body)
    got = rgpu.segment_info_from_lucene62(si)
    assert got["is_compound_file"] and got["n_sort_fields"] == 3 and got["max_doc"] == 77
    assert oracle.segment_info_read(si)["is_compound_file"]
    bad = bytearray(body)
    bad[-3] = 9                                                            # sort type id of "c"
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.segment_info_from_lucene62(_footer(bytes(bad)))
    assert e.value.status == -4
    with pytest.raises(oracle.OracleError):
        oracle.segment_info_read(_footer(bytes(bad)))


def test_hand_assembled_segments_file(rgpu, oracle):
    # segment_infos.rs:243-300, generation 37 -> suffix "11"
    seg = (b"\x02_0" + b"\x01" + SID + b"\x08Lucene62" + struct.pack(">qiqq", 3, 12, -1, -1) + bytes([0]) + struct.pack(">i", 0))
    data = _footer(_header(b"segments", 6, CID, b"11") + bytes([6, 4, 18]) + struct.pack(">qii", 9, 5, 1) + bytes([6, 4, 18]) + seg + bytes([0]))
    segs = [dict(name="_0", id=SID, max_doc=100, del_gen=3, del_count=12)]
    assert oracle.segments_file_write(segs, generation=37, commit_id=CID, version=9, counter=5) == data
    want = [dict(name="_0", id=SID, del_gen=3, del_count=12, field_infos_gen=-1, dv_gen=-1)]
    assert oracle.segments_file_read(data, 37, max_docs=[100]) == want
    got = rgpu.commit_from_segments_file(data, 37)
    assert [{k: g[k] for k in want[0]} for g in got] == want and got[0]["codec"] == "Lucene62"
    for call in (lambda: rgpu.commit_from_segments_file(data, 36), lambda: oracle.segments_file_read(data, 36)):   # wrong generation
        with pytest.raises((rgpu.RgpuError, oracle.OracleError)):
            call()
    with pytest.raises(oracle.OracleError):
        oracle.segments_file_read(data, 37, max_docs=[11])                  # del_count > max_doc
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.commit_from_segments_file(data[:-1], 37)
    assert e.value.status == -4
    other_codec = data.replace(b"Lucene62", b"Lucene70")
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.commit_from_segments_file(_footer(other_codec[:-16]), 37)
    assert e.value.status == -2                                             # IllegalArgument: Invalid codec name
    assert rgpu.commit_from_segments_file(oracle.segments_file_write([], generation=1), 1) == []


def test_product_matches_oracle_on_many_segments(rgpu, oracle):
    rng = np.random.default_rng(8)
    segs = [dict(name="_%s" % np.base_repr(i, 36).lower(), id=bytes(rng.integers(0, 256, 16, dtype=np.uint8)), max_doc=int(rng.integers(1, 10**6)),
                 del_gen=int(rng.integers(-1, 50)), field_infos_gen=int(rng.integers(-1, 3)), dv_gen=-1) for i in range(70)]
    for s in segs:
        s["del_count"] = int(rng.integers(0, s["max_doc"] + 1)) if s["del_gen"] >= 0 else 0
    data = oracle.segments_file_write(segs, generation=1295)
    want = oracle.segments_file_read(data, 1295, max_docs=[s["max_doc"] for s in segs])
    got = rgpu.commit_from_segments_file(data, 1295)
    assert [{k: g[k] for k in want[0]} for g in got] == want
    assert [w["name"] for w in want] == [s["name"] for s in segs]


def _write_segment(oracle, path, name, sid, seg, live=None, del_gen=-1):
    """Lay one synthetic segment out as the files a Rucene directory would hold for it."""
    present = [t for t in range(seg.terms.size) if seg.terms[t]["doc_freq"] > 0]
    st = np.zeros(len(present), dtype=oracle.FULL_TERM_STATE_DTYPE)
    st["base"] = seg.terms[present]
    st["last_pos_block_offset"] = -1
    tim, tip = oracle.blocktree_write([dict(number=1, doc_count=seg.doc_count, terms=[b"w%05d" % t for t in present], states=st)],
                                      segment_id=sid, suffix="Lucene50_0")
    nvm, nvd = oracle.norms_write(seg.norms.astype(np.int64), field_number=1, segment_id=sid)
    # every file of a segment carries the segment's id: re-stamp the synthetic .doc (its generator picked its own) and re-seal it
    doc = bytearray(seg.doc_bytes.tobytes())
    at = 4 + 1 + len("Lucene50PostingsWriterDoc") + 4
    doc[at:at + 16] = sid
    doc[-8:] = struct.pack(">q", zlib.crc32(bytes(doc[:-8])) & 0xFFFFFFFF)
    files = {name + ".fnm": oracle.field_infos_write([dict(name="title", number=0), dict(name="body", number=1, index_options=2)], segment_id=sid),
             name + "_Lucene50_0.doc": bytes(doc), name + "_Lucene50_0.tim": tim, name + "_Lucene50_0.tip": tip,
             name + ".nvm": nvm, name + ".nvd": nvd}
    del_count = 0
    if live is not None:
        bits = np.unpackbits(live.view(np.uint8), bitorder="little")[:seg.max_doc]
        del_count = int(seg.max_doc - bits.sum())
        files["%s_%s.liv" % (name, np.base_repr(del_gen, 36).lower())] = oracle.live_docs_write(live, seg.max_doc, del_count, segment_id=sid, gen=del_gen)
    files[name + ".si"] = oracle.segment_info_write(name, seg.max_doc, segment_id=sid, files=sorted(files) + [name + ".si"])
    for fname, data in files.items():
        with open(os.path.join(path, fname), "wb") as fh:
            fh.write(data)
    return dict(name=name, id=sid, max_doc=seg.max_doc, del_gen=del_gen if live is not None else -1, del_count=del_count)


def build_directory(oracle, path, sizes=((30_000, 2_000), (12_000, 900)), deletions=True):
    """Two synthetic segments + an older commit point. -> (synthetic segments, live-doc words per segment)"""
    from rucene_amd import indexgen
    rng = np.random.default_rng(77)
    segs, lives, commit = [], [], []
    for i, (max_doc, n_terms) in enumerate(sizes):
        seg = indexgen.build_zipf(max_doc, n_terms, seed=500 + i)
        live = None
        if deletions and i == 0:
            bits = rng.random(max_doc) < 0.9
            live = np.zeros((max_doc + 63) // 64, dtype=np.uint64)
            idx = np.nonzero(bits)[0]
            np.bitwise_or.at(live, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
        sid = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        commit.append(_write_segment(oracle, path, "_%d" % i, sid, seg, live, del_gen=2))
        segs.append(seg)
        lives.append(live)
    with open(os.path.join(path, "segments_1"), "wb") as fh:                # an older commit: only the first segment
        fh.write(oracle.segments_file_write(commit[:1], generation=1))
    with open(os.path.join(path, "segments_2"), "wb") as fh:
        fh.write(oracle.segments_file_write(commit, generation=2))
    return segs, lives


def test_open_directory_host_side(rgpu, oracle, tmp_path):
    segs, lives = build_directory(oracle, str(tmp_path))
    leaves = rgpu.open_directory(str(tmp_path), field="body")
    assert [l.max_doc for l in leaves] == [30_000, 12_000] and [l.doc_base for l in leaves] == [0, 30_000]
    assert leaves[0].field_number == 1 and leaves[0].live_docs is not None and leaves[1].live_docs is None
    assert (leaves[0].live_docs == lives[0]).all() and (leaves[0].norms == segs[0].norms).all()
    for leaf, seg in zip(leaves, segs):
        for t in (0, 1, 57, seg.terms.size - 1):
            assert leaf.term_state(b"w%05d" % t).tobytes() == seg.terms[t].tobytes()
        assert leaf.term_state(b"nope") is None
        assert leaf.sum_total_term_freq == int(seg.terms["total_term_freq"].sum())
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.open_directory(str(tmp_path), field="title")                  # not indexed
    assert e.value.status == -5
    with pytest.raises(rgpu.RgpuError) as e:
        rgpu.open_directory(str(tmp_path), field="missing")
    assert e.value.status == -2
    # a later commit whose field infos were rewritten (doc-values update): _0_3.fnm replaces _0.fnm for that commit
    commit = oracle.segments_file_read(open(str(tmp_path / "segments_2"), "rb").read(), 2)
    fnm = open(str(tmp_path / "_0.fnm"), "rb").read()
    sid0 = commit[0]["id"]
    renumbered = oracle.field_infos_write([dict(name="title", number=0), dict(name="body", number=1, index_options=2),
                                           dict(name="views", number=2, doc_values_type=1, dv_gen=3)], segment_id=sid0)
    with open(str(tmp_path / "_0_3.fnm"), "wb") as fh:
        fh.write(renumbered)
    with open(str(tmp_path / "segments_3"), "wb") as fh:
        fh.write(oracle.segments_file_write([dict(commit[0], max_doc=30_000, field_infos_gen=3, dv_gen=3), dict(commit[1], max_doc=12_000)],
                                            generation=3))
    os.remove(str(tmp_path / "_0.fnm"))                                    # only the generation file is left for segment _0
    assert [l.field_number for l in rgpu.open_directory(str(tmp_path), field="body")] == [1, 1]
    with open(str(tmp_path / "_0.fnm"), "wb") as fh:
        fh.write(fnm)
    os.remove(str(tmp_path / "segments_3"))
    os.remove(str(tmp_path / "segments_2"))
    assert len(rgpu.open_directory(str(tmp_path), field="body")) == 1    # falls back to the older commit point
    os.remove(str(tmp_path / "segments_1"))
    with pytest.raises(rgpu.RgpuError):
        rgpu.open_directory(str(tmp_path))


# ---- compound segments (".cfs" + ".cfe") ---------------------------------------------------------------------------------
def _compound_directory(oracle, path):
    """The two-segment directory again, with the second segment packed into a compound file the way IndexWriter does after a
    flush or merge: .si (is_compound_file = yes, files = .cfs/.cfe/.si) and .liv stay outside."""
    segs, lives = build_directory(oracle, path)
    name = "_1"
    inner = {f: open(os.path.join(path, f), "rb").read() for f in os.listdir(path) if f.startswith(name + ".") or f.startswith(name + "_")}
    si = inner.pop(name + ".si")
    sid = si[4 + 1 + len("Lucene62SegmentInfo") + 4:][:16]
    cfs, cfe = oracle.compound_write(inner, sid)
    for f in inner:
        os.remove(os.path.join(path, f))
    with open(os.path.join(path, name + ".cfs"), "wb") as fh:
        fh.write(cfs)
    with open(os.path.join(path, name + ".cfe"), "wb") as fh:
        fh.write(cfe)
    with open(os.path.join(path, name + ".si"), "wb") as fh:
        fh.write(oracle.segment_info_write(name, segs[1].max_doc, segment_id=sid, files=[name + ".cfs", name + ".cfe", name + ".si"],
                                           is_compound_file=True))
    return segs, lives, inner, sid, cfs, cfe


def test_compound_files(rgpu, oracle, tmp_path):
    segs, lives, inner, sid, cfs, cfe = _compound_directory(oracle, str(tmp_path))
    want = oracle.compound_read(cfe, cfs, sid)
    got = rgpu.compound_files_from_lucene50(cfe, cfs, expected_id=sid)
    assert sorted(got) == sorted(want) == sorted(n[2:] for n in inner)          # "_1.fnm" -> ".fnm", "_1_Lucene50_0.doc" -> "_Lucene50_0.doc"
    for n, data in inner.items():
        off, ln = want[n[2:]]
        assert cfs[off:off + ln] == data and got[n[2:]] == data                    # each file is inside, whole
    # the directory opens as before: the compound segment's files come out of its .cfs
    leaves = rgpu.open_directory(str(tmp_path), field="body")
    assert [l.max_doc for l in leaves] == [30_000, 12_000]
    for t in (0, 5, 899):
        assert leaves[1].term_state(b"w%05d" % t).tobytes() == segs[1].terms[t].tobytes()
    assert (leaves[1].norms == segs[1].norms).all()
    # rejections
    for bad_cfe, bad_cfs in ((cfe[:-1], cfs), (cfe, cfs[:-1]), (cfe, cfs + b"\\0"), (cfe[:-8] + bytes(8), cfs)):
        with pytest.raises(rgpu.RgpuError) as e:
            rgpu.compound_files_from_lucene50(bad_cfe, bad_cfs, expected_id=sid)
        assert e.value.status == -4
        with pytest.raises(oracle.OracleError):
            oracle.compound_read(bad_cfe, bad_cfs, sid)
    with pytest.raises(rgpu.RgpuError):
        rgpu.compound_files_from_lucene50(cfe, cfs, expected_id=bytes(16))
    with pytest.raises(oracle.OracleError):                                        # a file of another segment cannot be packed
        oracle.compound_write({"_1.fnm": oracle.field_infos_write([], segment_id=bytes(range(16)))}, sid)


def test_cpp_mirror_opens_directories(rgpu, oracle, tmp_path):
    """rucene::IndexDirectory::open (the C++ host mirror, header-only over the C ABI) reads the same directories as the Python
    mirror: plain and compound segments, deletions, term resolution by bytes. Host-only: no GPU is touched."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rucene_amd")
    exe = str(tmp_path / "open_dir")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-o", exe, os.path.join(root, "tests", "cpp", "open_directory_demo.cpp"),
                           "-L" + libdir, "-lrucene_gpu", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    d = tmp_path / "idx"
    d.mkdir()
    segs, lives, inner, sid, cfs, cfe = _compound_directory(oracle, str(d))
    out = subprocess.check_output([exe, str(d), "body", "w00000", "w00007", "w00899", "w01500", "nope"], text=True).strip().splitlines()
    leaves = rgpu.open_directory(str(d), field="body")
    assert len(out) == 2
    for line, leaf, seg in zip(out, leaves, segs):
        parts = line.split()
        assert [int(x) for x in parts[1:8]] == [leaf.max_doc, leaf.doc_base, leaf.doc_count, leaf.sum_total_term_freq, leaf.sum_doc_freq,
                                                 leaf.field_number, int(leaf.live_docs is not None)]
        for tok, t in zip(parts[8:], (0, 7, 899, 1500, None)):
            st = None if t is None or t >= seg.terms.size else seg.terms[t]
            assert tok == ("-" if st is None else "%d@%d" % (st["doc_freq"], st["doc_start_fp"]))
    golden = os.path.join(root, "tests", "golden", "dir")
    line = subprocess.check_output([exe, golden, "body", "w000", "w039", "w040"], text=True).split()
    assert line[1:4] == ["3000", "0", "3000"] and line[-1] == "-" and line[7] == "1"
    bad = subprocess.run([exe, golden, "stored"], capture_output=True, text=True)
    assert bad.returncode == 1 and bad.stdout.startswith("error -5")


def test_randomised_metadata_files_product_vs_oracle(rgpu, oracle):
    """Differential check on many VALID files of random shape: names of 0..300 bytes (vint lengths of 1 and 2 bytes), non-ASCII
    names, attribute maps, many fields / segments, every flag combination the writers allow."""
    import random
    rng = random.Random(99)

    def name(lo=1):
        n = rng.choice([lo, 1, 5, 127, 128, 300])
        return "".join(rng.choice("abcXYZ_09-é中") for _ in range(max(n, lo)))
    for _ in range(60):
        # .fnm
        fields, used = [], set()
        for number in sorted(rng.sample(range(0, 5000), rng.randint(0, 25))):
            nm = name()
            while nm in used:
                nm = name()
            used.add(nm)
            opts = rng.choice([0, 1, 2, 3, 4])
            dvt = rng.choice([0, 0, 1, 2, 3, 4, 5])
            fields.append(dict(name=nm, number=number, index_options=opts, store_term_vector=opts > 0 and rng.random() < 0.3,
                               omit_norms=rng.random() < 0.3, store_payloads=opts >= 3 and rng.random() < 0.5, doc_values_type=dvt,
                               dv_gen=rng.choice([-1, 7]) if dvt else -1,
                               attributes={name(0): name(0) for _ in range(rng.randint(0, 3))},
                               point_dimension_count=rng.choice([0, 0, 2]), point_num_bytes=0))
            if fields[-1]["point_dimension_count"]:
                fields[-1]["point_num_bytes"] = rng.choice([4, 8])
        sid = bytes(rng.randrange(256) for _ in range(16))
        fnm = oracle.field_infos_write(fields, segment_id=sid, suffix=rng.choice(["", "x"]))
        want = oracle.field_infos_read(fnm)
        got = rgpu.field_infos_from_lucene60(fnm)
        assert [(g["name"], g["number"], g["index_options"], g["has_payloads"], g["omit_norms"], g["store_term_vector"], g["doc_values_type"])
                for g in got] == [(w["name"], w["number"], w["index_options"], w["store_payloads"], w["omit_norms"], w["store_term_vector"],
                                   w["doc_values_type"]) for w in want]
        # .si
        seg_name = "_" + np.base_repr(rng.randrange(36 ** 3), 36).lower()
        files = sorted({seg_name + rng.choice([".fnm", ".nvd", "_Lucene50_0.doc", ".cfs", "_1.liv", ".si"]) for _ in range(rng.randint(0, 6))})
        si = oracle.segment_info_write(seg_name, rng.randrange(0, 2 ** 31 - 1), segment_id=sid, files=files, is_compound_file=rng.random() < 0.5,
                                       version=(rng.randint(5, 9), rng.randint(0, 255), rng.randint(0, 255)),
                                       diagnostics={name(0): name(0) for _ in range(rng.randint(0, 4))},
                                       attributes={name(0): name(0) for _ in range(rng.randint(0, 2))})
        w = oracle.segment_info_read(si, sid)
        g = rgpu.segment_info_from_lucene62(si, expected_id=sid)
        assert {k: g[k] for k in ("max_doc", "is_compound_file", "version", "n_files", "id")} == {k: w[k] for k in ("max_doc", "is_compound_file", "version", "n_files", "id")}
        # segments_N
        gen = rng.choice([1, 35, 36, 1295, 1296, 10 ** 9])
        segs = []
        for i in range(rng.randint(0, 40)):
            md = rng.randrange(1, 10 ** 7)
            dg = rng.choice([-1, -1, 1, 40])
            segs.append(dict(name="_" + np.base_repr(i, 36).lower(), id=bytes(rng.randrange(256) for _ in range(16)), max_doc=md, del_gen=dg,
                             del_count=rng.randrange(0, md + 1) if dg > 0 else 0, field_infos_gen=rng.choice([-1, 2]), dv_gen=rng.choice([-1, 2]),
                             version=(rng.randint(5, 6), rng.randint(0, 9), 0)))
        data = oracle.segments_file_write(segs, generation=gen, version=rng.randrange(2 ** 40), counter=len(segs))
        w = oracle.segments_file_read(data, gen, max_docs=[s["max_doc"] for s in segs])
        g = rgpu.commit_from_segments_file(data, gen)
        assert [{k: x[k] for k in w[0]} for x in g] == w if w else g == []
