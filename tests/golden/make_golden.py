"""Regenerates tests/golden/*: a tiny ".doc" v1 + v0 file pair written by the ORACLE's line-faithful
Lucene50PostingsWriter restatement (oracle/postings.hpp), the postings that went in, and the reference's own
known-answer vectors restated as data (the table of SURVEY.md §4).

The Rust reference cannot run in this image (no rustc; nightly-2020-03-12; un-vendored crates), so these fixtures
are NOT outputs of the reference itself: they freeze the oracle's behaviour so that a later change to the oracle
(or to the generator / kernels) that alters a single byte is caught. Run from the repo root:

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def postings():
    rng = np.random.default_rng(20260921)
    lists = []
    for df in (1, 2, 127, 128, 129, 300, 1025, 2100):
        docs = np.sort(rng.choice(40_000, size=df, replace=False)).astype(np.int32)
        freqs = np.minimum(10, rng.geometric(0.5, size=df)).astype(np.int32)
        lists.append((docs, freqs))
    lists.append((np.arange(7, 7 + 4 * 200, 4, dtype=np.int32), np.ones(200, np.int32)))  # all-equal doc + freq blocks
    return lists


def main():
    from oracle import binding as orc
    lists = postings()
    meta = {"max_doc": 40_000, "terms": []}
    for version in (1, 0):
        w = orc.Writer(40_000, version=version, segment_id=bytes(range(16)))
        states = [w.write_term(d, f) for d, f in lists]
        raw = w.close()
        raw.tofile(os.path.join(HERE, "golden_v%d.doc" % version))
        if version == 1:
            meta["terms"] = [{k: int(st[k]) for k in st.dtype.names} for st in states]
    np.savez_compressed(os.path.join(HERE, "golden_postings.npz"),
                        **{"docs_%d" % i: d for i, (d, _) in enumerate(lists)},
                        **{"freqs_%d" % i: f for i, (_, f) in enumerate(lists)})
    # reference known-answer vectors (file:line in /root/reference/src/core) as data
    meta["reference_kat"] = {
        "packed_simd.rs:507-521": {"values": "128*(i+1), i=0..127", "plain_bits": 15, "delta_bits": 14, "delta_base": 128},
        "partial_block_decoder.rs:128-141": {"bytes": [255, 255, 0, 255], "bits": 4, "values": [15, 15, 15, 15, 0, 0, 15, 15]},
        "partial_block_decoder.rs:143-152": {"bytes": [255, 15, 0, 0, 0, 0, 255, 0, 143, 255, 143, 143, 143, 143, 143, 143],
                                             "bits": 6, "format": "PackedSingleBlock", "values_at": {"0": 0, "1": 60, "9": 60, "10": 15}},
        "for_util.rs:42": {"MAX_DATA_SIZE": 147},
        "bm25_similarity.rs:413-428": {"idf(df=1,maxDoc=11,docCount=-1)": "ln 8", "idf(df=1,docCount=32)": "ln 22"},
        "bm25_similarity.rs:442-449": {"N": 32, "docCount": 32, "sumTTF": 120, "df": 1, "weight_squared": 9.5545435},
        "conjunction_scorer.rs:162-222": {"lists": [[1, 2, 3, 4, 5], [2, 5], [2, 3, 4, 5]], "docs": [2, 5], "scores": [6.0, 15.0]},
        "top_docs.rs:235-264": {"docs": [1, 2, 3, 3, 5], "k": 3, "top": [5, 3, 3], "total_hits": 5},
        "bulk_scorer.rs:167-200": {"docs": [1, 2, 3, 4, 5], "k": 3, "top": [5, 4, 3]},
        "searcher.rs:916-952": {"leaves": 3, "docs": [1, 5, 3, 4, 2], "early_terminate_after": 3, "total_hits": 9, "top_scores": [5, 5, 5]},
    }
    json.dump(meta, open(os.path.join(HERE, "golden_meta.json"), "w"), indent=1, sort_keys=True)
    write_directory(orc)
    write_positions(orc)
    print("wrote", sorted(os.listdir(HERE)))


def directory_postings():
    """A tiny docs+freqs segment: 40 terms named b"w%03d", 3000 docs, deletions. Deterministic."""
    rng = np.random.default_rng(20260922)
    max_doc = 3000
    lists = []
    for t in range(40):
        df = int([1, 2, 3, 50, 127, 128, 129, 200, 400, 1500][t % 10])
        docs = np.sort(rng.choice(max_doc, size=df, replace=False)).astype(np.int32)
        freqs = np.minimum(10, rng.geometric(0.5, size=df)).astype(np.int32)
        lists.append((docs, freqs))
    norms = rng.integers(97, 125, size=max_doc).astype(np.uint8)
    bits = rng.random(max_doc) < 0.9
    live = np.zeros((max_doc + 63) // 64, dtype=np.uint64)
    idx = np.nonzero(bits)[0]
    np.bitwise_or.at(live, idx >> 6, np.uint64(1) << (idx & 63).astype(np.uint64))
    return max_doc, lists, norms, live, int(max_doc - bits.sum())


def write_directory(orc):
    """tests/golden/dir: segments_1, _0.si, _0.fnm, _0_Lucene50_0.{doc,tim,tip}, _0.nvm, _0.nvd, _0_1.liv — every file the
    restated writers produce for one segment, frozen."""
    out = os.path.join(HERE, "dir")
    os.makedirs(out, exist_ok=True)
    for f in os.listdir(out):
        os.remove(os.path.join(out, f))
    max_doc, lists, norms, live, del_count = directory_postings()
    sid = bytes(range(200, 216))
    w = orc.Writer(max_doc, version=1, segment_id=sid)
    states = np.array([w.write_term(d, f) for d, f in lists])
    doc = w.close().tobytes()
    st = np.zeros(len(lists), dtype=orc.FULL_TERM_STATE_DTYPE)
    st["base"] = states
    st["last_pos_block_offset"] = -1
    tim, tip = orc.blocktree_write([dict(number=2, doc_count=max_doc, terms=[b"w%03d" % t for t in range(len(lists))], states=st)],
                                   5, 10, segment_id=sid, suffix="Lucene50_0")    # small blocks: a real tree with floor blocks
    nvm, nvd = orc.norms_write(norms.astype(np.int64), field_number=2, segment_id=sid)
    files = {"_0.fnm": orc.field_infos_write([dict(name="id", number=0, index_options=1, omit_norms=True), dict(name="stored", number=1),
                                              dict(name="body", number=2, index_options=2)], segment_id=sid),
             "_0_Lucene50_0.doc": doc, "_0_Lucene50_0.tim": tim, "_0_Lucene50_0.tip": tip, "_0.nvm": nvm, "_0.nvd": nvd,
             "_0_1.liv": orc.live_docs_write(live, max_doc, del_count, segment_id=sid, gen=1)}
    files["_0.si"] = orc.segment_info_write("_0", max_doc, segment_id=sid, files=sorted(files) + ["_0.si"], diagnostics={"source": "flush"})
    files["segments_1"] = orc.segments_file_write([dict(name="_0", id=sid, max_doc=max_doc, del_gen=1, del_count=del_count)], generation=1)
    for name, data in files.items():
        with open(os.path.join(out, name), "wb") as fh:
            fh.write(data)
    np.save(os.path.join(HERE, "dir_states.npy"), states)


def positions_postings():
    rng = np.random.default_rng(20260923)
    postings = []
    for df in (1, 3, 128, 129, 300):
        docs = np.sort(rng.choice(5000, size=df, replace=False)).tolist()
        postings.append([(d, np.sort(rng.choice(400, size=int(rng.integers(1, 9)), replace=False)).tolist()) for d in docs])
    return postings


def write_positions(orc):
    """tests/golden/pos.doc + pos.pos: a positions field written by the restated writer, frozen."""
    ix = orc.PositionsIndex(5000, positions_postings())
    doc, pos = ix.files()
    open(os.path.join(HERE, "pos.doc"), "wb").write(doc)
    open(os.path.join(HERE, "pos.pos"), "wb").write(pos)


if __name__ == "__main__":
    main()
