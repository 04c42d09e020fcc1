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
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
