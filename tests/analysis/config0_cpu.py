"""BASELINE.json configs[0]: single-term BM25 queries over a 100k-doc synthetic index on the CPU IndexSearcher — here the
oracle (C++ restatement; the Rust original cannot be built in this image). No GPU involved. usage: tests/analysis/config0_cpu.py [threads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from rucene_amd import indexgen
from oracle import binding as orc
threads = int(sys.argv[1]) if len(sys.argv) > 1 else os.cpu_count()
seg = indexgen.build_zipf(100_000, 1_000_000)
s = orc.Searcher([orc.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)])
tids = (indexgen.log_uniform_ranks(1024, 1, 10_000, 0x527563656E65 ^ 0x51) - 1).astype(np.int64)
ops = np.full(1024, orc.OP_TERM, np.int32)
offs = np.arange(1025, dtype=np.int32)
postings = int(seg.terms["doc_freq"][tids].sum())
for t in (1, threads):
    spent, reps = 0.0, 0
    while spent < 2.0:
        *_, secs = s.search_batch(ops, offs, tids, 10, tie_mode=orc.TIE_RUST_HEAP, threads=t)
        spent += secs
        reps += 1
    print("config 0: %d docs, 1024 single-term queries, k = 10, %d thread(s): %.0f queries/s, %.1f M postings/s (%d postings per batch)"
          % (seg.max_doc, t, 1024 * reps / spent, postings * reps / spent / 1e6, postings))
