"""Minimal driver for profiling: build the 10M-doc shard, run one workload a few times through the C ABI.
usage: run_workload.py [term|and3|and2sparse|mustor|mustand|or10|decode|cold|posdec|phrase2|sloppy2] [reps]   (DOCS=... sets the shard size; cold = a fresh segment per repetition:
skip decode + block framing + alignment + tails (k_prepare_terms, k_prepare_blocks), then k_decode_terms, for every df >= 128 term)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rucene_amd
from rucene_amd import indexgen
kind = sys.argv[1] if len(sys.argv) > 1 else "term"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
docs = int(os.environ.get("DOCS", "10000000"))
seg = indexgen.build_zipf(docs, 1_000_000)
ctx = rucene_amd.Context(profile_kernels=True, blocks_per_item=int(os.environ.get('BPI', '0')), and_blocks_per_item=int(os.environ.get('ABPI', '0')), or_window_docs=int(os.environ.get("ORW", "0")), or_dense_clauses=int(os.environ.get("ORD", "0")),
                         or_wide=int(os.environ.get("ORWIDE", "0")), or_wide_window_docs=int(os.environ.get("ORWW", "0")),
                         or_bitmaps=int(os.environ.get("ORBM", "0")), or_lazy_cells=int(os.environ.get("ORCELLS", "0")), and_bitmaps=int(os.environ.get("ANDBM", "0")))
leaf = rucene_amd.LeafReader.from_synthetic(seg)
if os.environ.get('NONORMS'):
    leaf.norms = None
s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
T, B = rucene_amd.TermQuery, rucene_amd.BooleanQuery
SEED = 0x527563656E65 ^ 0x51
if kind in ("posdec", "phrase2", "sloppy2"):
    # the positions side: the same corpus indexed with positions
    seg = indexgen.build_zipf(docs, 1_000_000, positions=True)
    leaf = rucene_amd.LeafReader.from_synthetic_positions(seg)
    s = rucene_amd.GpuIndexSearcher([leaf], ctx=ctx)
    leaf.segment.attach_positions(leaf.pos_bytes)
    leaf._pos_attached = True
    if kind == "posdec":
        import torch
        keep = seg.terms["doc_freq"] >= 128
        sel, selp = seg.terms[keep], leaf.term_positions[keep]
        tp = torch.empty(int(sel["total_term_freq"].sum()), dtype=torch.int32, device="cuda")
        for _ in range(reps + 2):
            leaf.segment.decode_positions_device(sel, selp, tp.data_ptr())
    else:
        ranks = indexgen.log_uniform_ranks(2 * 1024, 1, 1000, SEED ^ 0xF2).reshape(-1, 2) - 1
        # (sloppy2: the same pairs with slop 2 — SloppyPhraseScorer, one candidate per wavefront)
        qs, ts = s.pack_phrases([rucene_amd.PhraseQuery([int(a), int(b)], slop=2 if kind == "sloppy2" else 0) for a, b in ranks], leaf)
        for _ in range(reps + 2):
            leaf.segment.search_phrase_batch(qs, ts, 10)
elif kind == "cold":
    sel = seg.terms[seg.terms["doc_freq"] >= 128]
    import torch  # device buffers only
    total = int(sel["doc_freq"].sum())
    td = torch.empty(total, dtype=torch.int32, device="cuda")
    tf = torch.empty(total, dtype=torch.int32, device="cuda")
    for r in range(reps + 1):
        if r == 1:
            ctx.kernel_stats_reset()   # (the process's first launch carries the code object's load: ~3 ms inside k_skip_dir's events)
        leaf.segment.release_prepared_terms()
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        leaf.segment.decode_terms_device(sel, td.data_ptr(), tf.data_ptr())   # stage A of the preparation + the decode
        torch.cuda.synchronize()
        print("cold wall ms (decode_terms_device on a released store, %d terms): %.3f" % (sel.size, 1e3 * (time.perf_counter() - t0)))
        leaf.segment.prepare_terms(sel)                                       # + stage B (norms), what a search adds
    print("footprint", leaf.segment.footprint())
elif kind == "decode":
    sel = seg.terms[seg.terms["doc_freq"] >= 128]
    import torch  # device buffers only
    total = int(sel["doc_freq"].sum())
    td = torch.empty(total, dtype=torch.int32, device="cuda")
    tf = torch.empty(total, dtype=torch.int32, device="cuda")
    class _P:  # .value like a ctypes pointer
        def __init__(self, v): self.value = v
    pd, pf = _P(td.data_ptr()), _P(tf.data_ptr())
    for _ in range(reps + 2):
        leaf.segment.decode_terms_device(sel, pd.value, pf.value)
elif kind == "mustor":
    # "+a +(b c)" (RGPU_OP_SHOULD_REQUIRED): the and3 batch's term triples, the first as the MUST clause, the other two as the nested
    # disjunction. Hit counts are cross-checked by inclusion-exclusion over three conjunction batches: |a(b+c)| = |ab| + |ac| - |abc|
    tids = indexgen.log_uniform_ranks(3 * 1024, 1, 1000, SEED ^ 0xA3).reshape(-1, 3) - 1
    tids = np.array([r for r in tids if len(set(r.tolist())) == 3])
    qs, ts = s.pack([B.build([T(int(a)), B.build([], [T(int(b)), T(int(c))])], []) for a, b, c in tids], leaf)
    for _ in range(reps + 2):
        hits, totals = leaf.segment.search_batch(qs, ts, 10)
    t0 = time.perf_counter()
    for _ in range(10):
        leaf.segment.search_batch(qs, ts, 10)
    print("mustor: %d queries, wall per host-buffer batch %.3f ms" % (len(tids), 1e2 * (time.perf_counter() - t0)))
    def count(cols):
        q2, t2 = s.pack([B.build([T(int(r[c])) for c in cols], []) for r in tids], leaf)
        return leaf.segment.search_batch(q2, t2, 10)[1]
    want = count((0, 1)) + count((0, 2)) - count((0, 1, 2))
    print("mustor: hit counts equal |ab| + |ac| - |abc| on %d of %d queries; matches in all: %d" % (int((totals == want).sum()), len(tids), int(totals.sum())))
elif kind == "mustand":
    # "+a +(+b +c)" (RGPU_OP_NESTED_MUST): the and3 batch's triples as nested trees; rows against the flat conjunctions' (same docs and
    # counts; scores a + (b + c) against (x + y) + z in cost order: equal within 1e-5, and not always bit for bit)
    tids = indexgen.log_uniform_ranks(3 * 1024, 1, 1000, SEED ^ 0xA3).reshape(-1, 3) - 1
    tids = np.array([r for r in tids if len(set(r.tolist())) == 3])
    qs, ts = s.pack([B.build([T(int(a)), B.build([T(int(b)), T(int(c))], [])], []) for a, b, c in tids], leaf)
    assert all(q["op"] == (1 | (2 << 16) | (1 << 25) | (1 << 26)) for q in qs)   # AND, two nested clauses, NESTED_MUST, NESTED_AT(1)
    for _ in range(reps + 2):
        hits, totals = leaf.segment.search_batch(qs, ts, 10)
    t0 = time.perf_counter()
    for _ in range(10):
        leaf.segment.search_batch(qs, ts, 10)
    print("mustand: %d queries, wall per host-buffer batch %.3f ms" % (len(tids), 1e2 * (time.perf_counter() - t0)))
    ka = ctx.kernel_stats().get("k_search_and")
    print("mustand: k_search_and %.4f ms over %d launches (first launches included)" % (ka["total_ms"] / ka["launches"], ka["launches"]))
    ctx.kernel_stats_reset()
    q2, t2 = s.pack([B.build([T(int(x)) for x in r], []) for r in tids], leaf)
    fh, ft = leaf.segment.search_batch(q2, t2, 10)
    t0 = time.perf_counter()
    for _ in range(10):
        leaf.segment.search_batch(q2, t2, 10)
    print("flat conjunctions: wall per host-buffer batch %.3f ms" % (1e2 * (time.perf_counter() - t0)))
    same_bits = int((hits["score"].view(np.int32) == fh["score"].view(np.int32)).all(axis=1).sum())
    close = np.allclose(np.sort(hits["score"], axis=1), np.sort(fh["score"], axis=1), rtol=1e-5, atol=0)
    print("mustand: hit counts equal the flat conjunctions' on %d of %d queries; rows with the same score bits %d; all scores within 1e-5: %s"
          % (int((totals == ft).sum()), len(tids), same_bits, close))
else:
    if kind == "term":
        tids = indexgen.log_uniform_ranks(1024, 1, 10_000, SEED).reshape(-1, 1) - 1
        k = 10
    elif kind == "and3":
        tids = indexgen.log_uniform_ranks(3 * 1024, 1, 1000, SEED ^ 0xA3).reshape(-1, 3) - 1
        k = 10
    elif kind == "and2sparse":
        # "rare AND medium": a lead of a few hundred postings, a second clause below the bitmaps' density (round 6: membership bits alone)
        lead = indexgen.log_uniform_ranks(1024, docs // 2500, docs // 500, SEED ^ 0x51)
        second = indexgen.log_uniform_ranks(1024, max(2, docs // 160_000), docs // 2600, SEED ^ 0x52)
        tids = np.stack([lead, second], axis=1) - 1
        k = 10
    else:
        tids = indexgen.log_uniform_ranks(10 * 1024, 1, 10_000, SEED ^ 0x0A).reshape(-1, 10) - 1
        k = 100
    # the bench's own step: term ids -> rows in one C-ABI call, rows left in device memory (rgpu_planner_search_uniform_ids_device)
    import torch
    d_hits = torch.empty((tids.shape[0], k), dtype=torch.int64, device="cuda")
    d_tot = torch.empty((tids.shape[0],), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(reps + 2):
        s.search_uniform_device({"term": 0, "and3": 1, "and2sparse": 1}.get(kind, 2), tids, leaf, k, d_hits.data_ptr(), d_tot.data_ptr())
        ctx.synchronize()
    # ... and the same step without the profiler's events and without a sync per call: wall time per step, one stream
    plain = rucene_amd.Context()
    leaf2 = rucene_amd.LeafReader.from_synthetic(seg)
    if os.environ.get("WALL", "1") != "0":
        s2 = rucene_amd.GpuIndexSearcher([leaf2], ctx=plain)
        op = {"term": 0, "and3": 1, "and2sparse": 1}.get(kind, 2)
        for _ in range(3):
            s2.search_uniform_device(op, tids, leaf2, k, d_hits.data_ptr(), d_tot.data_ptr())
        plain.synchronize()
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20):
                s2.search_uniform_device(op, tids, leaf2, k, d_hits.data_ptr(), d_tot.data_ptr())
            plain.synchronize()
            walls.append((time.perf_counter() - t0) / 20)
        print("wall per step, one stream, no profiling: median %.4f ms (min %.4f)" % (1e3 * sorted(walls)[2], 1e3 * min(walls)))
    if kind in ("term", "and3", "and2sparse"):
        print("last launch decoded vs covered:", ctx.last_search_counters())
import ctypes as _C
_L = _C.CDLL(rucene_amd._lib.lib_path())
if hasattr(_L, "rgpu_debug_counters"):
    _o = (_C.c_ulonglong * 8)()
    _L.rgpu_debug_counters(_o, 0)
    print("dbg counters (whole run, %d launches)" % (reps + 2), list(_o))
print({n: (v["launches"], round(v["total_ms"] / max(1, v["launches"]), 4)) for n, v in ctx.kernel_stats().items()})
ctx.close()
