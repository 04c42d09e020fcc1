"""ORACLE — TEST INFRASTRUCTURE ONLY. Not shipped, not on the product path.

The parity rule for disjunctions the reference itself sums in heap order (DisjunctionSumScorer over a DisiPriorityQueue:
>= 10 SHOULD clauses, min_should_match <= 1 — search/scorer/disjunction_scorer.rs:41-45, 213-225; util/disi.rs:188-230).
There the reference pins a doc's score only up to the rounding of an f32 sum of non-negative terms, so "the same top-k"
cannot mean "the same bits". It means (north_star: bit-exact doc-id sets, scores within 1e-5 relative):

  1. TopDocs::total_hits is equal;
  2. the same number of rows is filled;
  3. every returned doc MATCHES the query and the oracle's own score OF THAT DOC (IndexSearcher::score_docs: the query's
     scorer advanced to it) is within `rtol` of the returned score — a wrong doc with a plausible score fails here;
  4. every oracle hit whose score exceeds the oracle's k-th score by more than `rtol` relative is returned — docs may
     only be exchanged inside the band of scores that tie with the k-th within the tolerance;
  5. rows are ordered (score desc; equal scores by doc id asc).

`docs_differing` = how many oracle hits are missing from the returned row (0 = identical doc-id sets).
"""
import numpy as np


class HeapOrderParityError(AssertionError):
    pass


def check_heap_order_row(osearcher, op, term_ids, got_docs, got_scores, got_total, want_docs, want_scores, want_n, want_total,
                         rtol=1e-5, min_should_match=0, what=""):
    """One query's row against the oracle's (canonical mode). `got_*`: the k-row under test, unused slots doc -1;
    `want_*`: the oracle's row, `want_n` slots filled. Returns docs_differing; raises HeapOrderParityError."""
    def fail(msg):
        raise HeapOrderParityError("%s: %s" % (what, msg))

    n = int(want_n)
    got_docs = np.asarray(got_docs)
    got_scores = np.asarray(got_scores, dtype=np.float32)
    if int(got_total) != int(want_total):
        fail("total_hits %d != %d" % (int(got_total), int(want_total)))
    if not (got_docs[n:] == -1).all() or (got_docs[:n] < 0).any():
        fail("filled rows: want %d, got %s" % (n, got_docs.tolist()))
    if n == 0:
        return 0
    gd, gs = got_docs[:n], got_scores[:n]
    wd, ws = np.asarray(want_docs)[:n], np.asarray(want_scores, dtype=np.float32)[:n]
    if np.unique(gd).size != n:
        fail("a doc is returned twice")
    # 5. order
    ds = np.diff(gs.astype(np.float64))
    if (ds > 0).any():
        fail("scores not in descending order")
    if (np.diff(gd)[ds == 0] <= 0).any():
        fail("equal scores not in ascending doc order")
    # 3. the oracle's score of the returned docs
    os_, matched = osearcher.score_docs(op, term_ids, gd, min_should_match=min_should_match)
    if not matched.all():
        fail("returned docs that do not match the query: %s" % gd[~matched].tolist())
    bad = np.abs(os_.astype(np.float64) - gs.astype(np.float64)) > rtol * np.abs(os_.astype(np.float64))
    if bad.any():
        i = int(np.nonzero(bad)[0][0])
        fail("doc %d: returned score %r, oracle scores it %r" % (int(gd[i]), float(gs[i]), float(os_[i])))
    # 4. everything clearly above the oracle's k-th score must be there
    kth = float(ws[n - 1])
    clear = ws.astype(np.float64) > kth * (1.0 + rtol)
    missing = np.setdiff1d(wd[clear], gd)
    if missing.size:
        fail("oracle hits above the k-th score band are missing: %s" % missing.tolist())
    return int(np.setdiff1d(wd, gd).size)


def check_heap_order_batch(osearcher, op, term_rows, got_hits, got_totals, want_docs, want_scores, want_counts, want_totals,
                           rtol=1e-5, what=""):
    """Rows of a uniform batch (`term_rows`: [n_queries, n_clauses] term ids; `got_hits`: structured rows with "doc" /
    "score"). Returns the total docs_differing over the batch."""
    differing = 0
    for i in range(len(term_rows)):
        differing += check_heap_order_row(osearcher, op, term_rows[i], got_hits[i]["doc"], got_hits[i]["score"], got_totals[i],
                                          want_docs[i], want_scores[i], want_counts[i], want_totals[i], rtol=rtol,
                                          what="%s query %d %s" % (what, i, list(map(int, term_rows[i]))))
    return differing
