"""oracle/parity.py — the rule that judges >= 10-clause disjunctions (doc ids AND scores) — on the CPU: it must accept
the oracle's own rows and reject a wrong doc that carries a plausible score, before it is trusted with the HIP path."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def small(oracle):
    import __graft_entry__ as g
    g.build()
    from rucene_amd import indexgen
    seg = indexgen.build_zipf(60_000, 6_000)
    oseg = oracle.Segment(seg.doc_bytes, seg.norms, seg.max_doc, seg.terms, sum_total_term_freq=seg.sum_total_term_freq)
    return seg, oracle.Searcher([oseg])


def test_score_docs_agrees_with_the_collectors_scores(small, oracle):
    seg, osr = small
    for op, tids in [(oracle.OP_TERM, [7]), (oracle.OP_AND, [0, 3, 11]), (oracle.OP_OR, [1, 50, 400]),
                     (oracle.OP_OR, [2, 9, 30, 77, 150, 600, 1000, 2500, 4000, 5999, 5, 12])]:
        d, s, total = osr.search(op, tids, 50, tie_mode=oracle.TIE_CANONICAL)
        assert d.size > 0
        sc, matched = osr.score_docs(op, tids, d)
        assert matched.all()
        if op == oracle.OP_OR and len(tids) >= 10:  # heap order on both sides, but not the same heap history
            np.testing.assert_allclose(sc, s, rtol=2e-6, atol=0)
        else:
            assert (sc.view(np.int32) == s.view(np.int32)).all()
    # docs that hold none of the terms are no matches
    d, _, _ = osr.search(oracle.OP_OR, [5000, 5001], 10)
    others = np.setdiff1d(np.arange(0, 200, dtype=np.int32), d)[:20]
    sc, matched = osr.score_docs(oracle.OP_OR, [5000, 5001], others)
    assert not matched.any() and (sc == 0).all()


def test_rule_accepts_the_oracle_and_rejects_a_plausible_wrong_doc(small, oracle):
    from oracle import parity
    seg, osr = small
    tids = [2, 9, 30, 77, 150, 600, 1000, 2500, 4000, 5999]
    k = 100
    d, s, total = osr.search(oracle.OP_OR, tids, k, tie_mode=oracle.TIE_CANONICAL)
    row_d = np.full(k, -1, np.int32)
    row_s = np.zeros(k, np.float32)
    row_d[:d.size], row_s[:d.size] = d, s
    assert parity.check_heap_order_row(osr, oracle.OP_OR, tids, row_d, row_s, total, d, s, d.size, total) == 0
    # a non-matching doc with the neighbour's score
    _, matched = osr.score_docs(oracle.OP_OR, tids, np.arange(seg.max_doc - 3000, seg.max_doc, dtype=np.int32))
    stranger = int(np.arange(seg.max_doc - 3000, seg.max_doc)[~matched][0])
    bad_d = row_d.copy()
    bad_d[5] = stranger
    with pytest.raises(parity.HeapOrderParityError, match="do not match"):
        parity.check_heap_order_row(osr, oracle.OP_OR, tids, bad_d, row_s, total, d, s, d.size, total)
    # a matching doc from far below the k-th score, dressed up with a top score
    d2, s2, _ = osr.search(oracle.OP_OR, tids, 4000, tie_mode=oracle.TIE_CANONICAL)
    low = int(d2[np.nonzero(s2 < 0.5 * s[d.size - 1])[0][0]])
    bad_d = row_d.copy()
    bad_d[0] = low
    with pytest.raises(parity.HeapOrderParityError):
        parity.check_heap_order_row(osr, oracle.OP_OR, tids, bad_d, row_s, total, d, s, d.size, total)
    # a missing top hit (rows shifted up, a tie-band doc appended) and a wrong count
    shifted_d = np.concatenate([row_d[1:d.size], d2[d.size:d.size + 1], row_d[d.size:]])
    shifted_s = np.concatenate([row_s[1:d.size], s2[d.size:d.size + 1], row_s[d.size:]])
    with pytest.raises(parity.HeapOrderParityError, match="missing"):
        parity.check_heap_order_row(osr, oracle.OP_OR, tids, shifted_d, shifted_s, total, d, s, d.size, total)
    with pytest.raises(parity.HeapOrderParityError, match="total_hits"):
        parity.check_heap_order_row(osr, oracle.OP_OR, tids, row_d, row_s, total + 1, d, s, d.size, total)
    # scores perturbed inside the tolerance pass; outside it they do not
    ok_s = row_s.copy()
    ok_s[:d.size] = (ok_s[:d.size].astype(np.float64) * (1 + 3e-6)).astype(np.float32)
    parity.check_heap_order_row(osr, oracle.OP_OR, tids, row_d, ok_s, total, d, s, d.size, total)
    off_s = row_s.copy()
    off_s[3] = np.float32(off_s[3] * (1 - 5e-5))
    with pytest.raises(parity.HeapOrderParityError):
        parity.check_heap_order_row(osr, oracle.OP_OR, tids, row_d, off_s, total, d, s, d.size, total)
