"""Host logic above the C ABI that needs no device: the vectorised per-batch prepare of a
serving loop (search.prepare_disjunctions) against the term-by-term mirror of the reference's
by_term::prepare + BM25::collect / TFIDF::collect (search.prepare)."""
import numpy as np
import pytest

from iresearch_amd import search
from iresearch_amd.search import BM25, TFIDF, Or, by_term


class _Seg:
    def __init__(self, n_terms):
        self.metas = np.zeros(n_terms)


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), BM25(0.0, 0.75), TFIDF(False), TFIDF(True)])
def test_prepare_disjunctions_equals_term_by_term(scorer):
    rng = np.random.default_rng(7)
    # two segments with different vocabularies: ordinals >= 300 are absent from the second
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(0, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(0, 30_000, 300))]
    segs = [_Seg(400), _Seg(300)]
    rows = rng.integers(0, 400, (64, 5))
    slow = search.QueryArrays.from_prepared(
        segs, search.prepare([Or([by_term(int(t), 1.5) for t in row]) for row in rows], scorer, stats), 17)
    fast = search.prepare_disjunctions(rows, scorer, stats, segs, 17, boost=1.5)
    assert slow.k == fast.k == 17
    assert slow.queries.tobytes() == fast.queries.tobytes()
    for name in search.TERM_SCORER.names:
        assert np.array_equal(slow.terms[name], fast.terms[name]), name
    assert (fast.terms["term"][1] == search.NO_TERM).any()
