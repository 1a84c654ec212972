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


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), TFIDF(False), TFIDF(True)])
def test_prepare_filters_equals_term_by_term(scorer):
    """The vectorised prepare of ANY flat filter list (what bench.py --tasks runs per step) against
    the term-by-term mirror: boolean kinds mixed in one list, phrases in one of their own."""
    from iresearch_amd.search import And, by_phrase
    rng = np.random.default_rng(11)
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(0, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(0, 30_000, 300))]
    segs = [_Seg(400), _Seg(300)]
    booleans = []
    for i in range(96):
        n = int(rng.integers(1, 13))
        subs = [by_term(int(t), float(rng.choice([1.0, 0.5, 2.25]))) for t in rng.integers(0, 400, n)]
        kind = i % 4
        if kind == 0:
            booleans.append(subs[0])
        elif kind == 1:
            booleans.append(Or(subs, merge=int(rng.integers(0, 3))))
        elif kind == 2:
            booleans.append(Or(subs, min_match=int(rng.integers(2, 5))))
        else:
            booleans.append(And(subs))
    phrases = [by_phrase([int(t) for t in rng.integers(0, 400, int(rng.integers(1, 6)))],
                         boost=float(rng.choice([1.0, 3.0]))) for _ in range(40)]
    phrases.append(by_phrase([5, 9, 2], offsets=[0, 2, 5]))
    for filters in (booleans, phrases):
        slow = search.QueryArrays.from_prepared(segs, search.prepare(filters, scorer, stats), 9)
        fast = search.prepare_filters(filters, scorer, stats, segs, 9)
        assert slow.queries.tobytes() == fast.queries.tobytes()
        for name in search.TERM_SCORER.names:
            assert np.array_equal(slow.terms[name], fast.terms[name]), name


@pytest.mark.parametrize("scorer", [BM25(), BM25(1.2, 0.0), TFIDF(False)])
def test_prepare_expansions_statistics(scorer):
    """The scored terms of a multi-term filter carry the statistics limited_sample_collector::score
    gives them — the field's over the whole index, the term's from the segments where the term is
    SCORED — equal to what scorer.collect / term_scorer yield term by term."""
    rng = np.random.default_rng(2)
    stats = [search.SegmentStats(100_000, 9_700_000, rng.integers(1, 50_000, 400)),
             search.SegmentStats(60_000, 5_100_000, rng.integers(1, 30_000, 400))]
    visits = [[np.arange(10, 60, dtype=np.uint32), np.arange(10, 45, dtype=np.uint32)],
              [np.arange(100, 104, dtype=np.uint32), np.zeros(0, np.uint32)]]
    for limit in (16, 3, 0):
        prep = search.prepare_expansions(visits, limit, scorer, stats, boost=1.5)
        for p, per_seg in zip(prep, visits):
            assert sum(len(x) for x in p.scored_in) == min(limit, sum(len(v) for v in per_seg))
            for t, sc in zip(p.scored, p.scorers):
                dwt = sum(int(stats[s].docs_count[t]) for s in range(2) if t in p.scored_in[s])
                want = scorer.term_scorer(scorer.collect(160_000, dwt, 14_800_000), 1.5)
                assert tuple(sc) == tuple(want), (limit, t)
            for s in range(2):      # every visited term is scored or unscored, never both
                assert sorted(list(p.scored_in[s]) + p.unscored_in[s].tolist()) == per_seg[s].tolist()
