"""The reference harness's task grammar (utils/index-search.cpp:91-449) on this path: the parser
against the reference's own task list (tests/golden/benchmark_tasks.json, every line's category /
filter words / min-match worked out by hand from prepareFilter), and one query of every class
through the C ABI against the oracle (emulator tier and GPU tier)."""
import json
from pathlib import Path

import numpy as np
import pytest

import parity
from iresearch_amd import search, synth, tasks
from iresearch_amd.search import BM25, TFIDF

GOLDEN = Path(__file__).parent / "golden" / "benchmark_tasks.json"


def test_task_grammar_against_the_reference_list():
    g = json.load(open(GOLDEN))
    parsed = tasks.parse_tasks(g["lines"])
    assert len(parsed) == len(g["expected"]) == 17
    for t, e in zip(parsed, g["expected"]):
        assert (t.category, t.words, t.min_match) == (e["category"], e["words"], e["min_match"])
        assert len(t.freqs) == len(t.words) and all(f > 0 for f in t.freqs)
    # prepareTasks keeps the first N lines of every category (:466-470), drops unknown ones
    twice = g["lines"] + g["lines"] + ["Nonsense: x # freq=1", "not a task line"]
    assert len(tasks.parse_tasks(twice, 1)) == 17 and len(tasks.parse_tasks(twice, 2)) == 34
    # splitFreq (:214-234): no `# freq` part -> the task is skipped (null filter)
    assert tasks.parse_tasks(["HighTerm: ref"]) == []
    # High / Med / Low land in their rank bands
    r = {t.category: tasks.ranks_of(t, 1 << 20) for t in parsed if t.words}
    assert r["HighTerm"][0] < 100 < r["MedTerm"][0] < 1000 < r["LowTerm"][0]
    assert len(set(r["MinMatch2High2Med"])) == 4      # equal frequencies -> still distinct terms


def _one_of_each(L, docs, max_rank, per_class=1):
    """`per_class` queries of every class (the task's ranks jittered +-25 % like bench.py --tasks)
    through the C ABI against the oracle; the two expansion classes as bit_union."""
    import oracle
    seg = synth.build_segment(docs, max_rank, with_positions=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    st = [parity.segment_stats(seg)]
    lines = json.load(open(GOLDEN))["lines"]
    parsed = [t for t in tasks.parse_tasks(lines, 1) if t.category not in tasks.EXPANSION]
    assert len(parsed) == 15
    rng = np.random.default_rng(5)
    for scorer in (BM25(), TFIDF(False)):
        boolean, phrases = [], []
        for t in parsed:
            for i in range(per_class):
                ranks = tasks.ranks_of(t, max_rank) if i == 0 else tasks.ranks_of(t, max_rank, 0.25, rng)
                (phrases if t.category in tasks.PHRASE else boolean).append(tasks.filter_of(t, ranks))
        for filters, check in ((boolean, parity.check_single_segment), (phrases, parity.check_phrase_segment)):
            b = sr.batch(search.prepare_filters(filters, scorer, st, [sr], 100), 100)
            hits, counts, totals = (x.copy() for x in b.run().results_to_host().host_results())
            check(seg, filters, scorer, 100, hits, counts, totals)
            assert (totals > 0).sum() >= len(filters) - 2 * per_class   # (the classes really match docs)
            b.close()
    # Prefix3 / Wildcard without scorers: the visit, then ONE bit_union
    n_words = (seg.num_docs + 64) // 64
    for t in tasks.parse_tasks(lines, 1):
        if t.category not in tasks.UNION:
            continue
        for i in range(per_class):
            v = tasks.expansion_of(t, max_rank, rng)
            assert len(v) and (np.diff(v.astype(np.int64)) > 0).all() and v[-1] < max_rank
            gb, gn = sr.bit_union(v, n_words)
            ob, on = oracle.bit_union(seg.doc_file, [seg.metas[int(x)] for x in v], seg.layout, True, n_words)
            assert gn == on and np.array_equal(gb, ob)
            # ... and WITH scorers, as the harness builds them (scored_terms_limit = 16)
            visits = [[v[np.asarray(seg.metas["docs_count"])[v] > 0]]]
            for scorer in (BM25(), TFIDF(False)):
                prep = search.prepare_expansions(visits, 16, scorer, st)
                h, c, tot = search.execute_expansions([sr], prep, 100)
                parity.check_expansions([seg], visits, 16, scorer, 100, h, c, tot)
    sr.close()


def test_expansion_visits_the_sorted_term_table():
    """expansion_of against a brute-force match of the pattern on the terms' bytes."""
    g = {t.category: t for t in tasks.parse_tasks(json.load(open(GOLDEN))["lines"], 1)}
    rng = np.random.default_rng(3)
    for n_terms in (5000, 70000, 262144):
        for cat, n_pre, n_suf in (("Prefix3", 3, 0), ("Wildcard", 2, 1)):
            v = tasks.expansion_of(g[cat], n_terms, rng)
            terms = [synth.term_bytes_of(i) for i in range(n_terms)]
            pre = terms[int(v[0])][:n_pre]
            suf = terms[int(v[0])][4 - n_suf:] if n_suf else b""
            want = [i for i, tb in enumerate(terms) if tb.startswith(pre) and tb.endswith(suf)]
            assert v.tolist() == want
    assert tasks.expansion_of(g["HighTerm"], 1000, rng) is None


def test_one_query_of_every_task_class(simlib):
    _one_of_each(simlib, 60_000, 2048, per_class=2)


@pytest.mark.gpu
def test_one_query_of_every_task_class_gpu(gpulib):
    _one_of_each(gpulib, 2_000_000, 65536, per_class=8)
