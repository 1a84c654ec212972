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


def _one_of_each(L, docs, max_rank):
    seg = synth.build_segment(docs, max_rank, with_positions=True)
    sr = search.SegmentReader.from_synth(seg, L=L)
    st = [parity.segment_stats(seg)]
    lines = json.load(open(GOLDEN))["lines"]
    parsed = [t for t in tasks.parse_tasks(lines, 1) if t.category not in tasks.EXPANSION]
    assert len(parsed) == 15
    for scorer in (BM25(), TFIDF(False)):
        boolean, phrases = [], []
        for t in parsed:
            flt = tasks.filter_of(t, tasks.ranks_of(t, max_rank))
            (phrases if t.category in tasks.PHRASE else boolean).append(flt)
        for filters, check in ((boolean, parity.check_single_segment), (phrases, parity.check_phrase_segment)):
            b = sr.batch(search.prepare(filters, scorer, st), 100)
            hits, counts, totals = (x.copy() for x in b.run().results_to_host().host_results())
            check(seg, filters, scorer, 100, hits, counts, totals)
            assert (totals > 0).sum() >= len(filters) - 2     # (the classes really match docs)
            b.close()
    sr.close()


def test_one_query_of_every_task_class(simlib):
    _one_of_each(simlib, 60_000, 2048)


@pytest.mark.gpu
def test_one_query_of_every_task_class_gpu(gpulib):
    _one_of_each(gpulib, 2_000_000, 65536)
