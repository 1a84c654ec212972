"""The reference harness's task grammar (utils/index-search.cpp:91-449), host side.

`scripts/iresearch-benchmark.tasks` holds one task per line, `Category: text # freq=...`;
`prepareTasks` (index-search.cpp:451-473) keeps the first `tasks_per_category` lines of every
category, `splitFreq` (:214-234) cuts the `# freq` annotation off, `prepareFilter` (:240-449)
turns (category, text) into a filter:

  HighTerm / MedTerm / LowTerm                  by_term
  HighPhrase / MedPhrase / LowPhrase            by_phrase of the analysed words
  AndHighHigh / AndHighMed / AndHighLow         And of by_term ("+term")
  OrHighHigh / OrHighMed / OrHighLow /
    Or4High / Or6High4Med2Low                   Or of by_term
  MinMatch2High2Med                             Or with min_match_count (first token)
  Prefix3 / Wildcard                            by_prefix / by_wildcard: multi-term expansion — the
                                                visit of the term table (`expansion_of`), then the
                                                scored form the harness builds (`scored_terms_limit`
                                                longest lists as a disjunction + one unscored bitset,
                                                index-search.cpp:363-399: search.prepare_expansions /
                                                execute_expansions) or, without scorers, ONE
                                                bit_union
  Fuzzy1 / Fuzzy2 / *NGram                      not on this path

The synthetic index has ranks, not words.  A task's words carry their document frequency in the
reference's benchmark index (`# freq=541190`, Wikipedia lines, 5 M docs in
scripts/search-benchmark.sh); a word is mapped to the Zipf RANK whose document frequency is the
same share of the index (df(r) / N = 1 - exp(-L / (H_V r))) — "High / Med / Low"
thereby land in the rank bands they name.  Words of one task get distinct ranks.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

from .search import And, Or, by_phrase, by_term

# category_t (index-search.cpp:91-116) by what prepareFilter builds from it
TERM = ("HighTerm", "MedTerm", "LowTerm")
PHRASE = ("HighPhrase", "MedPhrase", "LowPhrase")
AND = ("AndHighHigh", "AndHighMed", "AndHighLow")
OR = ("OrHighHigh", "OrHighMed", "OrHighLow", "Or4High", "Or6High4Med2Low")
MINMATCH = ("MinMatch2High2Med",)
EXPANSION = ("Prefix3", "Wildcard", "Fuzzy1", "Fuzzy2", "HighNGram", "MedNGram", "LowNGram")
UNION = ("Prefix3", "Wildcard")      # ... of which these run as ONE postings_reader::bit_union
CATEGORIES = TERM + PHRASE + AND + OR + MINMATCH + EXPANSION

REFERENCE_DOCS = 5_000_000     # scripts/search-benchmark.sh: MAX_LINES=5000000
_LINE = re.compile(r"(\S+): (.+)")                 # prepareTasks :458
_FREQ1 = re.compile(r"(\S+)\s*#\s*(.+)")           # splitFreq :215 single term, prefix
_FREQ2 = re.compile(r"\"(.+)\"\s*#\s*(.+)")        # :217 phrase
_FREQ3 = re.compile(r"((?:\S+\s+)+)\s*#\s*(.+)")   # :218 AND / OR groups


@dataclass
class Task:
    category: str
    text: str                      # what follows "Category: "
    words: list = field(default_factory=list)    # the filter's terms, in order
    freqs: list = field(default_factory=list)    # their `freq=` annotations (0: none given)
    min_match: int = 0


def split_freq(text: str):
    """splitFreq: the part in front of `# ...`, or None (the task is skipped, as the reference
    returns a null filter)."""
    for pat in (_FREQ1, _FREQ2, _FREQ3):
        m = pat.fullmatch(text)
        if m:
            return m.group(1), m.group(2)
    return None


def parse_tasks(lines, tasks_per_category: int = 1 << 30):
    """prepareTasks + the text handling of prepareFilter -> [Task]; lines that do not match are
    dropped like there.  One difference: the reference files a line of an unknown category under
    UNKNOWN, counts it and keeps it in the list, where it later yields a null filter and is skipped
    (index-search.cpp:451-473, 240-449) — here such a line is dropped at once; the tasks that
    execute are the same."""
    counts: dict = {}
    out = []
    for line in lines:
        m = _LINE.fullmatch(line.rstrip("\n"))
        if not m or m.group(1) not in CATEGORIES:
            continue
        cat, text = m.group(1), m.group(2)
        counts[cat] = counts.get(cat, 0) + 1
        if counts[cat] > tasks_per_category:
            continue
        t = Task(cat, text)
        if cat not in EXPANSION:
            sp = split_freq(text)
            if sp is None:
                continue
            body, note = sp
            toks = body.split()
            if cat in AND:
                toks = [w[1:] for w in toks]              # skip '+' at the start of the term
            if cat in MINMATCH:
                try:          # (std::stoi of the first token; a line it cannot parse is skipped)
                    t.min_match, toks = int(toks[0]), toks[1:]
                except (ValueError, IndexError):
                    continue
            t.words = toks
            # `freq=a|b|c` (phrase: phrase | word | word) or `freq=a freq=b ...`
            nums = [int(x) for x in re.findall(r"\d+", " ".join(re.findall(r"freq=[\d|]+", note)))]
            if cat in PHRASE and len(nums) == len(toks) + 1:
                nums = nums[1:]                             # (the first is the phrase's own)
            t.freqs = nums[:len(toks)] + [0] * max(0, len(toks) - len(nums))
        out.append(t)
    return out


def rank_of_freq(freq: int, reference_docs: int = REFERENCE_DOCS, mean_len: float = 100.0,
                 vocab_log2: int = 20) -> float:
    """The Zipf rank whose document frequency is the same SHARE of the synthetic index as `freq`
    is of the reference's: df(r) / N = 1 - exp(-L / (H_V r))."""
    h = math.log(2.0) * vocab_log2 + 0.5772156649
    share = min(max(freq / float(reference_docs), 1e-9), 0.999999)
    return mean_len / (h * -math.log1p(-share))


def ranks_of(task: Task, max_rank: int, jitter: float = 0.0, rng=None, lo_rank: int = 1):
    """Distinct term ranks (1-based) for the task's words; jitter: a seeded relative
    perturbation (several distinct queries of one class)."""
    used, out = set(), []
    for w, f in zip(task.words, task.freqs):
        r = rank_of_freq(f) if f else float(max_rank)
        if jitter and rng is not None:
            r *= 1.0 + jitter * (2.0 * rng.random() - 1.0)
        r = int(min(max(round(r), lo_rank), max_rank))
        while r in used:                    # words of one filter are different terms
            r = r + 1 if r < max_rank else lo_rank
        used.add(r)
        out.append(r)
    return out


def filter_of(task: Task, ranks):
    """prepareFilter's filter for a task whose words sit at `ranks` (term ordinal = rank - 1);
    None for the expansion categories."""
    terms = [int(r) - 1 for r in ranks]
    if task.category in TERM:
        return by_term(terms[0])
    if task.category in PHRASE:
        return by_phrase(terms)
    if task.category in AND:
        return And([by_term(t) for t in terms])
    if task.category in OR:
        return Or([by_term(t) for t in terms])
    if task.category in MINMATCH:
        return Or([by_term(t) for t in terms], min_match=task.min_match)
    return None


def expansion_of(task: Task, n_terms: int, rng=None):
    """The term ordinals a Prefix3 / Wildcard task VISITS in the synthetic field's sorted term table
    — what by_prefix's / by_wildcard's term visitor enumerates from the dictionary
    (prefix_filter.cpp:37-60; wildcard_filter.cpp -> the same visit under an automaton) — as a sorted
    uint32 array.  A synthetic term is its ordinal's 4-byte big-endian number (synth.term_bytes_of),
    so the task's pattern keeps its SHAPE and takes synthetic bytes: `sec*` (three fixed bytes, then
    anything) = the 256 ordinals sharing one 3-byte prefix; `re*f` (two fixed bytes, anything, one
    fixed last byte) = every 256th ordinal of one 65536-term range.  The fixed bytes are drawn per
    query (`rng`) from the ranges the vocabulary fills."""
    import numpy as np
    text = task.text.strip()
    star = text.find("*")
    if task.category not in UNION or star < 0:
        return None
    n_pre, n_suf = star, len(text) - star - 1
    if n_pre + n_suf > 4:
        return np.zeros(0, np.uint32)
    top = max(int(n_terms) - 1, 0).to_bytes(4, "big")
    pre = bytearray()
    for i in range(n_pre):      # leading bytes stay inside the vocabulary [0, n_terms)
        hi = top[i] if bytes(pre) == top[:i] else 255
        pre.append(int(rng.integers(0, hi + 1)) if rng is not None else hi // 2)
    suf = bytes(int(rng.integers(0, 256)) if rng is not None else 0x66 for _ in range(n_suf))
    # the visit itself: the sorted term table against prefix and suffix, byte by byte
    table = np.arange(int(n_terms), dtype=">u4").view(np.uint8).reshape(-1, 4)
    ok = np.ones(len(table), bool)
    for i, b in enumerate(pre):
        ok &= table[:, i] == b
    for i, b in enumerate(suf):
        ok &= table[:, 4 - n_suf + i] == b
    return np.nonzero(ok)[0].astype(np.uint32)
