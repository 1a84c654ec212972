#!/usr/bin/env python3
"""Generates tests/golden/phrase_golden.json from the reference's OWN test expectations
(run in the build container only; /root/reference does not travel):

  corpus : tests/resources/phrase_sequential.json (a data file of the reference's tests)
  vectors: every `irs::by_phrase` block of tests/search/phrase_filter_tests.cpp that is
           built from plain terms only (push_back<irs::by_term_options>(offs)), with the
           doc names the test asserts, in order, and the frequencies it asserts.

Nothing but DATA is extracted (phrase words, offsets, expected doc names / frequencies);
each vector is cross-checked against a brute-force evaluation of the corpus so that a
parsing slip cannot produce a wrong expectation silently.
"""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference/tests")
OUT = Path(__file__).parent / "phrase_golden.json"


def main():
    corpus = json.loads((REF / "resources/phrase_sequential.json").read_text())
    docs = [(d["name"], d["phrase"].split()) for d in corpus]
    src = (REF / "search/phrase_filter_tests.cpp").read_text()
    chunks = src.split("irs::by_phrase q;")[1:]
    part_re = re.compile(
        r"push_back<irs::by_term_options>\((\d*)\)\s*\.term\s*=\s*irs::ViewCast<irs::byte_type>\("
        r"\s*std::string_view\(\"([^\"]*)\"\)\)", re.S)
    vectors, seen = [], set()
    for ch in chunks:
        end = ch.find("ASSERT_FALSE(docs->next())")
        if end < 0:
            continue
        body = ch[:end]
        if "phrase_anl" not in body:
            continue
        # plain terms only: no other phrase part types, no positional insert
        if re.search(r"by_(prefix|wildcard|edit_distance|terms|range)_options|insert<|insert\(|"
                     r"lt\.|pt\.|wt\.|rt\.|st\.", body):
            continue
        parts = part_re.findall(body)
        if not parts or body.count("push_back<") != len(parts):
            continue
        pos, words, offsets = -1, [], []
        for offs, w in parts:
            pos = pos + 1 + (int(offs) if offs else 0)
            words.append(w)
            offsets.append(pos)
        base = offsets[0]
        offsets = [o - base for o in offsets]
        names = re.findall(r"ASSERT_EQ\(\s*\"([A-Z0-9]+)\"", body)
        order = [n for i, n in enumerate(names) if i == 0 or names[i - 1] != n]
        freqs = [int(x) for x in re.findall(r"ASSERT_EQ\((\d+),\s*freq->value\)", body)]
        # brute force over whitespace tokens (text analyzer, locale C, no stopwords)
        want, wf = [], []
        for name, toks in docs:
            c = sum(all(p + o < len(toks) and toks[p + o] == w for w, o in zip(words, offsets))
                    for p in range(len(toks)))
            if c:
                want.append(name)
                wf.append(c)
        if order != want:
            print("skip (parse/brute-force disagree):", words, offsets, order, want,
                  file=sys.stderr)
            continue
        if freqs and len(freqs) % len(want) == 0:
            per = len(freqs) // len(want)
            got = freqs[::per][:len(want)] if per else []
            if got != wf:
                print("freq mismatch:", words, got, wf, file=sys.stderr)
                continue
        key = (tuple(words), tuple(offsets))
        if key in seen:
            continue
        seen.add(key)
        vectors.append({"words": words, "offsets": offsets, "docs": order,
                        "freqs": wf if freqs else None})
    # ranked vectors: test_phrase of bm25_test.cpp (BM25 with b = 0) and tfidf_test.cpp
    # (TFIDF without norms) assert the ORDER of the docs of "jumps high" by score (a
    # std::multimap<score, name, greater>: equal scores keep doc order)
    ranked = []
    for fname, scorer in (("search/bm25_test.cpp", "bm25_b0"), ("search/tfidf_test.cpp", "tfidf")):
        text = (REF / fname).read_text()
        at = text.find('// "jumps high" with order')
        assert at > 0, fname
        body = text[at:at + 1500]
        words = re.findall(r'std::string_view\("([a-z]+)"\)', body[:600])
        assert words == ["jumps", "high"], (fname, words)
        exp = re.search(r"expected\{(.*?)\};", body, re.S).group(1)
        order = re.findall(r'"([A-Z0-9]+)"', re.sub(r"//[^\n]*", "", exp))
        assert order == ["O", "P", "Q", "R"], (fname, order)
        ranked.append({"words": words, "offsets": [0, 1], "scorer": scorer, "ranked": order})
    OUT.write_text(json.dumps({"corpus": corpus, "vectors": vectors, "ranked": ranked},
                              indent=1) + "\n")
    print(len(vectors), "doc-set vectors,", len(ranked), "ranked vectors ->", OUT)


if __name__ == "__main__":
    main()
