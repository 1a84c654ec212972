#!/usr/bin/env python3
"""tests/golden/benchmark_tasks.json: the task list the reference's benchmark scripts feed to
utils/index-search (scripts/iresearch-benchmark.tasks: 18 lines of `Category: text # freq=...`)
as a data fixture, with what the reference's grammar makes of every line (category, filter words,
min-match count) worked out BY HAND from index-search.cpp:214-449 for the parser test.  Build
container only (reads /root/reference); bench.py --tasks and the tests read the JSON.

  python tests/golden/make_tasks_golden.py
"""
import json
from pathlib import Path

SRC = Path("/root/reference/scripts/iresearch-benchmark.tasks")
OUT = Path(__file__).resolve().parent / "benchmark_tasks.json"

# (category, words the filter is built from, min_match) per line, in file order — read off the
# file against prepareFilter by hand; the parser test compares tasks.parse_tasks with this
EXPECTED = [
    ("HighTerm", ["ref"], 0), ("MedTerm", ["second"], 0), ("LowTerm", ["demographics"], 0),
    ("HighPhrase", ["ref", "name"], 0), ("MedPhrase", ["books", "id"], 0),
    ("LowPhrase", ["year", "ref"], 0),
    ("AndHighHigh", ["state", "south"], 0), ("AndHighMed", ["12", "federal"], 0),
    ("AndHighLow", ["from", "house's"], 0),
    ("OrHighHigh", ["about", "september"], 0), ("OrHighMed", ["south", "1929"], 0),
    ("OrHighLow", ["york", "projectile"], 0),
    ("Prefix3", [], 0), ("Wildcard", [], 0),
    ("Or4High", ["about", "ref", "from", "cite"], 0),
    ("Or6High4Med2Low", ["about", "ref", "from", "cite", "http", "which", "roman", "short",
                         "europe", "party", "rapid", "donald"], 0),
    ("MinMatch2High2Med", ["about", "ref", "roman", "short"], 2),
]


def main():
    lines = [l.rstrip("\n") for l in SRC.read_text().splitlines() if l.strip()]
    assert len(lines) == len(EXPECTED), (len(lines), len(EXPECTED))
    json.dump({"source": "scripts/iresearch-benchmark.tasks (iresearch v1.3)", "lines": lines,
               "expected": [{"category": c, "words": w, "min_match": m} for c, w, m in EXPECTED]},
              open(OUT, "w"), indent=1)
    print("wrote", OUT, len(lines), "lines")


if __name__ == "__main__":
    main()
