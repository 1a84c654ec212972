#!/usr/bin/env python3
"""Generates tests/golden/boolean_golden.json from the reference's OWN iterator tests
(run in the build container only; /root/reference does not travel):

tests/search/boolean_filter_tests.cpp holds, for its disjunction / conjunction / min-match
iterators, literal posting lists (`docs{{...}, {...}}`) and the literal doc ids the iterator
must yield (`expected{...}`):
  basic_disjunction / small_disjunction_test / block_disjunction_test / disjunction_test  :: next
  conjunction_test :: next
  block_disjunction_test :: min_match_next,  min_match_disjunction_test :: next

Only DATA is extracted (lists, min-match count, expected ids).  Every vector is cross-checked
against the set definition (union / intersection / "in at least m lists") so that a parsing
slip cannot become a wrong expectation; min-match vectors are kept for 1 <= m <= #lists, where
the iterator classes and MinMatchQuery::execute (boolean_query.cpp:212-247) agree.
"""
import json
import re
import sys
from pathlib import Path

REF = Path("/root/reference/tests/search/boolean_filter_tests.cpp")
OUT = Path(__file__).parent / "boolean_golden.json"
TESTS = {
    ("basic_disjunction", "next"): "or", ("small_disjunction_test", "next"): "or",
    ("block_disjunction_test", "next"): "or", ("disjunction_test", "next"): "or",
    ("conjunction_test", "next"): "and",
    ("block_disjunction_test", "min_match_next"): "minmatch",
    ("min_match_disjunction_test", "next"): "minmatch",
}
NUMS = re.compile(r"\d+")


def lists_of(text):
    inner = text.strip()
    out = [[int(x) for x in NUMS.findall(m)] for m in re.findall(r"\{([^{}]*)\}", inner)]
    return out


def main():
    src = REF.read_text()
    heads = [(m.start(), m.group(1), m.group(2))
             for m in re.finditer(r"^TEST\((\w+),\s*(\w+)\)", src, re.M)]
    heads.append((len(src), "", ""))
    vectors, seen = [], set()
    for (at, suite, name), (end, _, _) in zip(heads, heads[1:]):
        op = TESTS.get((suite, name))
        if not op:
            continue
        body = src[at:end]
        docs_defs = [(m.start(), lists_of(m.group(1))) for m in re.finditer(
            r"std::vector<std::vector<irs::doc_id_t>>\s+docs\s*\{(.*?)\};", body, re.S)]
        for r in re.finditer(r"std::vector<irs::doc_id_t>\s+result;", body):
            prev = [d for d in docs_defs if d[0] < r.start()]
            if not prev:
                continue
            dpos, lists = prev[-1]
            nxt = [d[0] for d in docs_defs if d[0] > r.start()]
            stop = min(nxt) if nxt else len(body)
            # the block this `result` belongs to ends at its ASSERT_EQ(..., result)
            tail = body[r.start():stop]
            a = re.search(r"ASSERT_EQ\(([\w.()]+),\s*result\)", tail)
            if not a:
                continue
            seg = body[dpos:r.start() + a.end()]
            if a.group(1) == "expected":
                e = re.findall(r"std::vector<irs::doc_id_t>\s+expected\s*\{(.*?)\};", seg, re.S)
                if not e:
                    continue
                expected = [int(x) for x in NUMS.findall(e[-1])]
            elif a.group(1) == "docs.front()":
                expected = list(lists[0])
            else:
                continue
            m = 1
            if op == "minmatch":
                blk = body[max(dpos, body.rfind("{\n", 0, r.start() - 200)):r.start() + a.end()]
                mm = re.findall(r"min_match_count\s*=\s*(\d+)\s*;", blk)
                mm2 = re.findall(r"\(docs\),\s*(\d+)U?\)", tail[:a.start()])
                if mm:
                    m = int(mm[-1])
                elif mm2:
                    m = int(mm2[-1])
                else:
                    m = 1
                if not (1 <= m <= len(lists)):
                    continue
            if not lists or any(not l for l in lists):
                continue
            # set definition
            cnt = {}
            for l in lists:
                for d in set(l):
                    cnt[d] = cnt.get(d, 0) + 1
            need = {"or": 1, "and": len(lists), "minmatch": m}[op]
            want = sorted(d for d, c in cnt.items() if c >= need)
            if want != expected:
                print("skip (parse / set definition disagree):", suite, name, op, m, lists,
                      expected, file=sys.stderr)
                continue
            key = (op, m, json.dumps(lists))
            if key in seen:
                continue
            seen.add(key)
            vectors.append({"test": "%s.%s" % (suite, name), "op": op, "min_match": m,
                            "lists": lists, "expected": expected})
    OUT.write_text(json.dumps({"vectors": vectors}) + "\n")
    by = {}
    for v in vectors:
        by[v["op"]] = by.get(v["op"], 0) + 1
    print(len(vectors), "vectors", by, "->", OUT)


if __name__ == "__main__":
    main()
