#!/usr/bin/env python3
"""tools/pmc_table.py DIR — per-kernel averages of every counter found in the rocprofv3
counter_collection.csv files under DIR (one sub-directory per --pmc pass)."""
import collections
import csv
import glob
import sys


def main():
    root = sys.argv[1]
    table = collections.defaultdict(dict)
    for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("irs_hip::", "")
            agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        per = collections.defaultdict(list)
        for (k, _), v in agg.items():
            per[k].append(v)
        for k, vs in per.items():
            table[k]["launches"] = len(vs)
            for c in vs[0]:
                table[k][c] = sum(v.get(c, 0.0) for v in vs) / len(vs)
    for k in sorted(table, key=lambda k: -table[k].get("SQ_BUSY_CYCLES", table[k].get("SQ_WAVES", 0))):
        if not (k.startswith("k_join") or k.startswith("k_score") or k.startswith("k_select")
                or k.startswith("k_items") or k.startswith("k_pilot") or k.startswith("k_plan")
                or k.startswith("k_conj") or k.startswith("k_phrase") or k.startswith("k_fast")):
            continue
        print("== %s" % k)
        for c in sorted(table[k]):
            print("   %-34s %18.0f" % (c, table[k][c]))


if __name__ == "__main__":
    main()
