#!/usr/bin/env python3
"""Turns rocprofv3 outputs under gpurun_out/ into the committed summaries in profiles/:
  profiles/rNN_kernel_stats.csv   (copy of --kernel-trace --stats)
  profiles/rNN_pmc.json           per-kernel counter sums per launch
  profiles/traffic_latest.json    k_score HBM traffic per launch (bench.py reads it)
usage: python tools/summarize_prof.py rNN <stats_dir> <pmc_dir> [<pmc_dir> ...]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    for f in glob.glob(os.path.join(stats_dir, "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(out, "%s_kernel_stats.csv" % tag))
    summary = {}
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            agg = collections.defaultdict(lambda: collections.defaultdict(float))
            launches = collections.defaultdict(set)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("irs_hip::", "")
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k].add(r["Dispatch_Id"])
            for k, v in agg.items():
                n = max(1, len(launches[k]))
                summary.setdefault(k, {"launches": n})
                for c, x in v.items():
                    summary[k][c + "_per_launch"] = x / n
    json.dump(summary, open(os.path.join(out, "%s_pmc.json" % tag), "w"), indent=1, sort_keys=True)
    # the kernels the roofline prices (bench.py): the work-item path's k_score, or the joined
    # path's two stages k_join + k_join_score (the pilot kernels are apart)
    staged = [k for k in summary if (k.startswith("k_join<") or k.startswith("k_join_score"))
              and "FETCH_SIZE_per_launch" in summary[k]]
    scored = [k for k in summary if k.startswith("k_score") and "FETCH_SIZE_per_launch" in summary[k]]
    for group in ([staged] if len(staged) == 2 else [[k] for k in scored]):
        if group:
            k = " + ".join(sorted(group))
            fetch_kb = sum(summary[g]["FETCH_SIZE_per_launch"] for g in group)
            write_kb = sum(summary[g].get("WRITE_SIZE_per_launch", 0.0) for g in group)
            t = {
                "kernel": k, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), %s" % tag,
                "fetch_bytes_raw": fetch_kb * 1024, "write_bytes": write_kb * 1024,
                # MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide
                # coalesced streams; our loads are 8 B/lane (uncalibrated width) so both bounds are given
                "fetch_bytes_x2_correction": 2 * fetch_kb * 1024,
                "bytes": 2 * fetch_kb * 1024 + write_kb * 1024,
                "kernel_sources_sha": __import__("bench").kernel_sources_sha(),
                "unit": "bytes per launch",
            }
            json.dump(t, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
            print(json.dumps(t))
    print("wrote", os.listdir(out))


if __name__ == "__main__":
    main()
