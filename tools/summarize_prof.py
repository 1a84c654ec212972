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


def config5(tag, pmc_dirs):
    """FETCH_SIZE / WRITE_SIZE passes over `bench.py --config 5` -> profiles/<tag>_pmc_config5.json
    (per kernel and launch) and profiles/traffic_config5_latest.json: HBM-side bytes per STEP (one
    AND batch + one phrase batch) of the kernels the config-5 roofline prices — summed over their
    launches of the profiled run, divided by its steps (two k_select launches per step)."""
    out = os.path.join(ROOT, "profiles")
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("irs_hip::", "")
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    summary = {k: {c + "_per_launch": v / max(1, len(launches[k][c])) for c, v in cs.items()}
               for k, cs in tot.items()}
    for k in summary:
        summary[k]["launches"] = max(len(x) for x in launches[k].values())
    json.dump(summary, open(os.path.join(out, "%s_pmc_config5.json" % tag), "w"), indent=1, sort_keys=True)
    steps = {c: len(launches["k_select"][c]) / 2.0 for c in ("FETCH_SIZE", "WRITE_SIZE") if launches["k_select"][c]}
    priced = [k for k in tot if k.startswith(("k_conj<", "k_phrase", "k_join_score", "k_join_rescore", "k_join<"))]
    fetch = sum(tot[k]["FETCH_SIZE"] for k in priced) * 1024 / max(steps.get("FETCH_SIZE", 1.0), 1.0)
    write = sum(tot[k]["WRITE_SIZE"] for k in priced) * 1024 / max(steps.get("WRITE_SIZE", 1.0), 1.0)
    t = {"kernels": sorted(priced), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) "
         "over bench.py --config 5, %s" % tag, "fetch_bytes_raw": fetch, "write_bytes": write,
         "fetch_bytes_x2_correction": 2 * fetch, "bytes": 2 * fetch + write,
         "kernel_sources_sha": __import__("bench").kernel_sources_sha(), "unit": "bytes per step (2000 queries)"}
    json.dump(t, open(os.path.join(out, "traffic_config5_latest.json"), "w"), indent=1)
    print(json.dumps(t))


def main():
    if sys.argv[1] == "--config5":
        return config5(sys.argv[2], sys.argv[3:])
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
    # (paired tiles: k_join_score<2> + k_join_rescore are the score stage)
    staged = [k for k in summary if k.startswith(("k_join<", "k_join_score", "k_join_rescore"))
              and "FETCH_SIZE_per_launch" in summary[k]]
    scored = [k for k in summary if k.startswith("k_score") and "FETCH_SIZE_per_launch" in summary[k]]
    for group in ([staged] if len(staged) in (2, 3) else [[k] for k in scored]):
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
