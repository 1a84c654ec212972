#!/bin/bash
# tools/collect_round2.sh TAG — after `gpurun -- 'bash tools/profile_round2.sh TAG'`: copies the
# summaries the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
TAG=$1
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/summarize_prof.py $TAG $G/${TAG}_stats $G/${TAG}_pmc_1 $G/${TAG}_pmc_2 $G/${TAG}_pmc_3 $G/${TAG}_pmc_4
cp $G/${TAG}_bench_plain.json profiles/${TAG}_bench_n1.json
cp $G/${TAG}_bench.json profiles/${TAG}_bench_n1_under_rocprof.json
cp $G/${TAG}_bench_seg.json profiles/${TAG}_bench_8seg_n1.json
cp $G/${TAG}_bench_c5.json profiles/${TAG}_bench_config5_n1.json
cp $G/${TAG}_stats_seg/*kernel_stats.csv profiles/${TAG}_kernel_stats_8seg.csv
cp $G/${TAG}_stats_c5/*kernel_stats.csv profiles/${TAG}_kernel_stats_config5.csv
cp $G/${TAG}_sweeps.txt profiles/${TAG}_sweeps.txt
python - <<PY
import collections, csv, glob, json
summary = {}
for d in sorted(glob.glob("$G/${TAG}_pmcand_*")):
    if not d[-1].isdigit(): continue
    for f in glob.glob(d + "/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void irs_hip::", "")
            # the pilot pass (1/64 of the grid) and the full pass are the same kernel: keep them apart
            k += " grid=%s" % r["Grid_Size"]
            agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        per = collections.defaultdict(list)
        for (k, _), v in agg.items():
            per[k].append(v)
        for k, vs in per.items():
            summary.setdefault(k, {"launches": len(vs)})
            for c in vs[0]:
                summary[k][c + "_per_launch"] = sum(v[c] for v in vs) / len(vs)
json.dump(summary, open("profiles/${TAG}_pmc_and.json", "w"), indent=1, sort_keys=True)
print("wrote profiles/${TAG}_pmc_and.json")
PY
ls profiles | grep $TAG
