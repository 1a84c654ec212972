#!/bin/bash
# tools/gpu/kstats.sh TAG — bench.py under --kernel-trace --stats, the per-kernel averages in ms
TAG=${1:-k}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R="$(cd "$(dirname "$0")/../.." && pwd)"
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- \
  python $R/bench.py --steps 5 --warmup 3 --no-cpu ${BENCH_ARGS} > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python - <<PY
import csv, json
d = json.load(open("$O/${TAG}_bench.json"))
print("$TAG", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("kernel_ms"))
for r in csv.DictReader(open("$O/${TAG}_stats/${TAG}_kernel_stats.csv")):
    if int(r["Calls"]) >= 5 and "rocclr" not in r["Name"] and "at::" not in r["Name"]:
        print("  %-28s %3s x %8.4f ms" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
