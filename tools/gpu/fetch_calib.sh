#!/bin/bash
# gpurun -- 'bash tools/gpu/fetch_calib.sh': FETCH_SIZE of a known 4 GB streaming read, 4-byte and
# 16-byte loads per lane (tools/micro/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch_calib; mkdir -p $O
$R/gpurun_variants/fetch_calib > $O/plain.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc -o fc --output-format csv -- $R/gpurun_variants/fetch_calib > $O/pmc.log 2>&1
python3 - <<PY > $R/gpurun_out/fetch_calib.txt
import csv, glob
rows = []
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print(open("$O/plain.txt").read())
acc = {}
for r in rows:
    if r.get("Counter_Name") == "FETCH_SIZE":
        acc.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
for k, v in acc.items():
    kb = sum(v) / len(v)
    print("%s: FETCH_SIZE %.0f KB per launch = %.3f GB reported for 4.295 GB read -> factor %.3f" % (k, kb, kb * 1024 / 1e9, 4.294967296e9 / (kb * 1024)))
PY
cat $R/gpurun_out/fetch_calib.txt
