#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TAG:-r04i}
timeout 900 python bench.py --config 5 --steps ${STEPS:-5} --warmup ${WARMUP:-5} --no-cpu > gpurun_out/${T}_bench_config5_n1.json 2> gpurun_out/${T}_bench_config5_n1.log
python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_config5_n1.json"))
print(d["ms_per_step"], d["value"], d["roofline"])
PY
