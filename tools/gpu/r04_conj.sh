#!/bin/bash
# round 4: k_conj / k_phrase: AND and phrase sweeps + BASELINE config 5
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TAG:-r04f}
O=gpurun_out/${T}_conj.txt; : > $O
run() {   # one sweep line per invocation, as profiles/r03c_sweeps.txt has them
  echo "== tools/sweep.py $*" >> $O
  timeout 400 python tools/sweep.py "$@" --configs 8192:64 2>&1 | grep -E "tile=|path:|touched|hits/query|WAND" >> $O
}
run --op and --terms 2 --path items --touched
run --op and --terms 3 --path items --touched
run --op phrase --terms 2 --k 100 --touched
run --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --touched
cat $O
timeout 900 python bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/${T}_bench_config5_n1.json 2> gpurun_out/${T}_bench_config5_n1.log
python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench_config5_n1.json"))
print(d["ms_per_step"], d["roofline"])
PY
