#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${TAG:-r04t}
TAG=$T bash tools/gpu/r04_last.sh
timeout 900 python bench.py --config 5 --steps 5 --warmup 5 > gpurun_out/${T}_bench_c5.json 2> gpurun_out/${T}_bench_c5.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_c5.json')); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'], d['config']['reruns_in_timed_steps'])"
