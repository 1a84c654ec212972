#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TAG:-r04l}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_tests.txt 2>&1; tail -2 gpurun_out/${T}_gpu_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > gpurun_out/${T}_bench_n1.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['first_run_ms_per_set'])"
bash tools/gpu/shares.sh 2>/dev/null
