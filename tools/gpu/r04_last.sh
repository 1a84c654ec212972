#!/bin/bash
# after the last kernel change of the round: GPU tests, the driver's bench command, kernel stats and
# the FETCH_SIZE / WRITE_SIZE passes profiles/traffic_latest.json is made from
cd $GRAFT_REPO_ROOT
T=${TAG:-r04q}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_tests.txt 2>&1; tail -1 gpurun_out/${T}_gpu_tests.txt
bash tools/gpu/traffic.sh $T > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_plain.json 2> gpurun_out/${T}_bench_plain.err
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_plain.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
