#!/bin/bash
# round 4, stage 3: the whole GPU tier at the current sources + the headline bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04e_gpu_tests.txt 2>&1
tail -5 gpurun_out/r04e_gpu_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04e_bench_n1.json 2> gpurun_out/r04e_bench_n1.log
tail -c 1500 gpurun_out/r04e_bench_n1.json
