#!/bin/bash
# round 4: plain disjunctions, joined streams against work items
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/cost_sweep.py --docs 10000000 --shapes 1000x4d,128x8d,16x8d,1000x1d,128x1d,16x1d,1000x8s,128x8s,16x8s,16x2d,1x8d > gpurun_out/r04e_cost_sweep.txt 2>&1
cat gpurun_out/r04e_cost_sweep.txt
timeout 600 python tools/cost_sweep.py --docs 2000000 --mean-len 1000 --shapes 1000x8s,128x8d > gpurun_out/r04e_cost_sweep_len1000.txt 2>&1
cat gpurun_out/r04e_cost_sweep_len1000.txt
