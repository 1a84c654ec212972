#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IRS_HIP_TRACE=1 timeout 900 python bench.py --config 5 --steps 2 --warmup 2 --no-cpu > gpurun_out/c5_trace.json 2> gpurun_out/c5_trace.log
grep -v amdgpu gpurun_out/c5_trace.log | tail -60
