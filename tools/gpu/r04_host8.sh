#!/bin/bash
# host stages of a fresh 8-segment batch (IRS_HIP_TRACE)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IRS_HIP_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 2 --force-segments --no-cpu > gpurun_out/host8.json 2> gpurun_out/host8.log
grep "irs_hip\]" gpurun_out/host8.log | tail -26
python -c "
import json; d=json.load(open('gpurun_out/host8.json')); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
