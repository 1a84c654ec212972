#!/bin/bash
# config 5's AND batch: conjunctions block driven / all joined / by the cost rule
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in 0 1 auto; do
  if [ $V = auto ]; then unset IRS_HIP_JOIN_AND; else export IRS_HIP_JOIN_AND=$V; fi
  timeout 600 python bench.py --config 5 --steps 3 --warmup 5 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('IRS_HIP_JOIN_AND=$V', d['ms_per_step'], d['roofline']['stage_ms']['and'], d['config']['reruns_in_timed_steps'])"
done
