#!/bin/bash
# gpurun -- 'bash tools/gpu/r04_stage1.sh': GPU tier + the fresh-batch bench protocol against the
# replay protocol of rounds 1-3 (same library), one segment and eight.
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/r04a_gputests.log 2>&1; echo "gpu tests rc=$?" 
tail -3 $O/r04a_gputests.log
IRS_HIP_TRACE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r04a_bench_fresh.json 2> $O/r04a_bench_fresh.err
grep "irs_hip" $O/r04a_bench_fresh.err | tail -6
cat $O/r04a_bench_fresh.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --query-sets 4 > $O/r04a_bench_replay.json 2> $O/r04a_bench_replay.err
cat $O/r04a_bench_replay.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --force-segments > $O/r04a_bench_fresh_8seg.json 2> $O/r04a_bench_fresh_8seg.err
cat $O/r04a_bench_fresh_8seg.json
