set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02h_gputests.log 2>&1; echo "gputests rc=$?"
grep "full vocabulary" gpurun_out/r02h_gputests.log; tail -3 gpurun_out/r02h_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02h_bench.json 2> gpurun_out/r02h_bench.err; echo "bench rc=$?"
cat gpurun_out/r02h_bench.json
( timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op phrase --terms 3 --k 100 --lo-rank 4 --hi-rank 512 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck 2>&1 | tail -3 ) > gpurun_out/r02h_sweep.txt 2>&1
cat gpurun_out/r02h_sweep.txt
