cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phrase or position or config5" > $O/r02r_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/r02r_gputests.log
( timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --touched 2>&1 | grep "step\|touched\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 --nocheck 2>&1 | grep "step\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 3 --k 100 --lo-rank 4 --hi-rank 512 --configs 8192:64 --nocheck --touched 2>&1 | grep "step\|touched\|rror" ) > $O/r02r_sweep.txt 2>&1
cat $O/r02r_sweep.txt
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 > $O/r02r_bench_c5.json 2> $O/r02r_bench_c5.err; echo "bench c5 rc=$?"
cat $O/r02r_bench_c5.json
PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 python tools/first_run_cost.py 2>&1 | tail -4
