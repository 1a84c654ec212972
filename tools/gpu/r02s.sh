cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
( timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck 2>&1 | grep "step\|touched\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 2>&1 | grep "step\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 3 --k 100 --lo-rank 4 --hi-rank 512 --configs 8192:64 --nocheck 2>&1 | grep "step\|touched\|rror" ) > $O/r02s_sweep.txt 2>&1
cat $O/r02s_sweep.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['first_run_ms_per_set'])"
