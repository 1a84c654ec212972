cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
( timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 2>&1 | grep "step\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 2>&1 | grep "step\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 3 --k 100 --lo-rank 4 --hi-rank 512 --configs 8192:64 --nocheck 2>&1 | grep "step\|rror" ) > $O/r02x_sweep.txt 2>&1
cat $O/r02x_sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phrase or pushdown or config5" 2>&1 | tail -2
