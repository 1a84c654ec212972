cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --touched 2>&1 | grep "step\|touched\|rror"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck 2>&1 | grep "step\|rror"
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck --touched 2>&1 | grep "step\|touched\|rror" ) 2>&1
