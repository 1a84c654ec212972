set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "conj or and or config5 or wand or golden or boolean" > $O/r02j_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/r02j_gputests.log
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --touched 2>&1 | tail -4
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --touched 2>&1 | tail -4
  timeout 300 python tools/sweep.py --op and --terms 3 --scorer tfidf --configs 8192:64 --nocheck --touched 2>&1 | tail -4 ) > $O/r02j_sweep.txt 2>&1
cat $O/r02j_sweep.txt
