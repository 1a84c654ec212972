set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rccl or legacy or zero_boost or wand or config5" > gpurun_out/r02f_gputests.log 2>&1; echo "gputests rc=$?"
tail -5 gpurun_out/r02f_gputests.log
( timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --wand 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck 2>&1 | tail -3 ) > gpurun_out/r02f_sweep.txt 2>&1
cat gpurun_out/r02f_sweep.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02f_stats -o r02f --output-format csv -- python $R/tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --steps 5 > $O/r02f_and3.log 2>&1
head -6 $O/r02f_stats/*kernel_stats.csv | cut -c1-150
