set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_gputests.log 2>&1; echo "gputests rc=$?"
tail -5 gpurun_out/r02d_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; echo "bench rc=$?"
cat gpurun_out/r02d_bench.json
( timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --wand 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op and --terms 3 --lo-rank 256 --configs 8192:64 --nocheck --wand 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --wand --scorer tfidf --clustered --k 10 2>&1 | tail -4 ) > gpurun_out/r02d_sweep.txt 2>&1
cat gpurun_out/r02d_sweep.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02d_stats -o r02d --output-format csv -- python $R/tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck > $O/r02d_and3.log 2>&1
head -8 $O/r02d_stats/*kernel_stats.csv | cut -c1-140
