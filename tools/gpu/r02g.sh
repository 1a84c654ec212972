set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02g_gputests.log 2>&1; echo "gputests rc=$?"
tail -4 gpurun_out/r02g_gputests.log
( timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_conj8.so 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_conj8.so 2>&1 | tail -3
  timeout 200 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck 2>&1 | tail -3 ) > gpurun_out/r02g_sweep.txt 2>&1
cat gpurun_out/r02g_sweep.txt
timeout 900 python bench.py --config 5 --steps 5 --warmup 2 > gpurun_out/r02g_bench_c5.json 2> gpurun_out/r02g_bench_c5.err; echo "c5 rc=$?"
cat gpurun_out/r02g_bench_c5.json; tail -3 gpurun_out/r02g_bench_c5.err
