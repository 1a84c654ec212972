set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_gputests.log 2>&1; echo "gputests rc=$?"
tail -5 gpurun_out/r02c_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; echo "bench rc=$?"
cat gpurun_out/r02c_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02c_stats -o r02c --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/r02c_bench_prof.json 2> $O/r02c_bench_prof.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d $O/r02c_pmc_2 -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/r02c_pmc_2.log 2>&1
cd $R
head -12 gpurun_out/r02c_stats/*kernel_stats.csv | cut -c1-160
( timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --wand 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --wand 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --wand --scorer tfidf --clustered 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op or --terms 8 --configs 12288:64 --nocheck --wand --clustered 2>&1 | tail -4
  timeout 200 python tools/sweep.py --op mm --terms 4 --configs 8192:64 --nocheck 2>&1 | tail -3 ) > gpurun_out/r02c_sweep.txt 2>&1
cat gpurun_out/r02c_sweep.txt
