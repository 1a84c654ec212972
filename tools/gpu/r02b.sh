set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_gputests.log 2>&1; echo "gputests rc=$?" 
tail -3 gpurun_out/r02b_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err; echo "bench rc=$?"
cat gpurun_out/r02b_bench.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02b_stats -o r02b --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/r02b_bench_prof.json 2> $O/r02b_bench_prof.err
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $O/r02b_pmc_$i -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/r02b_pmc_$i.log 2>&1
done
cd $R
cd $R
timeout 200 python tools/sweep.py --configs 12288:64 --nocheck 2>&1 | tail -3 > gpurun_out/r02b_sweep.txt
timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck 2>&1 | tail -3 >> gpurun_out/r02b_sweep.txt
timeout 200 python tools/sweep.py --scorer tfidf --configs 12288:64 --nocheck 2>&1 | tail -3 >> gpurun_out/r02b_sweep.txt
cat gpurun_out/r02b_sweep.txt
