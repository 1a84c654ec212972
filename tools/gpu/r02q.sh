cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 > $O/r02q_bench_c5.json 2> $O/r02q_bench_c5.err; echo "bench c5 rc=$?"
cat $O/r02q_bench_c5.json
timeout 300 python tools/first_run_cost.py 2>&1 | tail -4
