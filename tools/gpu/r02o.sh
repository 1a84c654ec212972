set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02o_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/r02o_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r02o_bench.json 2> $O/r02o_bench.err; echo "bench rc=$?"
cat $O/r02o_bench.json
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 > $O/r02o_bench_c5.json 2> $O/r02o_bench_c5.err; echo "bench c5 rc=$?"
cat $O/r02o_bench_c5.json
