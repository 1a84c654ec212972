cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r02t_bench.json 2> $O/r02t_bench.err; echo "bench rc=$?"
cat $O/r02t_bench.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --force-segments > $O/r02t_bench_seg.json 2> $O/r02t_bench_seg.err; echo "bench seg rc=$?"
cat $O/r02t_bench_seg.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02t_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/r02t_gputests.log
