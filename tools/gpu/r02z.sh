cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02z_gputests.log 2>&1; echo "gputests rc=$?"; tail -3 $O/r02z_gputests.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 > $O/r02z_bench_c5.json 2> $O/r02z_bench_c5.err; echo "bench c5 rc=$?"
cat $O/r02z_bench_c5.json; tail -3 $O/r02z_bench_c5.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
