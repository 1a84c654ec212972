cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -x -q > $O/final_gputests.log 2>&1; echo "gputests rc=$?"; tail -2 $O/final_gputests.log
timeout 300 python bench.py > $O/final_bench.json 2> $O/final_bench.err; echo "bench rc=$?"
cat $O/final_bench.json
