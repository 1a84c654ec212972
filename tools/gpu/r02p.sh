cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r02p_stats_c5 -o r02p --output-format csv -- python $R/bench.py --config 5 --steps 3 --warmup 1 > $O/r02p_bench_c5.json 2> $O/r02p_bench_c5.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02p_stats -o r02p --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $O/r02p_bench.json 2> $O/r02p_bench.err
cat $O/r02p_bench.json
ls $O/r02p_stats_c5 $O/r02p_stats
