#!/bin/bash
# gpurun -- 'bash tools/gpu/r04_stage2.sh': the two-pass joined path (fast.h) — GPU tier, the
# headline bench with it and with the one-pass exact kernel (IRS_HIP_FAST16=0), kernel stats.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r04b_gputests.log 2>&1; echo "gpu tests rc=$?"
tail -3 $O/r04b_gputests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r04b_bench_fast.json 2> $O/r04b_bench_fast.err
cat $O/r04b_bench_fast.json
IRS_HIP_FAST16=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r04b_bench_exact.json 2> $O/r04b_bench_exact.err
python -c "import json;d=json.load(open('$O/r04b_bench_exact.json'));print('exact:',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/r04b_stats -o r04b --output-format csv -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $O/r04b_bench_prof.json 2> $O/r04b_bench_prof.err
f=$(ls $O/r04b_stats/*/*kernel_stats.csv $O/r04b_stats/*kernel_stats.csv 2>/dev/null | head -1)
head -12 "$f" | cut -c1-160
