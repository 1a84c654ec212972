#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r01c'): the evidence behind
# bench.py's roofline numbers.  Kernel statistics and counters are separate rocprofv3
# runs (PMC passes are never combined with tracing).  Outputs land in gpurun_out/<tag>_*;
# `python tools/summarize_prof.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_pmc_*`
# turns them into the committed files under profiles/.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- \
  python $R/bench.py --steps 5 --warmup 1 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_seg -o ${TAG}seg --output-format csv -- \
  python $R/bench.py --steps 3 --warmup 1 --force-segments --no-cpu > $O/${TAG}_bench_seg.json 2> $O/${TAG}_bench_seg.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/${TAG}_pmc_$i -o p --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_pmc_$i.log 2>&1
done
cat $O/${TAG}_bench.json $O/${TAG}_bench_seg.json
