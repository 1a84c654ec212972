#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round4.sh r04k'): the evidence behind the
# round-4 numbers.  Kernel statistics and counters are separate rocprofv3 runs (PMC passes are never
# combined with tracing).  Outputs land in gpurun_out/<tag>_*; `tools/collect_round4.sh <tag>` turns
# them into the committed files under profiles/.
TAG=${1:-r04k}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
(cd $R && timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.txt 2>&1; tail -2 $O/${TAG}_gpu_tests.txt)
# headline bench exactly as the driver runs it (with the cpu baseline), then under --kernel-trace --stats
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_plain.json 2> $O/${TAG}_bench_plain.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/${TAG}_pmc_$i -o p --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_pmc_$i.log 2>&1
done
# the replay protocol of rounds 1-3 (A/B of the fresh-batch headline), 8 segments on one GPU, config 5
python $R/bench.py --steps 20 --warmup 5 --no-cpu --query-sets 4 > $O/${TAG}_bench_replay.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_seg -o ${TAG}seg --output-format csv -- \
  python $R/bench.py --steps 5 --warmup 2 --force-segments --no-cpu > $O/${TAG}_bench_seg.json 2> $O/${TAG}_bench_seg.err
python $R/bench.py --config 5 --steps 5 --warmup 5 > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_c5 -o ${TAG}c5 --output-format csv -- \
  python $R/bench.py --config 5 --steps 3 --warmup 4 --no-cpu > $O/${TAG}_bench_c5_rocprof.json 2> $O/${TAG}_bench_c5_rocprof.err
cat $O/${TAG}_bench_plain.json
