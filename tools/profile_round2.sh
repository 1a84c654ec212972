#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round2.sh r02u'): the evidence behind the
# round-2 numbers.  Kernel statistics and counters are separate rocprofv3 runs (PMC passes are
# never combined with tracing).  Outputs land in gpurun_out/<tag>_*; `tools/collect_round2.sh
# <tag>` turns them into the committed files under profiles/.
TAG=${1:-r02x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
# headline bench (with the cpu baseline) without a profiler, then under --kernel-trace --stats
python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_plain.json 2> $O/${TAG}_bench_plain.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_seg -o ${TAG}seg --output-format csv -- \
  python $R/bench.py --steps 3 --warmup 1 --force-segments --no-cpu > $O/${TAG}_bench_seg.json 2> $O/${TAG}_bench_seg.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_c5 -o ${TAG}c5 --output-format csv -- \
  python $R/bench.py --config 5 --steps 3 --warmup 1 > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/${TAG}_pmc_$i -o p --output-format csv -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_pmc_$i.log 2>&1
done
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/${TAG}_pmcand_$i -o p --output-format csv -- \
    python $R/tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --steps 2 > $O/${TAG}_pmcand_$i.log 2>&1
done
cd $R
( for a in "--op and --terms 3" "--op and --terms 2" "--op and --terms 4" "--op and --terms 3 --scorer tfidf --wand" \
           "--op mm --terms 4" "--op or --terms 8 --scorer tfidf" "--op or --terms 2 --k 100" \
           "--op phrase --terms 2 --k 100" "--op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000" \
           "--op phrase --terms 3 --k 100 --lo-rank 4 --hi-rank 512"; do
    echo "== tools/sweep.py $a --touched"
    python tools/sweep.py $a --configs 8192:64 --nocheck --touched 2>&1 | grep "step\|touched\|WAND\|hits/query"
  done ) > $O/${TAG}_sweeps.txt 2>&1
cat $O/${TAG}_bench_plain.json $O/${TAG}_bench_seg.json $O/${TAG}_bench_c5.json
