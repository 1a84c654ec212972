#!/bin/bash
# tools/collect_round4.sh TAG — after `gpurun -- 'bash tools/profile_round4.sh TAG'`: copies the
# summaries the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).
set -e
TAG=$1
cd "$(dirname "$0")/.."
G=gpurun_out
python tools/summarize_prof.py $TAG $G/${TAG}_stats $G/${TAG}_pmc_1 $G/${TAG}_pmc_2 $G/${TAG}_pmc_3 $G/${TAG}_pmc_4 $G/${TAG}_pmc_5 $G/${TAG}_pmc_6
cp $G/${TAG}_bench_plain.json profiles/${TAG}_bench_n1.json
cp $G/${TAG}_bench.json profiles/${TAG}_bench_n1_under_rocprof.json
cp $G/${TAG}_bench_replay.json profiles/${TAG}_bench_n1_replayed_batches.json
cp $G/${TAG}_gpu_tests.txt profiles/${TAG}_gpu_tests.txt
cp $G/${TAG}_bench_seg.json profiles/${TAG}_bench_8seg_n1.json && cp $G/${TAG}_stats_seg/*kernel_stats.csv profiles/${TAG}_kernel_stats_8seg.csv
cp $G/${TAG}_bench_c5.json profiles/${TAG}_bench_config5_n1.json && cp $G/${TAG}_stats_c5/*kernel_stats.csv profiles/${TAG}_kernel_stats_config5.csv
ls profiles | grep $TAG
