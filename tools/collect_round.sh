#!/bin/bash
# tools/collect_round.sh TAG — after `gpurun -- 'TAG=... bash tools/gpu/run.sh tests bench stats pmc
# replay seg c5 c5stats c5pmc ...'`: copies the summaries the judge reads from gpurun_out/ (scratch)
# into profiles/ (tracked).  Missing pieces are skipped.
TAG=$1
cd "$(dirname "$0")/.."
G=gpurun_out
[ -d $G/${TAG}_stats ] && python tools/summarize_prof.py $TAG $G/${TAG}_stats $G/${TAG}_pmc_1 $G/${TAG}_pmc_2 $G/${TAG}_pmc_3 $G/${TAG}_pmc_4 $G/${TAG}_pmc_5 $G/${TAG}_pmc_6
[ -d $G/${TAG}_c5pmc_1 ] && python tools/summarize_prof.py --config5 $TAG $G/${TAG}_c5pmc_1 $G/${TAG}_c5pmc_2
c() { [ -e "$1" ] && cp "$1" "$2"; }
c $G/${TAG}_bench_plain.json profiles/${TAG}_bench_n1.json
c $G/${TAG}_bench.json profiles/${TAG}_bench_n1_under_rocprof.json
c $G/${TAG}_bench_replay.json profiles/${TAG}_bench_n1_replayed_batches.json
c $G/${TAG}_gpu_tests.txt profiles/${TAG}_gpu_tests.txt
c $G/${TAG}_bench_seg.json profiles/${TAG}_bench_8seg_n1.json
for f in $G/${TAG}_stats_seg/*kernel_stats.csv; do c $f profiles/${TAG}_kernel_stats_8seg.csv; done
c $G/${TAG}_bench_c5.json profiles/${TAG}_bench_config5_n1.json
for f in $G/${TAG}_stats_c5/*kernel_stats.csv; do c $f profiles/${TAG}_kernel_stats_config5.csv; done
c $G/${TAG}_shares.txt profiles/${TAG}_shares.txt
c $G/${TAG}_sweeps.txt profiles/${TAG}_sweeps.txt
c $G/${TAG}_fetch_calib.txt profiles/${TAG}_fetch_calib.txt
c $G/${TAG}_tasks.txt profiles/${TAG}_tasks.txt
ls profiles | grep $TAG
