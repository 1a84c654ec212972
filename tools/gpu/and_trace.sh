# tools/gpu/and_trace.sh — per-kernel times of a conjunction sweep (which kernels the units took)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for P in ${PATHS:-auto}; do
  rm -rf /tmp/andtr; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/andtr -o t --output-format csv -- python $R/tools/sweep.py --op ${OP:-and} --terms ${TERMS:-2} --configs 8192:64 --path $P --steps 5 > $R/gpurun_out/and_trace_$P.log 2>&1
  F=$(find /tmp/andtr -name "*kernel_stats.csv" | head -1); echo "== $P $F"; head -12 "$F" | cut -c1-150
done
