#!/bin/bash
# A/B of k_join_score's two tile loops: shares of entries + barriers / wavefront-owned doc pieces
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/${TAG:-r04r}_owned.txt; : > $O
for OWNED in 0 1; do
  echo "== IRS_HIP_JOIN_OWNED=$OWNED" >> $O
  IRS_HIP_JOIN_OWNED=$OWNED timeout 600 python tools/join_tune.py --runs ${RUNS:-base:exact} 2>&1 | grep -E "path" >> $O
done
cat $O
