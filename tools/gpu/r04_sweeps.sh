#!/bin/bash
# round 4: the other query shapes, the cost rule's sweep, the two joined forms, per-rank shares
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${TAG:-r04k}
bash tools/gpu/sweeps.sh $T > /dev/null 2>&1
O=gpurun_out/${T}_sweeps.txt
echo "== tools/cost_sweep.py (plain disjunctions: work items / joined / what PATH_AUTO takes; d = no term shared by two queries)" >> $O
timeout 900 python tools/cost_sweep.py --docs 10000000 --shapes 1000x4d,128x8d,16x8d,1000x1d,128x1d,16x1d,1000x8s,128x8s,16x8s 2>&1 | grep -v amdgpu >> $O
echo "== tools/cost_sweep.py --docs 2000000 --mean-len 1000 (1000-word docs)" >> $O
timeout 600 python tools/cost_sweep.py --docs 2000000 --mean-len 1000 --shapes 1000x8s,128x8d 2>&1 | grep -v amdgpu >> $O
echo "== tools/cost_sweep.py --docs 2000000 --mean-len 1000 --lo-rank 1   (queries that draw the most frequent terms: a term in EVERY doc has idf -> 0, the batch leaves the 32-bit accumulators and with them the joined path, whatever its frequencies; tf >= 64 on joined streams is pinned by case_join_edge_blocks)" >> $O
timeout 600 python tools/cost_sweep.py --docs 2000000 --mean-len 1000 --lo-rank 1 --shapes 1000x8s 2>&1 | grep -v amdgpu >> $O
echo "== tools/join_tune.py --runs base:items,base:exact,base:1024   (work items / one-pass joined / two-pass joined, fast.h)" >> $O
timeout 600 python tools/join_tune.py --runs base:items,base:exact,base:1024 2>&1 | grep -E "path" >> $O
echo "== tools/gpu/shares.sh (one rank's share of the index at N = 1 / 2 / 4 / 8)" >> $O
bash tools/gpu/shares.sh 2>/dev/null >> $O
echo "== tools/gpu/shares.sh EXTRA=--no-shared-threshold" >> $O
CFGS="1250000:1" bash tools/gpu/shares.sh 2>/dev/null >> $O
tail -40 $O
