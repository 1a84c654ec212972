#!/bin/bash
# round 4: where k_phrase's time goes — the sweeps under ablation builds (tools/build_variant.sh)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/${TAG:-r04g}_phrase_abl.txt; : > $O
for V in ${VARIANTS:-base nomerge}; do
  LIB=""; [ $V != base ] && LIB="--lib gpurun_variants/libirs_hip_$V.so"
  echo "== $V: phrase 2 terms" >> $O
  timeout 400 python tools/sweep.py --op phrase --terms 2 --k 100 --configs 8192:64 --nocheck $LIB 2>&1 | grep -E "tile=" >> $O
  echo "== $V: phrase 2 terms, ranks 1..64, 2 M docs" >> $O
  timeout 400 python tools/sweep.py --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --configs 8192:64 --nocheck $LIB 2>&1 | grep -E "tile=" >> $O
done
cat $O
