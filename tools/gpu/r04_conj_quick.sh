#!/bin/bash
# quick A/B of k_conj / k_phrase: AND and phrase sweeps only (no config 5)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/${TAG:-r04h}_conj_quick.txt; : > $O
run() { echo "== tools/sweep.py $*" >> $O; timeout 400 python tools/sweep.py "$@" --configs 8192:64 2>&1 | grep -E "tile=|WAND" >> $O; }
run --op and --terms 2 --path items --nocheck
run --op and --terms 3 --path items
run --op and --terms 3 --scorer tfidf --wand --path items
[ -n "$PHRASE" ] && run --op phrase --terms 2 --k 100
cat $O
