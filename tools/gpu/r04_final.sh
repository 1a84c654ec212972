#!/bin/bash
# the round's final evidence: profile set + sweeps at the final sources
cd $GRAFT_REPO_ROOT
T=${TAG:-r04m}
bash tools/profile_round4.sh $T > /dev/null 2>&1
TAG=$T bash tools/gpu/r04_sweeps.sh > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
tail -2 gpurun_out/${T}_gpu_tests.txt
python -c "
import json; d=json.load(open('gpurun_out/${T}_bench_plain.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
