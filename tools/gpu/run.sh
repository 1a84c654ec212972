#!/bin/bash
# tools/gpu/run.sh — ONE parameterised entry for everything that runs on the GPU box
# (gpurun -- 'TAG=r05a bash tools/gpu/run.sh tests tune bench'); stages run in the order given,
# outputs land in gpurun_out/<TAG>_*.  Counter passes (pmc*) are separate rocprofv3 runs, never
# combined with tracing.  tools/collect_round.sh TAG turns them into the files under profiles/.
#
#   tests          pytest -m gpu                                   -> TAG_gpu_tests.txt
#   smoke          __graft_entry__.smoke()
#   tune           tools/join_tune.py --runs "$RUNS"               -> TAG_tune.txt
#   bench          bench.py as the driver runs it (cpu baseline)   -> TAG_bench_plain.json
#   quick          bench.py --no-cpu, short                        -> TAG_bench_quick.json
#   stats          bench.py under --kernel-trace --stats           -> TAG_stats/
#   pmc            the six counter groups over bench.py            -> TAG_pmc_<i>/
#   pmc_tune       the wide counter set over join_tune ($RUNS)     -> TAG_pmc/ + TAG_pmc.txt
#   replay         bench.py --query-sets 4 (rounds 1-3 protocol)   -> TAG_bench_replay.json
#   seg            bench.py --force-segments under --stats         -> TAG_bench_seg.json
#   shares         one rank's share at N = 1/2/4/8 ($CFGS, $EXTRA) -> TAG_shares.txt
#   c5             bench.py --config 5                             -> TAG_bench_c5.json
#   c5stats        ... under --kernel-trace --stats                -> TAG_stats_c5/
#   c5pmc          FETCH_SIZE / WRITE_SIZE passes over config 5    -> TAG_c5pmc_<i>/
#   sweeps         tools/sweep.py shapes + tools/cost_sweep.py     -> TAG_sweeps.txt
#   tasks          bench.py --tasks (the reference's task classes) -> TAG_tasks.txt
#   calib          tools/micro/fetch_calib.hip (stream ceiling)    -> TAG_fetch_calib.txt
#   cmd            eval "$CMD"                                     -> TAG_cmd.txt
TAG=${TAG:-r05}
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R="$(cd "$(dirname "$0")/../.." && pwd)"
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RUNS=${RUNS:-base:exact,base:pruned}
py() { python "$@"; }
line() { python -c "
import json,sys
d=json.load(open('$1'))
r=d.get('roofline') or {}
print('$1'.split('/')[-1], d['value'], d['ms_per_step'], r.get('frac'), r.get('kernel_ms'), d.get('value_with_results_on_host'), (d.get('config') or {}).get('first_run_ms_per_set'))"; }
pmc_groups=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
            "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
for stage in "$@"; do
  echo "== $stage"
  case $stage in
    tests) (cd $R && timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS} > $O/${TAG}_gpu_tests.txt 2>&1; tail -3 $O/${TAG}_gpu_tests.txt) ;;
    smoke) (cd $R && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) ;;
    tune) (cd $R && timeout 900 python tools/join_tune.py --runs $RUNS ${TUNE_ARGS} 2>&1 | grep -E "path|Error|error" | tee $O/${TAG}_tune.txt) ;;
    bench) py $R/bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 > $O/${TAG}_bench_plain.json 2> $O/${TAG}_bench_plain.err; line $O/${TAG}_bench_plain.json ;;
    quick) py $R/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu ${BENCH_ARGS} > $O/${TAG}_bench_quick.json 2> $O/${TAG}_bench_quick.err; line $O/${TAG}_bench_quick.json ;;
    stats) rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- \
             python $R/bench.py --steps 5 --warmup 3 --no-cpu > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; line $O/${TAG}_bench.json ;;
    pmc) i=0; for grp in "${pmc_groups[@]}"; do i=$((i+1))
           rocprofv3 --pmc $grp -d $O/${TAG}_pmc_$i -o p --output-format csv -- \
             python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_pmc_$i.log 2>&1; echo "group $i rc=$?"; done ;;
    pmc_tune) bash $R/tools/gpu/pmc_join.sh $TAG $RUNS ;;
    replay) py $R/bench.py --steps 20 --warmup 5 --no-cpu --query-sets 4 > $O/${TAG}_bench_replay.json 2>/dev/null; line $O/${TAG}_bench_replay.json ;;
    seg) rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_seg -o ${TAG}seg --output-format csv -- \
           python $R/bench.py --steps 5 --warmup 2 --force-segments --no-cpu > $O/${TAG}_bench_seg.json 2> $O/${TAG}_bench_seg.err; line $O/${TAG}_bench_seg.json ;;
    shares) bash $R/tools/gpu/shares.sh 2>/dev/null | tee $O/${TAG}_shares.txt ;;
    c5) py $R/bench.py --config 5 --steps ${STEPS:-5} --warmup 5 ${C5_ARGS} > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err; line $O/${TAG}_bench_c5.json ;;
    c5stats) rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_c5 -o ${TAG}c5 --output-format csv -- \
               python $R/bench.py --config 5 --steps 3 --warmup 4 --no-cpu > $O/${TAG}_bench_c5_rocprof.json 2> $O/${TAG}_bench_c5_rocprof.err; line $O/${TAG}_bench_c5_rocprof.json ;;
    c5pmc) i=0; for grp in "FETCH_SIZE" "WRITE_SIZE"; do i=$((i+1))
             rocprofv3 --pmc $grp -d $O/${TAG}_c5pmc_$i -o p --output-format csv -- \
               python $R/bench.py --config 5 --steps 2 --warmup 2 --no-cpu > $O/${TAG}_c5pmc_$i.log 2>&1; echo "group $i rc=$?"; done ;;
    sweeps) TAG=$TAG bash $R/tools/gpu/sweeps.sh $TAG > /dev/null 2>&1
            S=$O/${TAG}_sweeps.txt
            echo "== tools/cost_sweep.py (plain disjunctions: work items / joined / what PATH_AUTO takes; d = no term shared by two queries)" >> $S
            (cd $R && timeout 900 python tools/cost_sweep.py --docs 10000000 --shapes 1000x4d,128x8d,16x8d,1000x1d,128x1d,16x1d,1000x8s,128x8s,16x8s 2>&1 | grep -v amdgpu >> $S)
            tail -30 $S ;;
    tasks) (cd $R && timeout 900 python bench.py --tasks ${TASKS_ARGS} 2> $O/${TAG}_tasks.err | tee $O/${TAG}_tasks.txt) ;;
    calib) (cd $R && bash tools/gpu/fetch_calib.sh 2>&1 | tee $O/${TAG}_fetch_calib.txt) ;;
    cmd) (cd $R && eval "$CMD" 2>&1 | tee $O/${TAG}_cmd.txt) ;;
    *) echo "unknown stage $stage" ;;
  esac
done
