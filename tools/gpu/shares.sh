# tools/gpu/shares.sh — what ONE rank does per step at N = 1 / 2 / 4 / 8 GPUs, measured on one
# GPU with that rank's share of the index: 8 / 4 / 2 / 1 segments of 1.25 M docs.
# EXTRA="--no-shared-threshold" for the A/B of irs_hip_batch_set_shared_threshold.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
for cfg in ${CFGS:-"10000000:8" "5000000:4" "2500000:2" "1250000:1"}; do
  for X in "" $EXTRA; do
    timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu --docs ${cfg%:*} --segments ${cfg#*:} --force-segments $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('docs ${cfg%:*} segments ${cfg#*:} $X:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
