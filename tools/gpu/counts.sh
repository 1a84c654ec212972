# tools/gpu/counts.sh — conjunctions / min-match as joined streams (match counts in the
# accumulators) against the block-driven / work-item kernels, and the headline beside them.
#   PATHS="items joined auto" bash tools/gpu/counts.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/${TAG:-counts}.log; : > $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "join_counts or paths_agree or wand_equals" >> $O 2>&1; echo "pytest rc=$?" >> $O
for P in ${PATHS:-items joined auto}; do
  for SH in ${SHAPES:-and:2 and:3 and:4 mm:4 mm:8}; do
    echo "== sweep --op ${SH%:*} --terms ${SH#*:} --path $P" >> $O
    timeout 300 python tools/sweep.py --op ${SH%:*} --terms ${SH#*:} --configs 8192:64 --path $P 2>&1 | grep -v amdgpu.ids | grep -E "tile=|path|hits/query" >> $O
  done
done
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])" >> $O
grep -v amdgpu.ids $O | tail -60
