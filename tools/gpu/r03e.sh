cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base ilp memc iter; do
  cp gpurun_variants/libirs_hip_$v.so iresearch_amd/csrc/libirs_hip.so
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  timeout 200 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck 2>&1 | grep step | cut -c1-110
done
