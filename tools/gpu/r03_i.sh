cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
for X in "" "--plan-ahead"; do
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench $X:', d['value'], d['ms_per_step'], d['value_with_results_on_host'], d['roofline']['kernel_ms'])"
done
