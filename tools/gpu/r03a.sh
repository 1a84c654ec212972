cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r03a_bench.json 2> $O/r03a_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/r03a_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-plan-ahead 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('no plan-ahead', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --force-segments 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('8seg', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "plan_ahead or config3" 2>&1 | tail -2
