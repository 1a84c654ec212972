cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu > $O/r02v_bench.json 2> $O/r02v_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/r02v_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or all_scorers or many_items or ragged or tiles" 2>&1 | tail -2
