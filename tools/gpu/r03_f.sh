cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python bench.py --no-cpu --steps 20 --warmup 4 > $O/r03f_bench.json 2> $O/r03f_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/r03f_bench.json')); print(d['value'], d['ms_per_step'], d['value_with_results_on_host'], d['roofline']['frac'], d['roofline']['dominant_kernel'], d['roofline']['kernel_ms'])"
