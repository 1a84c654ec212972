cd $GRAFT_REPO_ROOT
for t in 256 512 1024; do
  IRS_HIP_WG_THREADS=$t timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('threads $t', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
