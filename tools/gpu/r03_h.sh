cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
for cfg in "10000000 1 1000" "1250000 1 1000" "1250000 1 10" ; do
  set -- $cfg
  X=""; [ "$1" != "10000000" ] && X="--force-segments"
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu --docs $1 --segments $2 --k $3 $X 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('docs $1 segments $2 k $3:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
