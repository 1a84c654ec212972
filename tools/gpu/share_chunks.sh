cd $GRAFT_REPO_ROOT
for c in 8 10 12 16 18 26 34 52; do
  IRS_HIP_JOIN_CHUNK=$c python bench.py --steps 20 --warmup 4 --no-cpu --docs 1250000 --segments 1 --force-segments 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('chunk $c:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
