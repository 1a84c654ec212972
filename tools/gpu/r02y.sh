cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
for d in 1250000 2500000; do
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu --docs $d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print($d, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/r02y_stats -o r02y --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --docs 1250000 > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$GRAFT_REPO_ROOT/$O/r02y_stats/*kernel_stats.csv")[0])))
for r in rows[:16]:
    print(r['Name'][:60], r['Calls'], "%.1f us"%(float(r['AverageNs'])/1e3), r['Percentage'])
PY
