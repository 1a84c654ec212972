# tools/gpu/sweeps.sh TAG — the other query shapes (tools/sweep.py) and a timed run over the full
# vocabulary, into gpurun_out/TAG_sweeps.txt (copied to profiles/ by hand).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/${1:-r03}_sweeps.txt; : > $O
run() { echo "== tools/sweep.py $*" >> $O; timeout 400 python tools/sweep.py "$@" --configs 8192:64 2>&1 | grep -v amdgpu.ids | grep -E "tile=|path:|hits/query|touched|WAND" >> $O; }
run --op and --terms 2 --path items --touched
run --op and --terms 2
run --op and --terms 3 --path items --touched
run --op and --terms 3
run --op and --terms 4
run --op and --terms 3 --scorer tfidf --wand
run --op mm --terms 4 --path items
run --op mm --terms 4
run --op mm --terms 8
run --op or --terms 8 --scorer tfidf
run --op or --terms 2 --k 100
run --op phrase --terms 2 --k 100 --touched
run --op phrase --terms 2 --k 100 --lo-rank 1 --hi-rank 64 --docs 2000000 --touched
echo "== bench.py --max-rank 1048576 (10 M docs, every one of the 2^20 ranks indexed)" >> $O
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu --max-rank 1048576 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: d[k] for k in ('value','ms_per_step','config','roofline')}))" >> $O
tail -50 $O
