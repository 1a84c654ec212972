# FETCH_SIZE / WRITE_SIZE passes of the headline bench at the current sources
# (profiles/traffic_latest.json via tools/summarize_prof.py)
TAG=${1:-r02y}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o $TAG --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $O/${TAG}_pmc_$i -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu > $O/${TAG}_pmc_$i.log 2>&1
done
cat $O/${TAG}_bench.json
