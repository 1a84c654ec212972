# tools/gpu/c5_abl_pmc.sh — instruction counters of config 5's kernels under ablation builds
# (gpurun_variants/libirs_hip_NAME.so): VARIANTS="base a b" TAG=x bash tools/gpu/c5_abl_pmc.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cp $R/iresearch_amd/csrc/libirs_hip.so /tmp/libirs_hip_base.so
for V in ${VARIANTS:-base}; do
  if [ $V = base ]; then cp /tmp/libirs_hip_base.so $R/iresearch_amd/csrc/libirs_hip.so
  else cp $R/gpurun_variants/libirs_hip_$V.so $R/iresearch_amd/csrc/libirs_hip.so; fi
  touch $R/iresearch_amd/csrc/libirs_hip.so
  rm -rf /tmp/pmc_$V
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d /tmp/pmc_$V -o p --output-format csv -- \
    python $R/bench.py --config 5 --steps 2 --warmup 2 --no-cpu > $O/${TAG}_$V.log 2>&1
  python - $V /tmp/pmc_$V >> $O/${TAG}.txt <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1], sys.argv[2]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void irs_hip::', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k in sorted(agg):
        if 'phrase' in k or 'k_conj<' in k:
            print(v, k, 'launches', len(disp[k]), {c: round(x) for c, x in agg[k].items()})
PY
done
cp /tmp/libirs_hip_base.so $R/iresearch_amd/csrc/libirs_hip.so
cat $O/${TAG}.txt
