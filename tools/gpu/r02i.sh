set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out
for v in base noepi noitems nonorm nohits fixedonly; do
  cp gpurun_variants/libirs_hip_$v.so iresearch_amd/csrc/libirs_hip.so
  timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu > $O/r02i_var_$v.json 2> $O/r02i_var_$v.err
  python - <<PY
import json
d=json.load(open("$O/r02i_var_$v.json"))
print("$v", d["ms_per_step"], d["roofline"].get("kernel_ms"))
PY
done
cp gpurun_variants/libirs_hip_base.so iresearch_amd/csrc/libirs_hip.so
( timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --touched 2>&1 | tail -4
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --touched 2>&1 | tail -4 ) > $O/r02i_sweep.txt 2>&1
cat $O/r02i_sweep.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "FETCH_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/$O/r02i_pmc_and3_$i -o p --output-format csv -- \
    python $R/tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --steps 2 > $R/$O/r02i_pmc_and3_$i.log 2>&1
done
ls $R/$O/r02i_pmc_and3_1
