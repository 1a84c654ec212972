cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in cnofinal cnorow; do
  echo "== $v"
  timeout 300 python tools/sweep.py --op and --terms 3 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
  timeout 300 python tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --lib gpurun_variants/libirs_hip_$v.so 2>&1 | grep "step"
done > gpurun_out/r02k.txt 2>&1
cat gpurun_out/r02k.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/$O/r02k_pmc_and2_$i -o p --output-format csv -- \
    python $R/tools/sweep.py --op and --terms 2 --configs 8192:64 --nocheck --steps 2 > $R/$O/r02k_pmc_and2_$i.log 2>&1
done
