cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcx
rocprofv3 --list-avail > $R/gpurun_out/pmcx/avail.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_IFETCH" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/gpurun_out/pmcx/g$i -o g$i --output-format csv -- python $R/tools/sweep.py --lib $R/gpurun_variants/lib_old.so --configs 8192:16 --steps 2 > $R/gpurun_out/pmcx/log$i.txt 2>&1
  echo "group $i rc=$?"
done
