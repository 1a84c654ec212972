#!/bin/bash
# Dev utility (run on the GPU box via gpurun): PMC passes over tools/sweep.py, one
# counter group per pass, never combined with tracing.  Output: gpurun_out/pmcx/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIBARG=${1:+--lib $R/$1}
mkdir -p $R/gpurun_out/pmcx
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $R/gpurun_out/pmcx/g$i -o g$i --output-format csv -- python $R/tools/sweep.py $LIBARG --configs 8192:16 --steps 2 > $R/gpurun_out/pmcx/log$i.txt 2>&1
  echo "group $i rc=$?"
done
