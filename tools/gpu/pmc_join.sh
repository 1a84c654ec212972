#!/bin/bash
# PMC passes over tools/join_tune.py (one counter group per pass, never combined with tracing).
# usage: bash tools/gpu/pmc_join.sh TAG RUNS   ->  gpurun_out/TAG_pmc/gN + TAG_pmc.txt
TAG=${1:-r03p}; RUNS=${2:-base:1024}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_pmc; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp -d $O/g$i -o g$i --output-format csv -- \
    python $R/tools/join_tune.py --runs $RUNS --steps 2 > $O/log$i.txt 2>&1
  echo "group $i rc=$?"
done
python $R/tools/pmc_table.py $O > $R/gpurun_out/${TAG}_pmc.txt 2>&1
cat $R/gpurun_out/${TAG}_pmc.txt
