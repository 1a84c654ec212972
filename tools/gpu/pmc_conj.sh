#!/bin/bash
# PMC passes over a conjunction sweep (one counter group per pass, never combined with tracing).
# usage: bash tools/gpu/pmc_conj.sh TAG "sweep args"   ->  gpurun_out/TAG_pmc_conj/gN + TAG_pmc_conj.txt
TAG=${1:-r04h}; ARGS=${2:---op and --terms 2 --path items}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_pmc_conj; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
           "SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp -d $O/g$i -o g$i --output-format csv -- \
    python $R/tools/sweep.py $ARGS --configs 8192:64 --nocheck --steps 2 > $O/log$i.txt 2>&1
  echo "group $i rc=$?"
done
python $R/tools/pmc_table.py $O > $R/gpurun_out/${TAG}_pmc_conj.txt 2>&1
cat $R/gpurun_out/${TAG}_pmc_conj.txt
