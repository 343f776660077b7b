#!/bin/bash
# PMC counter passes (each its own rocprofv3 run, kernel-trace only) for a driver script: gpu_pmc.sh <driver.py> <tag>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_${2:-x}; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TA_TA_BUSY TD_TD_BUSY GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/$1 > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
python $ROOT/scripts/pmc_table.py $OUT "${3:-.}" > $OUT/table.txt 2>&1
