#!/bin/bash
# PMC passes for the attention kernel alone (each its own rocprofv3 run, kernel-trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_attn; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
python scripts/prof_attn.py
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/scripts/prof_attn.py > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
cd $ROOT && python scripts/pmc_table.py gpurun_out/pmc_attn 2>/dev/null | grep -A18 "tr_attention"
