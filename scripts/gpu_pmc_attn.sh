#!/bin/bash
# Attention A/B + ablations, then PMC passes for ONE fp16 variant (each pass its own rocprofv3 run, kernel-trace only).
#   gpurun -- 'bash scripts/gpu_pmc_attn.sh [variant]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_attn; mkdir -p $OUT; V=${1:-0}
export PYTHONDONTWRITEBYTECODE=1
python scripts/prof_attn.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r04_attn_ab.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/scripts/prof_attn.py --only $V > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
cd $ROOT && python scripts/pmc_table.py gpurun_out/pmc_attn "tr_attention16" > gpurun_out/pmc_attn/table.txt 2>&1
cat gpurun_out/pmc_attn/table.txt | cut -c1-160
python scripts/pmc_table_summary.py gpurun_out/pmc_attn/table.txt
rm -rf gpurun_out/pmc_attn/p[0-9]*
