#!/bin/bash
# PMC counter passes over the whole bench step (each pass its own rocprofv3 run, kernel-trace only): per-kernel pipe
# utilisation of the MFMA convolutions and everything else.   gpurun -- 'bash scripts/gpu_pmc_bench.sh <tag> [bench args]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); TAG=${1:-conv}; shift || true
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY TD_TD_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile "$@" > $OUT/p$i.log 2>&1
  tail -1 $OUT/p$i.log
done
cd $ROOT && python scripts/pmc_table.py gpurun_out/pmc_$TAG > gpurun_out/pmc_$TAG/table.txt 2>&1; wc -l gpurun_out/pmc_$TAG/table.txt
python scripts/pmc_table_summary.py gpurun_out/pmc_$TAG/table.txt > gpurun_out/pmc_$TAG/summary.txt 2>&1; head -60 gpurun_out/pmc_$TAG/summary.txt
rm -rf gpurun_out/pmc_$TAG/p[0-9]*          # the raw per-dispatch CSVs exceed what gpurun copies back
