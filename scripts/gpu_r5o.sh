#!/bin/bash
# round 5, visit o: PMC pipe counters of the direct gather (product library) and of the LDS-window form on the same fp16 tiles
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
sed -i 's/default=30)/default=2)/' scripts/prof_gather_direct.py
bash scripts/gpu_pmc.sh scripts/prof_gather_direct.py direct "gl_" > /dev/null 2>&1
cp gpurun_out/pmc_direct/table.txt gpurun_out/pmc_direct_table.txt
MVS_HIP_LIB=$PWD/mvsformerplusplus_amd/csrc/libmvs_hip_direct0.so bash scripts/gpu_pmc.sh scripts/prof_gather_direct.py lds "gl_" > /dev/null 2>&1
cp gpurun_out/pmc_lds/table.txt gpurun_out/pmc_lds_table.txt
rm -rf gpurun_out/pmc_direct gpurun_out/pmc_lds
wc -l gpurun_out/pmc_direct_table.txt gpurun_out/pmc_lds_table.txt
