#!/bin/bash
# Round 5, visit 5: ablation of the COARSE stages' U-Net launches (split-bf16 format, stage-1 / stage-2 shapes): scripts/conv_ablate.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
ABL_SET=coarse timeout 600 python scripts/conv_ablate.py run 2>&1 | grep -v amdgpu.ids
