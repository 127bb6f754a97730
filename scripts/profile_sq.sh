#!/bin/bash
# SQ / LDS counter passes of the fused feature kernel at config 2 (run on the GPU box through gpurun):
#   scripts/profile_sq.sh <tag>    -> gpurun_out/<tag>/{sq,sq2,sq3}.txt
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-sq}
O="$GRAFT_REPO_ROOT/gpurun_out/$T"
mkdir -p "$O"
R="python3 scripts/gpu_run_features.py 256 3"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex af_features --output-format csv -d "$O/sq" -- $R > "$O/sq.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex af_features --output-format csv -d "$O/sq2" -- $R > "$O/sq2.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT SQ_WAVES --kernel-include-regex af_features --output-format csv -d "$O/sq3" -- $R > "$O/sq3.log" 2>&1
for s in sq sq2 sq3; do python3 scripts/summarize_prof.py pmc "$O/$s" "$O/$s.txt" af_features > /dev/null; done
cat "$O/sq.txt" "$O/sq2.txt" "$O/sq3.txt" | grep -v "^#" | cut -d, -f2,4
