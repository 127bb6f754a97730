#!/bin/bash
# Config 4's chain under rocprofv3 (run on the GPU box through gpurun):
#   scripts/profile_chain.sh <tag>  -> gpurun_out/<tag>/{kernel_stats,pmc_*}.txt
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T=${1:-chain}
O="$GRAFT_REPO_ROOT/gpurun_out/$T"
mkdir -p "$O"
R="python3 scripts/gpu_run_chain.py 256 40"
R3="python3 scripts/gpu_run_chain.py 256 3"
K="resfir|af_features"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- $R > "$O/kt.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d "$O/fetch" -- $R3 > "$O/fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$K" --output-format csv -d "$O/write" -- $R3 > "$O/write.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex resfir --output-format csv -d "$O/sq" -- $R3 > "$O/sq.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex resfir --output-format csv -d "$O/sq2" -- $R3 > "$O/sq2.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/kt" "$O/kernel_stats.txt" "af_|af::|resfir|banded" > /dev/null
for s in fetch write; do python3 scripts/summarize_prof.py pmc "$O/$s" "$O/pmc_$s.txt" "$K" > /dev/null; done
for s in sq sq2; do python3 scripts/summarize_prof.py pmc "$O/$s" "$O/pmc_$s.txt" resfir > /dev/null; done
cat "$O/kernel_stats.txt"; cat "$O"/pmc_*.txt | grep -v "^#" | cut -d, -f1,2,4
