#!/bin/bash
# scripts/prof_pmc.sh TAG KERNEL_REGEX CMD...   (on the GPU box): kernel trace + two SQ counter passes of CMD,
# condensed into gpurun_out/$AF_ROUND/TAG/{kernel_stats,pmc_sq,pmc_sq2}.txt (AF_ROUND defaults to r04)
set -eu
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG="$1"; RX="$2"; shift 2
O="$GRAFT_REPO_ROOT/gpurun_out/${AF_ROUND:-r04}/$TAG"; rm -rf "$O"; mkdir -p "$O"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- "$@" > "$O/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex "$RX" --output-format csv -d "$O/sq" -- "$@" > "$O/sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-include-regex "$RX" --output-format csv -d "$O/sq2" -- "$@" > "$O/sq2.log" 2>&1
bash scripts/ablate/check_fault.sh "$O/kt.log" "$O/sq.log" "$O/sq2.log"
python3 scripts/summarize_prof.py kernel_stats "$O/kt" "$O/kernel_stats.txt" "$RX" > /dev/null
python3 scripts/summarize_prof.py pmc "$O/sq" "$O/pmc_sq.txt" "$RX" > /dev/null
python3 scripts/summarize_prof.py pmc "$O/sq2" "$O/pmc_sq2.txt" "$RX" > /dev/null
cat "$O/kernel_stats.txt" "$O/pmc_sq.txt" "$O/pmc_sq2.txt"
