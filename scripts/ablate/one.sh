#!/bin/bash
# Timing + LDS counters of the feature kernel for the given library builds (default: the product library):
#   bash scripts/ablate/one.sh TAG [LIB ...]
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG="$1"; shift
LIBS="$@"; [ -z "$LIBS" ] && LIBS="audio-flow-rs_amd/libaudioflow_dsp.so"
O="$GRAFT_REPO_ROOT/gpurun_out/${AF_ROUND:-r04}/$TAG"; mkdir -p "$O"
NAMES=""; for l in $LIBS; do NAMES="$NAMES $(basename $l .so)"; done
python3 scripts/ablate/run_features.py $LIBS > "$O/time.txt" 2> "$O/time.err" || { cat "$O/time.err" | tail -5; bash scripts/ablate/check_fault.sh "$O/time.err"; exit 4; }
bash scripts/ablate/check_fault.sh "$O/time.err" "$O/time.txt"
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
   --kernel-include-regex af_features --output-format csv -d "$O/pmc" -- python3 scripts/ablate/run_features.py --warm 0 --reps 3 $LIBS > "$O/pmc.log" 2>&1
bash scripts/ablate/check_fault.sh "$O/pmc.log"
python3 scripts/ablate/pmc_table.py "$O/pmc" af_features 3 $NAMES > "$O/pmc_table.csv"
cat "$O/time.txt" "$O/pmc_table.csv"
