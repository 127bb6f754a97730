#!/bin/bash
# LDS cycles / bank conflicts of the hop-256 kernel by access group (run on the GPU box after scripts/ablate/build.sh
# built libaf_abl{0,1,2,...,1024}.so): one timing pass, one PMC pass.
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r03/abl"; mkdir -p "$O"
M="0 1 2 4 8 16 32 64 128 256 512 1024"
LIBS=""; NAMES=""
for m in $M; do LIBS="$LIBS build/ablate/libaf_abl$m.so"; NAMES="$NAMES abl$m"; done
python3 scripts/ablate/run_features.py $LIBS > "$O/time.txt" 2> "$O/time.err" || { cat "$O/time.err" | tail -5; bash scripts/ablate/check_fault.sh "$O/time.err"; exit 4; }
bash scripts/ablate/check_fault.sh "$O/time.err" "$O/time.txt"
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE \
   --kernel-include-regex af_features --output-format csv -d "$O/pmc" -- python3 scripts/ablate/run_features.py --warm 0 --reps 3 $LIBS > "$O/pmc.log" 2>&1
bash scripts/ablate/check_fault.sh "$O/pmc.log"
python3 scripts/ablate/pmc_table.py "$O/pmc" af_features 3 $NAMES > "$O/pmc_table.csv"
cat "$O/time.txt" "$O/pmc_table.csv"
