#!/bin/bash
# On the GPU box (through gpurun): steady-clock phase table of af_features_v2_kernel, BASELINE config 2.
#   bash scripts/ablate/v2_phases_run.sh [TAG]     -> gpurun_out/r05/TAG/{time.txt,valu_rate.txt}
# The builds come from scripts/ablate/v2_phases.sh (cross-compiled in the container, they travel with the snapshot).
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT"
TAG="${1:-v2_phases}"
O="gpurun_out/r05/$TAG"; mkdir -p "$O"
L=build/ablate
LIBS="$L/libaf_v2p_base.so"
for n in skel skel_nb skel_nm skel_nmnb skel_nopf nofft1 nofft2 nomfma nomel nobar nost nopf nolds nolds_nb arith arith_nb arith1 arith2 nopfld nofrag nomelrd notile base; do
  LIBS="$LIBS $L/libaf_v2p_$n.so"
done
timeout -k 10 300 python3 scripts/ablate/run_features.py --warm 60 --reps 40 $LIBS > "$O/time.txt" 2> "$O/time.err" || { tail -5 "$O/time.err"; bash scripts/ablate/check_fault.sh "$O/time.err"; exit 4; }
bash scripts/ablate/check_fault.sh "$O/time.err" "$O/time.txt"
cat "$O/time.txt"
if [ -x scripts/ubench/valu_rate ]; then timeout -k 10 120 scripts/ubench/valu_rate > "$O/valu_rate.txt" 2>&1; cat "$O/valu_rate.txt"; fi
