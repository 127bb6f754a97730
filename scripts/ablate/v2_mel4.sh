#!/bin/bash
# Ablation builds of the MEL4 instances of af_features_v2_kernel (mel phase on 4 x 4 x 1 blocks), round 5:
#   bash scripts/ablate/v2_mel4.sh            (here: cross-compiles build/ablate/libaf_m4_*.so)
#   bash scripts/ablate/v2_mel4.sh run [TAG]  (on the GPU box through gpurun -> gpurun_out/r05/TAG/time.txt)
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = run ]; then
  : "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
  O="gpurun_out/r05/${2:-m4}"; mkdir -p "$O"
  L=build/ablate; LIBS="$L/libaf_m4_base.so"
  for n in nomfma nost nodpp nomel nomfma_nost lane0 base; do LIBS="$LIBS $L/libaf_m4_$n.so"; done
  timeout -k 10 300 python3 scripts/ablate/run_features.py --clk --warm 60 --reps 40 $LIBS > "$O/time.txt" 2> "$O/time.err" || { tail -5 "$O/time.err"; exit 4; }
  timeout -k 10 100 python3 scripts/ablate/run_features.py --clk --mel4 0 --warm 60 --reps 40 $L/libaf_m4_base.so $L/libaf_m4_lane0.so $L/libaf_m4_base.so | sed 's/^/mel4=0 /' >> "$O/time.txt" 2>> "$O/time.err"
  cat "$O/time.txt"
  exit 0
fi
bash scripts/ablate/build.sh features_v2 \
  m4_base   "-DAF_V2_CLK=1" \
  m4_nomfma "-DAF_V2_CLK=1 -DAF_V2_M4=1" \
  m4_nost   "-DAF_V2_CLK=1 -DAF_V2_M4=2" \
  m4_nodpp  "-DAF_V2_CLK=1 -DAF_V2_M4=4" \
  m4_nomel  "-DAF_V2_CLK=1 -DAF_V2_M4=7" \
  m4_nomfma_nost "-DAF_V2_CLK=1 -DAF_V2_M4=3" \
  m4_lane0  "-DAF_V2_CLK=1 -DAF_V2_LANE_NOW=0"
