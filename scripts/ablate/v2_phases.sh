#!/bin/bash
# Phase-ablation builds of af_features_v2_kernel for the steady-clock table profiles/r05_v2_ablations.txt:
#   bash scripts/ablate/v2_phases.sh          (here: cross-compiles build/ablate/libaf_v2p_*.so)
# AF_V2_PH bits (phases), AF_ABL bits (LDS access groups): see features_v2.hip.  ALL = every LDS group (2047 minus the unused 32).
set -e
cd "$(dirname "$0")/../.."
ALL=2015
bash scripts/ablate/build.sh features_v2 \
  v2p_base     "" \
  v2p_skel     "-DAF_V2_PH=3" \
  v2p_skel_nb  "-DAF_V2_PH=11" \
  v2p_skel_nm  "-DAF_V2_PH=131" \
  v2p_skel_nmnb "-DAF_V2_PH=139" \
  v2p_skel_nopf "-DAF_V2_PH=35 -DAF_ABL=1024" \
  v2p_nofft1   "-DAF_V2_PH=1" \
  v2p_nofft2   "-DAF_V2_PH=2" \
  v2p_nomfma   "-DAF_V2_PH=4" \
  v2p_nomel    "-DAF_V2_PH=128" \
  v2p_nobar    "-DAF_V2_PH=8" \
  v2p_nost     "-DAF_V2_PH=16" \
  v2p_nopf     "-DAF_V2_PH=32 -DAF_ABL=1024" \
  v2p_nolds    "-DAF_ABL=$ALL" \
  v2p_nolds_nb "-DAF_ABL=$ALL -DAF_V2_PH=8" \
  v2p_arith    "-DAF_ABL=$ALL -DAF_V2_PH=168" \
  v2p_arith_nb "-DAF_ABL=$ALL -DAF_V2_PH=184" \
  v2p_arith1   "-DAF_ABL=$ALL -DAF_V2_PH=170" \
  v2p_arith2   "-DAF_ABL=$ALL -DAF_V2_PH=169" \
  v2p_nopfld   "-DAF_V2_PH=32" \
  v2p_nofrag   "-DAF_V2_PH=64" \
  v2p_nomelrd  "-DAF_ABL=512" \
  v2p_notile   "-DAF_ABL=1024"
