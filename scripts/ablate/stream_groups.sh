#!/bin/bash
# Which resource does the streamed resampler kernel's time belong to?  One part compiled out per build
# (AF_STREAM_ABL, banded_mfma.hip), 48 -> 44.1 kHz on 1024 x 30 s, all builds in one call on one box:
#   bash scripts/ablate/build.sh banded_mfma sx1 "-DAF_STREAM_ABL=1" ...   then   bash scripts/ablate/stream_groups.sh TAG NAME...
set -e
cd "$GRAFT_REPO_ROOT"
TAG="$1"; shift
O="gpurun_out/r03/$TAG"; mkdir -p "$O"
export AF_BENCH_ONLY="${AF_BENCH_ONLY:-48000_44100}"
python3 scripts/gpu_bench_resample_ratios.py > "$O/product.json" 2> "$O/product.err"
for n in "$@"; do
  AF_BENCH_LIB="build/ablate/libaf_$n.so" python3 scripts/gpu_bench_resample_ratios.py > "$O/$n.json" 2> "$O/$n.err" || true
  bash scripts/ablate/check_fault.sh "$O/$n.err"
done
python3 scripts/gpu_bench_resample_ratios.py > "$O/product2.json" 2>> "$O/product.err"
for n in product "$@" product2; do printf "%-12s " "$n"; grep '"ms"' "$O/$n.json" || echo missing; done | tee "$O/summary.txt"
