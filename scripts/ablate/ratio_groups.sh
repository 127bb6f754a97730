#!/bin/bash
# All ratios of scripts/gpu_bench_resample_ratios.py through the product library and the given ablation builds, one call:
#   bash scripts/ablate/ratio_groups.sh TAG NAME...
set -e
cd "$GRAFT_REPO_ROOT"
TAG="$1"; shift
O="gpurun_out/r03/$TAG"; mkdir -p "$O"
export AF_BENCH_ONLY=
python3 scripts/gpu_bench_resample_ratios.py > "$O/product.json" 2> "$O/product.err"
for n in "$@"; do
  AF_BENCH_LIB="build/ablate/libaf_$n.so" python3 scripts/gpu_bench_resample_ratios.py > "$O/$n.json" 2> "$O/$n.err" || true
  bash scripts/ablate/check_fault.sh "$O/$n.err"
done
python3 scripts/gpu_bench_resample_ratios.py > "$O/product2.json" 2>> "$O/product.err"
for n in product "$@" product2; do printf "%-10s " "$n"; python3 -c "
import json,sys
d=json.load(open('$O/$n.json'))
print(' '.join(f\"{k.replace('resample_','')}={v['ms']:.3f}\" for k,v in d.items()))"; done | tee "$O/summary.txt"
