#!/bin/bash
# Round 5, banded_stream_kernel: where the 1.28 x traffic of 48 -> 44.1 kHz comes from, and the two sides of 16 -> 44.1 kHz.
#   here:        bash scripts/ablate/stream_r05.sh build      (cross-compiles build/ablate/libaf_sx*.so)
#   on the box:  bash scripts/ablate/stream_r05.sh run        -> gpurun_out/r05/stream/
set -e
cd "$(dirname "$0")/../.."
NAMES="sx1 sx2 sx3 sx4 sx1024 sx64 sx65 sx66 sx8"
if [ "$1" = build ]; then
  ARGS=""
  for n in $NAMES; do ARGS="$ARGS $n -DAF_STREAM_ABL=${n#sx}"; done
  bash scripts/ablate/build.sh banded_mfma $ARGS | grep -c built
  exit 0
fi
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="gpurun_out/r05/stream"; mkdir -p "$O"
for ratio in 48000_44100 16000_44100; do
  export AF_BENCH_ONLY=$ratio
  { printf "== %s\n" $ratio
    printf "%-10s " product; python3 scripts/gpu_bench_resample_ratios.py 2>> "$O/err.txt" | grep '"ms"'
    for n in $NAMES; do
      printf "%-10s " $n; AF_BENCH_LIB="build/ablate/libaf_$n.so" python3 scripts/gpu_bench_resample_ratios.py 2>> "$O/err.txt" | grep '"ms"' || echo missing
    done
    printf "%-10s " product2; python3 scripts/gpu_bench_resample_ratios.py 2>> "$O/err.txt" | grep '"ms"'
  } | tee -a "$O/times.txt"
done
bash scripts/ablate/check_fault.sh "$O/err.txt"
# HBM bytes per build (separate passes; FETCH_SIZE in units of 2 KiB, WRITE_SIZE of 1 KiB on gfx950: profiles/r05_pmc_cal_*)
for ratio in "48000 44100" "16000 44100"; do
  tag=$(echo $ratio | tr ' ' '_')
  for n in product sx4 sx1024 sx1 sx2; do
    lib=""; [ $n = product ] || lib="build/ablate/libaf_$n.so"
    for c in FETCH_SIZE WRITE_SIZE; do
      d="$O/pmc_${tag}_${n}_$c"
      AF_LIB=$lib timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex banded --output-format csv -d "$d" -- python3 scripts/gpu_run_resample.py $ratio 1024 30 3 > "$d.log" 2>&1 || true
      bash scripts/ablate/check_fault.sh "$d.log"
      python3 scripts/summarize_prof.py pmc "$d" "$d.txt" banded > /dev/null 2>&1 || true
      printf "%s %-8s %s " $tag $n $c; grep -v "^#\|^kernel," "$d.txt" | head -1 | awk -F, '{print $(NF-2)}'
    done
  done
done | tee "$O/pmc.txt"
