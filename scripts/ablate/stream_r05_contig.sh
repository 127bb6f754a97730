set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/stream2; mkdir -p $O
for ratio in 48000_44100 16000_44100 48000_22050 44100_48000; do
  export AF_BENCH_ONLY=$ratio
  printf "== %s\n" $ratio
  for n in product contig0 p16 p8 product; do
    lib=""; [ $n = product ] || lib="build/ablate/libaf_$n.so"
    printf "%-10s " $n; AF_BENCH_LIB=$lib python3 scripts/gpu_bench_resample_ratios.py 2>> $O/err.txt | grep '"ms"' || echo missing
  done
done | tee $O/times.txt
for ratio in "48000 44100" "16000 44100"; do
  tag=$(echo $ratio | tr ' ' '_')
  for n in product contig0 p16; do
    lib=""; [ $n = product ] || lib="build/ablate/libaf_$n.so"
    for c in FETCH_SIZE WRITE_SIZE; do
      d="$O/pmc_${tag}_${n}_$c"
      AF_LIB=$lib timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex banded_stream --output-format csv -d "$d" -- python3 scripts/gpu_run_resample.py $ratio 1024 30 3 > "$d.log" 2>&1 || true
      python3 scripts/summarize_prof.py pmc "$d" "$d.txt" banded_stream > /dev/null 2>&1 || true
      printf "%s %-8s %s " $tag $n $c; grep "banded_stream" "$d.txt" | awk -F, '{print $(NF-2)}'
    done
  done
done | tee $O/pmc.txt
