#!/bin/bash
# Round-5 profiling recipe (run on the GPU box through gpurun; output under gpurun_out/r05p, condensed copies are
# committed under profiles/ by scripts/make_pmc_json.py):
#   kernel trace + stats of the default bench command, then separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ
#   counters) for the fused feature kernel, the calibration kernel, config 4's chain, the resampler kernels and
#   the ops bench.  Counters are never combined with a trace.
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r05p"
rm -rf "$O" && mkdir -p "$O"
B="python3 bench.py --no-cpu-baseline --no-secondary"          # the default steps / warmup
R="python3 scripts/gpu_run_features.py 256 3"
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
pmc() { # dir regex counters -- cmd
  local d="$1" rx="$2" c="$3"; shift 3
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "$rx" --output-format csv -d "$O/$d" -- "$@" > "$O/$d.log" 2>&1
  bash scripts/ablate/check_fault.sh "$O/$d.log"
  python3 scripts/summarize_prof.py pmc "$O/$d" "$O/pmc_$d.txt" "$rx" > /dev/null
}
# ---- headline kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- $B > "$O/kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/kt" "$O/kernel_stats.txt" "af_|af::" > /dev/null
pmc fetch af_features "FETCH_SIZE" $R
pmc write af_features "WRITE_SIZE" $R
pmc sq af_features "$SQ1" $R
pmc sq2 af_features "$SQ2" $R
pmc cal_fetch to_mono "FETCH_SIZE" python3 scripts/pmc_calibrate.py
pmc cal_write to_mono "WRITE_SIZE" python3 scripts/pmc_calibrate.py
echo "headline done"
# (the matrix-core DFT kernel, features_v3.hip, is frozen since round 4: its record is profiles/r04_v3_*; the bench line still
#  times it as secondary.features_v3)
# ---- the bench line itself, no profiler attached (driver's arguments and the defaults)
python3 bench.py --steps 20 --warmup 5 > "$O/bench_driver_args.json" 2> "$O/bench_driver_args.err"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
echo "bench done"
# ---- config 4 chain
C="python3 scripts/gpu_run_chain.py 256 40"; C3="python3 scripts/gpu_run_chain.py 256 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/chain_kt" -- $C > "$O/chain_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/chain_kt" "$O/chain_kernel_stats.txt" "af_|af::|resfir|banded" > /dev/null
pmc chain_fetch "resfir|af_features" "FETCH_SIZE" $C3
pmc chain_write "resfir|af_features" "WRITE_SIZE" $C3
pmc chain_sq resfir "$SQ1" $C3
pmc chain_sq2 resfir "$SQ2" $C3
echo "chain done"
# ---- resampler with streamed fragments (48 -> 44.1 k) and config 3
X="python3 scripts/gpu_run_resample.py 48000 44100 1024 30 4"
pmc res_fetch banded "FETCH_SIZE" $X
pmc res_write banded "WRITE_SIZE" $X
pmc res_sq banded "$SQ1" $X
pmc res_sq2 banded "$SQ2" $X
python3 scripts/gpu_bench_resample_ratios.py > "$O/resample_ratios.json" 2> "$O/resample_ratios.err"
echo "resampler done"
# ---- the wire path in one launch (af_wire_*, wire_kernel) on 1024 x 30 s stereo
W="python3 scripts/gpu_bench_wire.py --only fused --reps 20"
W3="python3 scripts/gpu_bench_wire.py --only fused --reps 3 --warm_s 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/wire_kt" -- $W > "$O/wire_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/wire_kt" "$O/wire_kernel_stats.txt" "wire|af_|af::" > /dev/null
pmc wire_fetch wire_kernel "FETCH_SIZE" $W3
pmc wire_write wire_kernel "WRITE_SIZE" $W3
pmc wire_sq wire_kernel "$SQ1" $W3
pmc wire_sq2 wire_kernel "$SQ2" $W3
python3 scripts/gpu_bench_wire.py --json "$O/wire_bench.json" > /dev/null
python3 scripts/gpu_bench_wire.py --frame 320 --level --json "$O/wire_bench_messages320.json" > /dev/null
python3 scripts/gpu_bench_wire.py --fmt i16 --json "$O/wire_bench_i16.json" > /dev/null
echo "wire done"
# ---- the chain at 44.1 kHz (resfir10_kernel: ten block matrices) against its stage-by-stage form
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/chain44_kt" -- python3 scripts/gpu_check_resfir10.py --time-only > "$O/chain44_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/chain44_kt" "$O/chain44_kernel_stats.txt" "af_|af::|resfir|banded" > /dev/null
python3 scripts/gpu_check_resfir10.py --time-only > "$O/chain44.txt" 2>&1
echo "chain 44.1 done"
# ---- the speech framing (n_fft 400 / hop 160: af_features_dft_kernel, the DFT as a matrix product)
N="python3 scripts/gpu_bench_nfft.py --reps 20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/nfft_kt" -- $N > "$O/nfft_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/nfft_kt" "$O/nfft_kernel_stats.txt" "af_|af::" > /dev/null
pmc nfft_fetch af_features_dft "FETCH_SIZE" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_write af_features_dft "WRITE_SIZE" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_sq af_features_dft "$SQ1" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_sq2 af_features_dft "$SQ2" python3 scripts/gpu_bench_nfft.py --reps 3
python3 scripts/gpu_bench_nfft.py --reps 30 --stamps --json "$O/nfft400.json" > /dev/null
python3 scripts/gpu_bench_nfft.py --reps 30 --n_fft 480 --hop 160 --json "$O/nfft480.json" > /dev/null
echo "nfft done"
# ---- secondary kernels
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ops_kt" -- python3 scripts/gpu_bench_ops.py > "$O/ops.json" 2> "$O/ops_kt.log"
python3 scripts/summarize_prof.py kernel_stats "$O/ops_kt" "$O/ops_kernel_stats.txt" "af_|af::|banded|fir|resample" > /dev/null
ls "$O"
