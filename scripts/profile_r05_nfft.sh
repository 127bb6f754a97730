#!/bin/bash
# The matrix-product DFT kernel's part of scripts/profile_r05.sh alone (run on the GPU box through gpurun): kernel trace, FETCH / WRITE /
# SQ counter passes (never combined with a trace), the bench lines with the in-kernel stamps.  Output: gpurun_out/r05p/nfft_*.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r05p"; mkdir -p "$O"
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
pmc() { local d="$1" rx="$2" c="$3"; shift 3
  rm -rf "$O/$d"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "$rx" --output-format csv -d "$O/$d" -- "$@" > "$O/$d.log" 2>&1
  bash scripts/ablate/check_fault.sh "$O/$d.log"
  python3 scripts/summarize_prof.py pmc "$O/$d" "$O/pmc_$d.txt" "$rx" > /dev/null; }
rm -rf "$O/nfft_kt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/nfft_kt" -- python3 scripts/gpu_bench_nfft.py --reps 20 > "$O/nfft_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/nfft_kt" "$O/nfft_kernel_stats.txt" "af_|af::" > /dev/null
pmc nfft_fetch af_features_dft "FETCH_SIZE" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_write af_features_dft "WRITE_SIZE" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_sq af_features_dft "$SQ1" python3 scripts/gpu_bench_nfft.py --reps 3
pmc nfft_sq2 af_features_dft "$SQ2" python3 scripts/gpu_bench_nfft.py --reps 3
python3 scripts/gpu_bench_nfft.py --reps 30 --stamps --json "$O/nfft400.json"
python3 scripts/gpu_bench_nfft.py --reps 30 --n_fft 480 --hop 160 --json "$O/nfft480.json"
cat "$O/nfft_kernel_stats.txt" | head -6
