#!/bin/bash
# Wire path (af_wire_*, wire_kernel) profile: kernel trace + stats, then separate PMC passes (FETCH_SIZE / WRITE_SIZE /
# SQ counters) of the fused launch on 1024 x 30 s stereo 48 kHz f32.  Run on the GPU box through gpurun; output under
# gpurun_out/r05p; condensed copies go to profiles/r05_wire_*.
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r05p"
mkdir -p "$O"
W="python3 scripts/gpu_bench_wire.py --only fused --reps 20"
W3="python3 scripts/gpu_bench_wire.py --only fused --reps 3 --warm_s 0"
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
pmc() { # dir regex counters -- cmd
  local d="$1" rx="$2" c="$3"; shift 3
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-include-regex "$rx" --output-format csv -d "$O/$d" -- "$@" > "$O/$d.log" 2>&1
  bash scripts/ablate/check_fault.sh "$O/$d.log"
  python3 scripts/summarize_prof.py pmc "$O/$d" "$O/pmc_$d.txt" "$rx" > /dev/null
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/wire_kt" -- $W > "$O/wire_kt.log" 2>&1
python3 scripts/summarize_prof.py kernel_stats "$O/wire_kt" "$O/wire_kernel_stats.txt" "wire|af_|af::" > /dev/null
pmc wire_fetch wire_kernel "FETCH_SIZE" $W3
pmc wire_write wire_kernel "WRITE_SIZE" $W3
pmc wire_sq wire_kernel "$SQ1" $W3
pmc wire_sq2 wire_kernel "$SQ2" $W3
pmc cal_fetch to_mono "FETCH_SIZE" python3 scripts/pmc_calibrate.py
pmc cal_write to_mono "WRITE_SIZE" python3 scripts/pmc_calibrate.py
python3 scripts/gpu_bench_wire.py --json "$O/wire_bench.json" > /dev/null
python3 scripts/gpu_bench_wire.py --frame 320 --json "$O/wire_bench_320.json" > /dev/null
python3 scripts/gpu_bench_wire.py --fmt i16 --json "$O/wire_bench_i16.json" > /dev/null
cat "$O/wire_kernel_stats.txt" "$O"/pmc_wire_*.txt "$O"/pmc_cal_*.txt
