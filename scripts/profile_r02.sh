#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun):
#   kernel trace + stats of the default bench, then separate PMC passes (FETCH_SIZE / WRITE_SIZE /
#   SQ counters) restricted to the fused feature kernel, the calibration kernel and the ops bench.
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r02p"
rm -rf "$O" && mkdir -p "$O"
B="python3 bench.py --no-cpu-baseline --no-secondary"          # the default steps / warmup
R="python3 scripts/gpu_run_features.py 256 3"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- $B > "$O/kt.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex af_features --output-format csv -d "$O/fetch" -- $R > "$O/fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex af_features --output-format csv -d "$O/write" -- $R > "$O/write.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex af_features --output-format csv -d "$O/sq" -- $R > "$O/sq.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex af_features --output-format csv -d "$O/sq2" -- $R > "$O/sq2.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex to_mono --output-format csv -d "$O/cal_fetch" -- python3 scripts/pmc_calibrate.py > "$O/cal_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex to_mono --output-format csv -d "$O/cal_write" -- python3 scripts/pmc_calibrate.py > "$O/cal_write.log" 2>&1
# the bench line itself, no profiler attached
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
# secondary kernels
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ops_kt" -- python3 scripts/gpu_bench_ops.py > "$O/ops.json" 2> "$O/ops_kt.log"
for s in fetch write sq sq2; do python3 scripts/summarize_prof.py pmc "$O/$s" "$O/pmc_$s.txt" af_features > /dev/null; done
for s in cal_fetch cal_write; do python3 scripts/summarize_prof.py pmc "$O/$s" "$O/pmc_$s.txt" to_mono > /dev/null; done
python3 scripts/summarize_prof.py kernel_stats "$O/kt" "$O/kernel_stats.txt" "af_|af::" > /dev/null
python3 scripts/summarize_prof.py kernel_stats "$O/ops_kt" "$O/ops_kernel_stats.txt" "af_|af::|banded|fir|resample" > /dev/null
ls "$O"
