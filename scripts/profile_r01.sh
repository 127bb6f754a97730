#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun):
#   kernel trace + stats of the default bench, then separate PMC passes
#   (FETCH_SIZE / WRITE_SIZE / SQ counters) restricted to the product kernels.
set -e
: "${GRAFT_REPO_ROOT:?run this on the GPU box through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r01"
rm -rf "$O" && mkdir -p "$O"
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline"
Bs="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- $B > "$O/kt.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex af_features --output-format csv -d "$O/fetch" -- $Bs > "$O/fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex af_features --output-format csv -d "$O/write" -- $Bs > "$O/write.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --kernel-include-regex af_features --output-format csv -d "$O/sq" -- $Bs > "$O/sq.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-include-regex af_features --output-format csv -d "$O/sq2" -- $Bs > "$O/sq2.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex to_mono --output-format csv -d "$O/cal_fetch" -- python3 scripts/pmc_calibrate.py > "$O/cal_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex to_mono --output-format csv -d "$O/cal_write" -- python3 scripts/pmc_calibrate.py > "$O/cal_write.log" 2>&1
ls "$O"
# secondary kernels (banded MFMA resampler / FIR, streaming ops, power mode, mel_apply)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ops_kt" -- python3 scripts/gpu_bench_ops.py > "$O/ops.json" 2> "$O/ops_kt.log"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex banded_kernel --output-format csv -d "$O/banded_fetch" -- python3 scripts/gpu_bench_banded.py r441 > "$O/banded_fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex banded_kernel --output-format csv -d "$O/banded_write" -- python3 scripts/gpu_bench_banded.py r441 > "$O/banded_write.log" 2>&1
ls "$O"
