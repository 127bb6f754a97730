set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O="$GRAFT_REPO_ROOT/gpurun_out/r05/m4pmc"; rm -rf "$O"; mkdir -p "$O"
for v in arranged first_fit; do
  if [ $v = first_fit ]; then export AF_M4_NO_ARRANGE=1; else unset AF_M4_NO_ARRANGE; fi
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-include-regex af_features --output-format csv -d "$O/$v" -- python3 scripts/gpu_run_features.py 256 3 > "$O/$v.log" 2>&1
  python3 scripts/summarize_prof.py pmc "$O/$v" "$O/pmc_$v.txt" af_features > /dev/null
  echo $v; sed 's/.*>,//' "$O/pmc_$v.txt" | tail -3
done
