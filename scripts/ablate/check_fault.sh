#!/bin/bash
# scripts/ablate/check_fault.sh LOG [LOG ...]: exit 3 if the HIP runtime reported a GPU fault in any of the logs
# (a faulting process can still leave a pipeline's exit status at 0: gpurun_out/r02/call15.log).
rc=0
for f in "$@"; do
  [ -f "$f" ] || continue
  if grep -qiE "memory access fault|HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION|hardware exception" "$f"; then
    echo "GPU fault reported in $f:" >&2
    grep -iE "memory access fault|HSA_STATUS_ERROR|hardware exception" "$f" | head -3 >&2
    rc=3
  fi
done
exit $rc
