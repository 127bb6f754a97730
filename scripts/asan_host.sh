#!/bin/bash
# Host-side AddressSanitizer run (CPU container, no GPU): the C-ABI library's HOST code with -fsanitize=address (device
# code untouched: -fno-gpu-sanitize; GPU ASAN is not available on this pool), then the host-only tests against it.
#   bash scripts/asan_host.sh        (the product library is not touched: the tests load the sanitizer build via AF_LIB)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/audio-flow-rs_amd/csrc"
OUT="$ROOT/build/asan"; mkdir -p "$OUT"
make -C "$CSRC" -j8 > /dev/null
HOSTY="core stream vad resample fir features_api chain"
for f in $HOSTY; do
  /opt/rocm/bin/hipcc -O1 -g -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize \
      -c "$CSRC/$f.hip" -o "$OUT/$f.o" &
done
wait
REST=""; for o in "$CSRC"/*.o; do b=$(basename "$o" .o); case " $HOSTY " in *" $b "*) ;; *) REST="$REST $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libsan \
    -o "$OUT/libaudioflow_dsp.so" "$OUT"/*.o $REST -Wl,-rpath,/opt/rocm/lib
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
cd "$ROOT"
# the tests load the sanitizer build through AF_LIB (audio-flow-rs_amd/_ffi.py); the product library is never touched
AF_LIB="$OUT/libaudioflow_dsp.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0 \
    python3 -m pytest tests/test_host_mirror.py tests/test_abi.py tests/test_oracle_reference_kat.py -x -q
