#!/bin/bash
# Host-side AddressSanitizer run (CPU container, no GPU): the C-ABI library's HOST code with -fsanitize=address (device
# code untouched: -fno-gpu-sanitize; GPU ASAN is not available on this pool), then the host-only tests against it.
#   bash scripts/asan_host.sh        (leaves the product library as it was)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/audio-flow-rs_amd/csrc"
OUT=/tmp/af_asan; mkdir -p "$OUT"
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
LIB="$ROOT/audio-flow-rs_amd/libaudioflow_dsp.so"
cp "$LIB" "$OUT/product.so"
trap 'cp "$OUT/product.so" "$LIB"' EXIT
cp "$OUT/libaudioflow_dsp.so" "$LIB"
cd "$ROOT"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:protect_shadow_gap=0 \
    python3 -m pytest tests/test_host_mirror.py tests/test_abi.py tests/test_oracle_reference_kat.py -x -q
