#!/bin/bash
# Ablation builds of the library (diagnostics, never shipped): scripts/ablate/build.sh FILE[,FILE2] NAME "FLAGS" ...
#   compiles audio-flow-rs_amd/csrc/FILE.hip (and FILE2.hip) with FLAGS (e.g. "-DAF_ABL=8") and links it with the product's other
#   objects into build/ablate/libaf_NAME.so (build/ is git-ignored but travels with gpurun).  Load with AF_LIB=...
# Keeping the variants as macros in the product source + this recipe means a fault in an ablation run can be traced
# to the exact code that ran ("What's weak" 7).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CSRC="$ROOT/audio-flow-rs_amd/csrc"
OUT="$ROOT/build/ablate"
mkdir -p "$OUT"
make -C "$CSRC" -j8 > /dev/null
FILES="${1//,/ }"; shift
while [ $# -ge 2 ]; do
  NAME="$1"; FLAGS="$2"; shift 2
  for FILE in $FILES; do
    /opt/rocm/bin/hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
        $FLAGS -c "$CSRC/$FILE.hip" -o "$OUT/${FILE}_$NAME.o" &
    # at most 8 compilers at a time
    while [ "$(jobs -r | wc -l)" -ge 8 ]; do sleep 0.2; done
  done
  NAMES="$NAMES $NAME"
done
wait
for NAME in $NAMES; do
  OBJS=""
  for o in "$CSRC"/*.o; do
    skip=0; for FILE in $FILES; do [ "$(basename "$o")" = "$FILE.o" ] && skip=1; done
    [ $skip = 1 ] || OBJS="$OBJS $o"
  done
  for FILE in $FILES; do OBJS="$OBJS $OUT/${FILE}_$NAME.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libaf_$NAME.so" $OBJS -Wl,-rpath,/opt/rocm/lib
  echo "built $OUT/libaf_$NAME.so"
done
