#!/bin/bash
# A/B library variants: tools/build_variant.sh NAME "EXTRA_FLAGS" file1.hip [file2.hip ...]
# recompiles the listed sources of sensorium_amd/csrc with EXTRA_FLAGS into build_ab/NAME/ and links them with the product build's
# other objects -> build_ab/NAME/libdwiseneuro_hip.so (same ABI; load it with DWN_LIB_PATH for same-box A/B runs).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2
CS=$R/sensorium_amd/csrc
make -C "$CS" -j8 >/dev/null
OUT=$R/build_ab/$NAME
mkdir -p "$OUT"
OBJS=""
for o in "$CS"/build/*.o; do
  b=$(basename "$o" .o)
  skip=0
  for f in "$@"; do [ "$b" = "$(basename "$f" .hip)" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for f in "$@"; do
  b=$(basename "$f" .hip)
  hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wno-unused-result $EXTRA -I"$CS" -c "$CS/$f" -o "$OUT/$b.o" &
done
wait
for f in "$@"; do OBJS="$OBJS $OUT/$(basename "$f" .hip).o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libdwiseneuro_hip.so"
echo "built $OUT/libdwiseneuro_hip.so"
