#!/bin/bash
# A/B build of ONE translation unit with extra compiler flags: links it with the product's other objects into
# end-to-end-slu_amd/lib_alt/libslu_hip.so (select with SLU_HIP_LIB=<that path>; never loaded by the product).
# usage: tools/build_alt.sh <unit> <flags...>      e.g. tools/build_alt.sh slu_gru_bf16 -fno-slp-vectorize
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
UNIT=$1; shift
SRC="$ROOT/end-to-end-slu_amd/csrc"; LIB="$ROOT/end-to-end-slu_amd/lib"; ALT="${ALT_DIR:-$ROOT/end-to-end-slu_amd/lib_alt}"
mkdir -p "$ALT"
bash "$SRC/build.sh" > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$SRC/$UNIT.hip" -o "$ALT/$UNIT.o"
OBJS=()
for o in "$LIB"/*.o; do
  b=$(basename "$o")
  if [ "$b" = "$UNIT.o" ]; then OBJS+=("$ALT/$UNIT.o"); else OBJS+=("$o"); fi
done
# EXTRA_UNITS="path/to/a.hip ...": further translation units linked into the alt library only (A/B baselines)
for x in ${EXTRA_UNITS:-}; do
  b=$(basename "$x" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I"$SRC" -c "$x" -o "$ALT/$b.o"
  OBJS+=("$ALT/$b.o")
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -o "$ALT/libslu_hip.so"
echo "[build_alt] $ALT/libslu_hip.so ($UNIT with $*)"
