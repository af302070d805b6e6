#!/bin/bash
# Probe builds of the split-precision convolution kernel (tools/probes/make_wconv_probe.py): one alt library per probe
# value, end-to-end-slu_amd/lib_alt/libslu_hip_wprobe<bits>.so (select with SLU_HIP_LIB; never loaded by the product).
# usage: tools/build_wconv_probe.sh <bits...>
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
SRC="$ROOT/end-to-end-slu_amd/csrc"; LIB="$ROOT/end-to-end-slu_amd/lib"; ALT="$ROOT/end-to-end-slu_amd/lib_alt"
mkdir -p "$ALT"
bash "$SRC/build.sh" > /dev/null
python "$ROOT/tools/probes/make_wconv_probe.py" "$ALT/slu_wconv_bf16_probe.hip"
for bits in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I"$SRC" -I"$ROOT/include" -DSLU_WPROBE=$bits \
      -c "$ALT/slu_wconv_bf16_probe.hip" -o "$ALT/slu_wconv_bf16_probe$bits.o"
  OBJS=()
  for o in "$LIB"/*.o; do
    if [ "$(basename "$o")" = "slu_wconv_bf16.o" ]; then OBJS+=("$ALT/slu_wconv_bf16_probe$bits.o"); else OBJS+=("$o"); fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -o "$ALT/libslu_hip_wprobe$bits.so"
  echo "[build_wconv_probe] $ALT/libslu_hip_wprobe$bits.so"
done
