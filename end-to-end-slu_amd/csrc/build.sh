#!/bin/bash
# Builds libslu_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
# Out-of-date translation units are compiled in parallel (SLU_BUILD_JOBS, default = the host's cores, max 16).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${SLU_EXTRA_FLAGS:-}"
JOBS="${SLU_BUILD_JOBS:-$(( $(nproc) < 16 ? $(nproc) : 16 ))}"
UNITS="slu_api slu_sinc slu_wconv slu_wconv_bf16 slu_gemm slu_gemm_bf16 slu_gru slu_gru_step slu_gru_bf16 slu_gru_proj slu_pool slu_head slu_optim slu_framece slu_comm slu_comm_ipc slu_seq2seq"
OBJS=()
STALE=()
for f in $UNITS; do
  if [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/slu_common.h" -nt "$OUT/$f.o" ] || [ "$HERE/slu_bf16.h" -nt "$OUT/$f.o" ] || [ "$HERE/slu_philox.h" -nt "$OUT/$f.o" ] || [ "$HERE/slu_gemm_tile.h" -nt "$OUT/$f.o" ] \
     || [ "$HERE/../../include/slu_hip.h" -nt "$OUT/$f.o" ]; then
    STALE+=("$f")
  fi
  OBJS+=("$OUT/$f.o")
done
if [ "${#STALE[@]}" -gt 0 ]; then
  printf '%s\n' "${STALE[@]}" | xargs -P "$JOBS" -I{} bash -c "echo '[build] {}.hip'; '$HIPCC' $FLAGS -c '$HERE/{}.hip' -o '$OUT/{}.o.tmp' && mv '$OUT/{}.o.tmp' '$OUT/{}.o'"
fi
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${OBJS[@]}" -ldl -o "$OUT/libslu_hip.so"
echo "[build] $OUT/libslu_hip.so"
