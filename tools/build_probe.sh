#!/bin/bash
# Probe build: libslu_hip_probe.so = the product objects with slu_gru.hip recompiled under -DSLU_GRU_PROBE
# (ablation switches read from SLU_GRU_DBG at launch).  Load it with SLU_HIP_LIB=<path> (tools/gru_probe.py).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../end-to-end-slu_amd/csrc"; O="$HERE/../end-to-end-slu_amd/lib"
bash "$C/build.sh" > /dev/null
for f in slu_gru slu_gru_bf16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DSLU_GRU_PROBE ${SLU_PROBE_FLAGS:-} -c "$C/$f.hip" -o "$O/${f}_probe.o"
done
OBJS=$(ls "$O"/slu_*.o | grep -v "slu_gru.o$" | grep -v "slu_gru_bf16.o$" | grep -v probe)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$O/slu_gru_probe.o" "$O/slu_gru_bf16_probe.o" -ldl -o "$O/libslu_hip_probe.so"
echo "$O/libslu_hip_probe.so"
