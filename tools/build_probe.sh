#!/bin/bash
# Probe build: libslu_hip_probe.so = the product objects with slu_gru.o replaced by a PROBE copy of csrc/slu_gru.hip
# (tools/probes/make_gru4_probe.py: ablation switches read from SLU_GRU_DBG at launch; the product source carries no hooks).
# Load it with SLU_HIP_LIB=<path> (tools/gru_probe.py).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
C="$HERE/../end-to-end-slu_amd/csrc"; O="$HERE/../end-to-end-slu_amd/lib"; ALT="$HERE/../end-to-end-slu_amd/lib_alt"
mkdir -p "$ALT"
bash "$C/build.sh" > /dev/null
python "$HERE/probes/make_gru4_probe.py" "$ALT/slu_gru_probe.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I"$C" -I"$HERE/../include" ${SLU_PROBE_FLAGS:-} -c "$ALT/slu_gru_probe.hip" -o "$ALT/slu_gru_probe.o"
OBJS=$(ls "$O"/slu_*.o | grep -v "slu_gru.o$" | grep -v probe)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS "$ALT/slu_gru_probe.o" -ldl -o "$ALT/libslu_hip_probe.so"
echo "$ALT/libslu_hip_probe.so"
