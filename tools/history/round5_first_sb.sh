#!/bin/bash
# Round-5 experiment: size and CU mask of the FIRST super-batch of the driver's 20-step command (default arithmetic).
TAG=${1:-r05_g}
O=gpurun_out/$TAG; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-kernel-table --no-side-runs"
run() {
  name=$1; shift
  env "$@" timeout 300 python bench.py $B > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d.get("steady_state") or {}
    print("%-28s value %9.0f  ms/step %.4f  fill %.3f ms  steady %9.0f (%.4f ms)  graphs %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], d["pipeline_fill_ms"], s.get("utterances_per_s", 0), s.get("ms_per_step", 0), d["graphs_captured"]))
except Exception as e:
    print("%-28s ERR %s" % (sys.argv[1], e))
PY
}
run default_12_8               SLU_X=1
run first_10                   SLU_RAMP=10
run first_14                   SLU_RAMP=14
run first_16                   SLU_RAMP=16
run first_20                   SLU_RAMP=20
run unmasked_12_8              SLU_FIRST_UNMASKED=1
run unmasked_14                SLU_FIRST_UNMASKED=1 SLU_RAMP=14
run unmasked_16                SLU_FIRST_UNMASKED=1 SLU_RAMP=16
run unmasked_20                SLU_FIRST_UNMASKED=1 SLU_RAMP=20
