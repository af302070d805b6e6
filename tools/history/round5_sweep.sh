#!/bin/bash
# Round-5 sweep (one gpurun call): the driver's 20-step command + the 512-step steady state of the default (bf16x3) loop
# over CU partitions, look-ahead widths and start-up ramps.   usage: tools/round5_sweep.sh <tag>
TAG=${1:-r05_sweep}
O=gpurun_out/$TAG; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-kernel-table --no-side-runs"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py $B > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d.get("steady_state") or {}
    print("%-34s value %9.0f  ms/step %.4f  fill %.3f ms  steady %9.0f (%.4f ms)  width %s  graphs %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], d["pipeline_fill_ms"], s.get("utterances_per_s", 0), s.get("ms_per_step", 0),
        d["config"]["encoder_lookahead_batches"], d["graphs_captured"]))
except Exception as e:
    print("%-34s ERR %s" % (sys.argv[1], e))
PY
}
run default_96_ramp            SLU_X=1
run old_plan_96_2slots         SLU_RAMP=0 SLU_LOOKAHEAD_SLOTS=2
run old_plan_96_3slots         SLU_RAMP=0
run cu64_ramp                  SLU_CU_SPLIT=64
run cu80_ramp                  SLU_CU_SPLIT=80
run cu48_ramp                  SLU_CU_SPLIT=48
run cu64_w20_ramp              SLU_CU_SPLIT=64 SLU_LOOKAHEAD=20
run cu96_ramp_2_5_13           SLU_RAMP=2,5,13
run cu96_ramp_4_6_10           SLU_RAMP=4,6,10
run cu64_ramp_2_5_13           SLU_CU_SPLIT=64 SLU_RAMP=2,5,13
run f16x2_guarded_96_ramp      SLU_FROZEN_MATH=auto
run exact_fp32_96_ramp         SLU_FROZEN_MATH=fp32
