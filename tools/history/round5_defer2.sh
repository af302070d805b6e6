#!/bin/bash
TAG=${1:-r05_i}
O=gpurun_out/$TAG; mkdir -p $O
run() {
  name=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-table > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d.get("steady_state") or {}
    print("%-34s value %8.0f  ms/step %.4f  steady %8.0f (%.4f ms)" % (sys.argv[1], d["value"], d["ms_per_step"], s.get("utterances_per_s", 0), s.get("ms_per_step", 0)))
except Exception as e:
    print("%-34s ERR %s" % (sys.argv[1], e))
PY
}
run unfreeze_all_join_512   unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=512
run unfreeze_all_join_360   unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=360
run unfreeze_all_join_288   unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=288
run unfreeze_all_join_216   unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=216
run unfreeze_all_join_144   unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=144
run unfreeze_all_join_72    unfreeze_all SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=72
run asr_join_216            asr_pretrain SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=216
run asr_join_512            asr_pretrain SLU_WGRAD_BRANCH=layer SLU_WGRAD_WGS=512
run cfg4_off                unfreeze_all SLU_WGRAD_BRANCH=0
