#!/bin/bash
# Round-5 experiment: weight-gradient launches of long GRU layers left open until the end of backward (fully trainable steps)
TAG=${1:-r05_h}
O=gpurun_out/$TAG; mkdir -p $O
run() {
  name=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-table > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d.get("steady_state") or {}
    print("%-34s value %8.0f  ms/step %.4f  steady %8.0f (%.4f ms)  loss %s graphs %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], s.get("utterances_per_s", 0), s.get("ms_per_step", 0), d["config"]["mean_loss"], d["graphs_captured"]))
except Exception as e:
    print("%-34s ERR %s" % (sys.argv[1], e))
PY
}
run unfreeze_all_off        unfreeze_all SLU_WGRAD_BRANCH=0
run unfreeze_all_join_216   unfreeze_all SLU_WGRAD_BRANCH=layer
run unfreeze_all_defer_216  unfreeze_all SLU_WGRAD_BRANCH=pass
run unfreeze_all_defer_144  unfreeze_all SLU_WGRAD_BRANCH=pass SLU_WGRAD_WGS=144
run unfreeze_all_defer_288  unfreeze_all SLU_WGRAD_BRANCH=pass SLU_WGRAD_WGS=288
run unfreeze_all_defer_512  unfreeze_all SLU_WGRAD_BRANCH=pass SLU_WGRAD_WGS=512
run asr_off                 asr_pretrain SLU_WGRAD_BRANCH=0
run asr_defer_216           asr_pretrain SLU_WGRAD_BRANCH=pass
