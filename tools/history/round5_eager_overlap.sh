O=gpurun_out/r05_k; mkdir -p $O
run() {
  name=$1; wl=$2; shift; shift
  env "$@" timeout 300 python bench.py --workload $wl --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-table --no-side-runs --no-large-batch > $O/$name.json 2> $O/$name.err
  python - "$name" "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    s = d.get("steady_state") or {}
    print("%-34s value %8.0f  ms/step %.4f  steady %8.0f (%.4f ms) graphs %s" % (sys.argv[1], d["value"], d["ms_per_step"], s.get("utterances_per_s", 0), s.get("ms_per_step", 0), d["graphs_captured"]))
except Exception as e:
    print("%-34s ERR %s" % (sys.argv[1], e))
PY
}
run graph_layer          unfreeze_all SLU_WGRAD_BRANCH=layer
run eager_layer          unfreeze_all SLU_GRAPHS=0 SLU_WGRAD_BRANCH=layer
run eager_pass           unfreeze_all SLU_GRAPHS=0 SLU_WGRAD_BRANCH=pass
run eager_pass_512       unfreeze_all SLU_GRAPHS=0 SLU_WGRAD_BRANCH=pass SLU_WGRAD_WGS=512
run eager_off            unfreeze_all SLU_GRAPHS=0 SLU_WGRAD_BRANCH=0
