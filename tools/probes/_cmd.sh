mkdir -p gpurun_out/r06_n
B="--no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs"
for cfg in "20 1" "40 2" "40 1" "32 2" "30 1" "20 1" "40 2"; do set -- $cfg
  SLU_MAX_TABLE=63 SLU_LOOKAHEAD=$1 SLU_GRU_TILES=$2 timeout 300 python bench.py $B > gpurun_out/r06_n/bench_512_la$1_t$2.json 2> gpurun_out/r06_n/bench_la$1_t$2.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r06_n/bench_512_la$1_t$2.json").read().strip().splitlines()[-1])
    print("lookahead $1 tiles $2:", d["value"], d["ms_per_step"])
except Exception as e: print("lookahead $1 tiles $2: ERR", e)
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-side-runs --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default 20-step', d['value'], d.get('steady_state'), d.get('scaling_model'))"
