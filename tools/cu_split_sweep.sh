# CU partition x look-ahead width sweep of the default workload: the driver's 20-step value and the 512-step steady state
CONFIGS=${CONFIGS:-"96 20;128 16;144 14;160 12;128 12;144 12;112 16;160 10"}
IFS=';' read -ra CFG <<< "$CONFIGS"
for cfg in "${CFG[@]}"; do set -- $cfg
  for steps in 20 512; do
    v=$(SLU_CU_SPLIT=$1 SLU_LOOKAHEAD=$2 timeout 200 python bench.py --steps $steps --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "split=$1 lookahead=$2 steps=$steps: $v"
  done
done
