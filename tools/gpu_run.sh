#!/bin/bash
# One gpurun call: GPU tests, then bench lines (all output under gpurun_out/<tag>/).
# usage: tools/gpu_run.sh <tag> [pytest-args...]
TAG=${1:-run}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export SLU_BENCH_VERBOSE=1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err; echo "bench20 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-large-batch > $OUT/bench_512.json 2> $OUT/bench_512.err; echo "bench512 rc=$?"
timeout 300 python bench.py --workload unfreeze_all --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench_unfrozen.json 2> $OUT/bench_unfrozen.err; echo "unfrozen rc=$?"
timeout 300 python bench.py --workload asr_pretrain --steps 100 --warmup 10 > $OUT/bench_asr.json 2> $OUT/bench_asr.err; echo "asr rc=$?"
for f in bench_20 bench_512 bench_unfrozen bench_asr; do echo "== $f"; tail -c 1500 $OUT/$f.err | tail -5; python - <<PY
import json
try:
    d=json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print({k:d.get(k) for k in ("value","ms_per_step","pipeline_fill_ms","graphs_captured","parity","steady_state")})
    print("roofline", {k:r.get(k) for k in ("kernel","bound","achieved","peak","frac","avg_launch_ms")})
    for k in r.get("kernels",[]): print("  ", k["kernel"], k["bound"], k["frac"], k["algorithmic_tflops"], k["gpu_ms_per_step"], [(s["us"], s["algorithmic_tflops"]) for s in k["shapes"]])
    print("cpu", d.get("cpu_baseline"))
except Exception as e: print("no json:", e)
PY
done
