mkdir -p gpurun_out/r03_w; cd /root/repo
for f in 1 0 1 0; do
  v=$(SLU_ONE_STEP_GRAPH=$f python bench.py --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline_fill_ms'], (d.get('steady_state') or {}).get('utterances_per_s'), d['graphs_captured'])")
  echo "one_step_graph=$f: $v" | tee -a gpurun_out/r03_w/onegraph.txt
done
python -m pytest tests/test_hip_bench_path.py tests/test_hip_train_loop.py tests/test_hip_dp.py -q -m gpu 2>&1 | tail -2
