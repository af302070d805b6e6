mkdir -p gpurun_out/r03_s; cd /root/repo
python -m pytest tests/test_hip_model.py tests/test_hip_train_loop.py -q -m gpu -rP > gpurun_out/r03_s/model.txt 2>&1; grep -E "passed|failed" gpurun_out/r03_s/model.txt | tail -1; grep -E "SLU_TRAIN_MATH" gpurun_out/r03_s/model.txt | head
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline_fill_ms'], (d.get('steady_state') or {}).get('utterances_per_s'))"
done
