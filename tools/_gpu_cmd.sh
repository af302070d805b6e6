mkdir -p gpurun_out/r03_r; cd /root/repo
for w in 1 0; do echo "== SLU_WIDE_FILL=$w"; SLU_WIDE_FILL=$w python tools/pipeline_timeline.py --lookahead 16 --steps 20 2>&1 | grep -E "prefix|steps "; done | tee gpurun_out/r03_r/timeline_wide.txt
for w in 1 0 1 0; do
  v=$(SLU_WIDE_FILL=$w python bench.py --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline_fill_ms'], (d.get('steady_state') or {}).get('utterances_per_s'))")
  echo "wide=$w: $v" | tee -a gpurun_out/r03_r/timeline_wide.txt
done
python -m pytest tests/test_hip_bench_path.py tests/test_hip_train_loop.py -q -m gpu 2>&1 | tail -2
