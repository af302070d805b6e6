mkdir -p gpurun_out/r03_v; cd /root/repo
for f in 1 0 1 0; do
  v=$(SLU_FUSE_GRU_INPUT=$f python bench.py --steps 20 --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['pipeline_fill_ms'], (d.get('steady_state') or {}).get('utterances_per_s'), d['parity']['max_abs_logit_dev'])")
  echo "fuse_gru_input=$f: $v" | tee -a gpurun_out/r03_v/fuse.txt
done
SLU_FUSE_GRU_INPUT=1 python -m pytest tests/test_hip_bench_path.py tests/test_hip_bf16.py -q -m gpu -k "pipeline or super_batch or fused_input" 2>&1 | tail -2
