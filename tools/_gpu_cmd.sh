mkdir -p gpurun_out/r03_t; cd /root/repo
for m in bf16x3 fp32; do
  SLU_FROZEN_MATH=$m timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_train_loop.py tests/test_hip_bench_path.py tests/test_hip_seq2seq.py -q -m gpu -x > gpurun_out/r03_t/$m.txt 2>&1
  echo "SLU_FROZEN_MATH=$m: $(grep -E "passed|failed" gpurun_out/r03_t/$m.txt | tail -1)"
  grep -E "^E  |^FAILED" gpurun_out/r03_t/$m.txt | head -5
done
