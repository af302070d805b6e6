mkdir -p gpurun_out/r03_m; cd /root/repo
for f in tests/test_hip_ops.py tests/test_hip_model.py tests/test_hip_bench_path.py tests/test_hip_train_loop.py tests/test_hip_dp.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -q -m gpu -rP > gpurun_out/r03_m/$n.txt 2>&1
  echo "$n: $(tail -1 gpurun_out/r03_m/$n.txt)"
  awk '/=+ FAILURES =+/{p=1} /=+ PASSES =+/{p=0} p && (/^_+ .* _+$/ || /^E    +(Assertion|assert)/)' gpurun_out/r03_m/$n.txt | cut -c1-220 | head -20
done
python tools/suffix_bench.py 2>&1 | grep -E "head|dropout" | tee gpurun_out/r03_m/suffix.txt
for steps in 20 512; do
  for fuse in 1 0; do
    v=$(SLU_FUSE_HEAD_DROPOUT=$fuse python bench.py --steps $steps --warmup 5 --no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "fuse=$fuse steps=$steps: $v" | tee -a gpurun_out/r03_m/suffix.txt
  done
done
