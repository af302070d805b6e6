mkdir -p gpurun_out/r03_j; cd /root/repo
for f in tests/test_hip_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -q -m gpu -rP > gpurun_out/r03_j/$n.txt 2>&1
  echo "$n: $(tail -1 gpurun_out/r03_j/$n.txt)"
  awk '/=+ FAILURES =+/{p=1} /=+ PASSES =+/{p=0} p && (/^_+ .* _+$/ || /^E    +(Assertion|assert)/)' gpurun_out/r03_j/$n.txt | cut -c1-220 | head -20
done
for w in unfreeze_all asr_pretrain; do
  for m in split fp32; do
    SLU_TRAIN_MATH=$m python bench.py --workload $w --steps 30 --warmup 5 --no-side-runs --no-kernel-table > gpurun_out/r03_j/bench_${w}_$m.json 2> gpurun_out/r03_j/bench_${w}_$m.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_j/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("steady_state"))
    except Exception as e:
        print(f, "ERR", e)
PY
