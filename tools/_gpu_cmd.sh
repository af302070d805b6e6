mkdir -p gpurun_out/r03_q; cd /root/repo
for f in tests/test_hip_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -q -m gpu -rP > gpurun_out/r03_q/$n.txt 2>&1
  echo "$n: $(grep -E "passed|failed|error" gpurun_out/r03_q/$n.txt | tail -1)"
  awk '/=+ FAILURES =+/{p=1} /=+ PASSES =+/{p=0} p && (/^_+ .* _+$/ || /^E    +(Assertion|assert)/)' gpurun_out/r03_q/$n.txt | cut -c1-220 | head -20
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python bench.py --gpus 2 --share-gpu --steps 10 --warmup 3 --no-side-runs --no-kernel-table --no-cpu-baseline --no-large-batch 2>&1 | tail -1 | cut -c1-400
