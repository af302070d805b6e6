mkdir -p gpurun_out/r03_x; cd /root/repo
for f in tests/test_hip_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -q -m gpu -rP > gpurun_out/r03_x/$n.txt 2>&1
  echo "$n: $(grep -E "passed|failed|error" gpurun_out/r03_x/$n.txt | tail -1)"
  awk '/=+ FAILURES =+/{p=1} /=+ PASSES =+/{p=0} p && (/^_+ .* _+$/ || /^E    +(Assertion|assert)/)' gpurun_out/r03_x/$n.txt | cut -c1-220 | head -20
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
