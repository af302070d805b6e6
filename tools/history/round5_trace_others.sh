#!/bin/bash
# Round 5: kernel tables of the traced fully trainable and ASR pre-training steps (rocprofv3 --kernel-trace)
TAG=${1:-r05_z}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
B="--no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs"
for w in unfreeze_all asr_pretrain; do
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o u -- python $R/bench.py $B --workload $w --steps 100 --warmup 10 > $O/bench_${w}_under_rocprofv3.json 2> $O/bench_${w}_prof.err
  cd $R
  f=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_summary.py $f 34 > $O/${w}_kernel_stats.txt
  python tools/rocprof_summary.py $f 50 --by-shape > $O/${w}_kernel_stats_by_shape.txt
  rm -rf $O/trace_$w
  head -14 $O/${w}_kernel_stats.txt | cut -c1-150
done
