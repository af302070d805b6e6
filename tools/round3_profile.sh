#!/bin/bash
# Round-3 measurement pass (one gpurun call): bench lines, then rocprofv3 kernel traces of the bench.py command lines.
# usage: tools/round3_profile.sh <tag>
TAG=${1:-r03}
R=/root/repo
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export SLU_BENCH_VERBOSE=1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?"
timeout 400 python bench.py --no-cpu-baseline --no-large-batch --no-side-runs > $O/bench_512.json 2> $O/bench_512.err; echo "bench512 rc=$?"
timeout 300 python bench.py --workload seq2seq --steps 40 --warmup 10 --no-side-runs > $O/bench_seq2seq.json 2> $O/bench_seq2seq.err; echo "seq2seq rc=$?"
for d in bf16 f32; do
  timeout 300 python bench.py --dtype $d --workload unfreeze_all --seconds 10 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_$d.json 2> $O/bench_cfg4_$d.err; echo "cfg4 $d rc=$?"
done
cd /tmp && export TMPDIR=/tmp
B="--no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o d -- python $R/bench.py $B > $O/bench_default_prof.json 2> $O/bench_default_prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_unfrozen -o u -- python $R/bench.py $B --workload unfreeze_all --steps 100 --warmup 10 > $O/bench_unfrozen_prof.json 2> $O/bench_unfrozen_prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg4 -o c -- python $R/bench.py $B --dtype bf16 --workload unfreeze_all --seconds 10 --batch 32 --steps 40 --warmup 10 > $O/bench_cfg4_prof.json 2> $O/bench_cfg4_prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_seq2seq -o s -- python $R/bench.py $B --workload seq2seq --steps 40 --warmup 10 > $O/bench_seq2seq_prof.json 2> $O/bench_seq2seq_prof.err
cd $R
for t in default unfrozen cfg4 seq2seq; do
  f=$(find $O/trace_$t -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_summary.py $f 34 > $O/${t}_kernel_stats.txt
  python tools/rocprof_summary.py $f 50 --by-shape > $O/${t}_kernel_stats_by_shape.txt
  rm -rf $O/trace_$t
done
python tools/inloop_json.py $O/default_kernel_stats_by_shape.txt > $O/inloop_kernel_us.json
if [ -f end-to-end-slu_amd/lib/libslu_hip_probe.so ]; then timeout 300 python tools/gru_probe.py > $O/gru_probe.txt 2>&1; fi
timeout 400 python tools/host_inputs_probe.py 0 16 999 > $O/host_inputs_probe.txt 2>&1
head -16 $O/default_kernel_stats.txt | cut -c1-150
