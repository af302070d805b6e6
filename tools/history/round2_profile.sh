#!/bin/bash
# rocprofv3 passes of round 2 (run on the GPU box through gpurun): kernel traces of the bench.py command lines and
# separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ set) on the dominant kernels' headline launch shapes.
TAG=${1:-r02_prof}
O=/root/repo/gpurun_out/$TAG; mkdir -p $O
R=/root/repo
cd /tmp && export TMPDIR=/tmp
B="--no-kernel-table --no-cpu-baseline --no-large-batch"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o d -- python $R/bench.py $B > $O/bench_default_prof.json 2> $O/bench_default_prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_unfrozen -o u -- python $R/bench.py $B --workload unfreeze_all --steps 100 --warmup 10 > $O/bench_unfrozen_prof.json 2> $O/bench_unfrozen_prof.err
for W in gemm_bf_ip0 gemm_bf_ip1 gru_bf; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$W -o p -- python $R/tools/run_one.py $W > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$W -o p -- python $R/tools/run_one.py $W > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq_$W -o p -- python $R/tools/run_one.py $W > /dev/null 2>&1
done
cd $R
for t in default unfrozen; do
  f=$(find $O/trace_$t -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_summary.py $f 30 > $O/${t}_kernel_stats.txt
  python tools/rocprof_summary.py $f 45 --by-shape > $O/${t}_kernel_stats_by_shape.txt
done
for W in gemm_bf_ip0 gemm_bf_ip1 gru_bf; do for c in fetch write sq; do
  f=$(find $O/pmc_${c}_$W -name "*counter_collection.csv" | head -1)
  k=$(find $O/pmc_${c}_$W -name "*kernel_trace.csv" | head -1)
  echo "== $W $c"; python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$f")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_bf_kernel" in r["Kernel_Name"] or "gemm_bf_panel" in r["Kernel_Name"] or "gru_bf_fwd" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, cs in agg.items():
    for c, v in cs.items():
        print(kname, c, "launches", len(v), "mean", sum(v) / len(v))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("$k")) if "gemm_bf_kernel" in r["Kernel_Name"] or "gemm_bf_panel" in r["Kernel_Name"] or "gru_bf_fwd" in r["Kernel_Name"]]
if dur: print("kernel duration under this pass: mean %.1f us over %d launches" % (sum(dur) / len(dur), len(dur)))
PY
done; done > $O/pmc_summary.txt 2>&1
head -14 $O/default_kernel_stats.txt | cut -c1-160; cat $O/pmc_summary.txt; head -c 300 $O/bench_default_prof.json
