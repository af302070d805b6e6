#!/bin/bash
# Round-3 measurement pass (one gpurun call): bench lines, rocprofv3 kernel traces of the bench.py command lines, PMC
# traffic of the dominant kernel.   usage: tools/round3_profile.sh <tag>
TAG=${1:-r03}
R=/root/repo
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
export SLU_BENCH_VERBOSE=1
cd /tmp && export TMPDIR=/tmp
B="--no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs"
# the trace of the default command first: profiles/inloop_kernel_us.json feeds roofline.frac of the bench lines below
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o d -- python $R/bench.py $B > $O/bench_default_prof.json 2> $O/bench_default_prof.err
cd $R
f=$(find $O/trace_default -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $f 34 > $O/default_kernel_stats.txt
python tools/rocprof_summary.py $f 50 --by-shape > $O/default_kernel_stats_by_shape.txt
rm -rf $O/trace_default
python tools/inloop_json.py $O/default_kernel_stats_by_shape.txt > $O/inloop_kernel_us.json
sed -i "s#$O/#profiles/${TAG}_#" $O/inloop_kernel_us.json
cp $O/inloop_kernel_us.json profiles/inloop_kernel_us.json
# HBM traffic of the dominant kernel (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KiB, FETCH x 2 on gfx950): its two
# launch kinds — T = 300 with the fused input projection (reads x planes), T = 150 plain (reads gx)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/tools/run_one.py gru_bf_fused 1024 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc2_$c -o p -- python $R/tools/run_one.py gru_bf 1024 2 > /dev/null 2>&1
done
cd $R
python - <<PY > $O/pmc_gru_bf.txt 2>&1
import csv, glob, json
def mean(dirname, c):
    f = glob.glob("$O/%s_%s/**/*counter_collection.csv" % (dirname, c), recursive=True)
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "gru_bf_fwd" in r["Kernel_Name"] and r["Counter_Name"] == c]
    return sum(v) / len(v), len(v)
T, B, H, D, I = 300, 1024, 128, 2, 60
shapes, tot_all, alg_all = {}, 0.0, 0.0
for dirname, label, alg in (("pmc", "T=300 B=1024 H=128 D=2 K=60 fused input projection", 2.0 * 2 * T * B * 64 + 4.0 * (T * B * D * H + D * 3 * H * H) + 2.0 * 2 * D * 3 * H * 64),
                            ("pmc2", "T=300 B=1024 H=128 D=2 (plain: gx in)", 4.0 * (T * B * D * 3 * H + T * B * D * H + D * 3 * H * H))):
    fe, n = mean(dirname, "FETCH_SIZE"); wr, _ = mean(dirname, "WRITE_SIZE")
    tot = 2 * fe * 1024 + wr * 1024
    print("%s: %d launches, fetch x 2 = %d B, write = %d B, total %d B; algorithmic %d B; ratio %.3f" % (label, n, 2 * fe * 1024, wr * 1024, tot, alg, tot / alg))
    shapes[label] = {"fetch_x2_bytes": round(2 * fe * 1024), "write_bytes": round(wr * 1024), "algorithmic_bytes": round(alg)}
    tot_all += tot; alg_all += alg
j = json.load(open("profiles/pmc_traffic.json"))
j["gru_bf_fwd_kernel<128,2>"] = {
    "bytes_per_launch_mean_of_measured_shapes": round(tot_all / 2), "shapes": shapes,
    "traffic_over_algorithmic": round(tot_all / alg_all, 3),
    "source": "profiles/${TAG}_pmc_gru_bf.txt",
    "note": "T = 300 launches of the default 16-batch (1024-sequence) super-batch: the first layer's (fused input projection: x planes in) and "
            "the plain kernel at the same size (gx in; the default's plain launches are T = 150 / 75 / 38); roofline.algorithmic_bytes_per_launch "
            "in the bench line averages the four launch shapes"}
json.dump(j, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(j, open("$O/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc2_FETCH_SIZE $O/pmc2_WRITE_SIZE
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/pmc_gru_bf.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?"
timeout 400 python bench.py --no-cpu-baseline --no-large-batch --no-side-runs > $O/bench_512.json 2> $O/bench_512.err; echo "bench512 rc=$?"
timeout 300 python bench.py --workload seq2seq --steps 40 --warmup 10 --no-side-runs > $O/bench_seq2seq.json 2> $O/bench_seq2seq.err; echo "seq2seq rc=$?"
for d in bf16 f32; do
  timeout 300 python bench.py --dtype $d --workload unfreeze_all --seconds 10 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_cfg4_$d.json 2> $O/bench_cfg4_$d.err; echo "cfg4 $d rc=$?"
done
if [ "${TRACE_UNFROZEN:-1}" = "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_unfrozen -o u -- python $R/bench.py $B --workload unfreeze_all --steps 100 --warmup 10 > $O/bench_unfrozen_prof.json 2> $O/bench_unfrozen_prof.err
  cd $R
  f=$(find $O/trace_unfrozen -name "*kernel_trace.csv" | head -1)
  python tools/rocprof_summary.py $f 34 > $O/unfrozen_kernel_stats.txt
  python tools/rocprof_summary.py $f 50 --by-shape > $O/unfrozen_kernel_stats_by_shape.txt
  rm -rf $O/trace_unfrozen
fi
head -16 $O/default_kernel_stats.txt | cut -c1-150
python - <<PY
import json
for f in ("bench_20", "bench_512", "bench_seq2seq", "bench_cfg4_bf16", "bench_cfg4_f32"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], d.get("steady_state"), {k: r.get(k) for k in ("kernel", "frac", "frac_isolated", "frac_in_loop", "prefix_traffic_over_8d", "traffic")})
        for k in ("exact_fp32", "frozen_bf16x3", "host_inputs", "other_workloads", "cpu_baseline", "parity"):
            if d.get(k): print("   ", k, json.dumps(d[k])[:300])
    except Exception as e:
        print(f, "ERR", e)
PY
