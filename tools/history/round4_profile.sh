#!/bin/bash
# Round-4 measurement pass (one gpurun call): rocprofv3 kernel trace of the default bench.py command (in-loop durations),
# PMC passes of the dominant kernel in its product form (HBM traffic; SQ activity), then the bench lines.
#   usage: tools/round4_profile.sh <tag>
TAG=${1:-r04}
R=$PWD
O=$R/gpurun_out/$TAG; mkdir -p $O
export SLU_BENCH_VERBOSE=1
cd /tmp && export TMPDIR=/tmp
B="--no-kernel-table --no-cpu-baseline --no-large-batch --no-side-runs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o d -- python $R/bench.py $B > $O/bench_default_under_rocprofv3.json 2> $O/bench_default_prof.err
cd $R
f=$(find $O/trace_default -name "*kernel_trace.csv" | head -1)
python tools/rocprof_summary.py $f 34 > $O/default_kernel_stats.txt
python tools/rocprof_summary.py $f 50 --by-shape > $O/default_kernel_stats_by_shape.txt
rm -rf $O/trace_default
python tools/inloop_json.py $O/default_kernel_stats_by_shape.txt > $O/inloop_kernel_us.json
sed -i "s#$O/#profiles/${TAG}_#" $O/inloop_kernel_us.json
cp $O/inloop_kernel_us.json profiles/inloop_kernel_us.json
# dominant kernel, product form, 20-batch super-batch (1280 sequences): HBM traffic (separate passes; KiB, FETCH x 2 on gfx950)
cd /tmp
for w in gru_bf_pool_fused gru_bf_pool; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${w}_$c -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
  done
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_${w}_SQ -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_${w}_SQ2 -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
done
cd $R
python - <<PY > $O/pmc_gru_bf.txt 2>&1
import csv, glob, json, collections
O = "$O"
def counters(dirname):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (O, dirname), recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gru_bf_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, (max(len(v) for v in acc.values()) if acc else 0)
def durations(dirname):
    f = glob.glob("%s/%s/**/*kernel_trace.csv" % (O, dirname), recursive=True)
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "gru_bf_fwd" in r["Kernel_Name"]]
B, H, D, I = 1280, 128, 2, 60
shapes, tot_all, alg_all = {}, 0.0, 0.0
for w, label, T, in_bytes in (("gru_bf_pool_fused", "T=300 B=1280 H=128 D=2 K=60: fused input projection + dropout/pool epilogue (x planes in, pooled planes out)", 300, lambda T: 2.0 * 2 * T * B * 64 + 2.0 * 2 * D * 3 * H * 64),
                              ("gru_bf_pool", "T=150 B=1280 H=128 D=2: gx in, dropout/pool epilogue (pooled planes out)", 150, lambda T: 4.0 * T * B * D * 3 * H)):
    fe, n = counters("pmc_%s_FETCH_SIZE" % w); wr, _ = counters("pmc_%s_WRITE_SIZE" % w)
    fe, wr = fe["FETCH_SIZE"], wr["WRITE_SIZE"]
    alg = in_bytes(T) + 2.0 * 2 * ((T + 1) // 2) * B * D * H + T * B * D * H / 8.0 + 4.0 * D * 3 * H * H
    tot = 2 * fe * 1024 + wr * 1024
    print("%s: %d launches, fetch x 2 = %d B, write = %d B, total %d B; algorithmic %d B; ratio %.3f" % (label, n, 2 * fe * 1024, wr * 1024, tot, alg, tot / alg))
    shapes[label] = {"fetch_x2_bytes": round(2 * fe * 1024), "write_bytes": round(wr * 1024), "algorithmic_bytes": round(alg)}
    tot_all += tot; alg_all += alg
    sq, n = counters("pmc_%s_SQ" % w); sq2, _ = counters("pmc_%s_SQ2" % w)
    us = durations("pmc_%s_SQ" % w)
    waves = (B // 16) * D * 8
    print("   SQ counters (mean of %d launches, %.1f us each, %d waves, whole chip unmasked): %s" % (n, sum(us) / len(us), waves, {k: round(v) for k, v in sq.items()}))
    print("   per wave and step: wave cycles %.0f, active %.0f, VALU-active %.0f, waiting (s_waitcnt / barrier) %.0f, issue stalls %.0f; MFMA pipe busy %.0f cycles per wave-step"
          % (tuple(4 * sq[k] / waves / T for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")) + (sq["SQ_VALU_MFMA_BUSY_CYCLES"] / waves / T,)))
    print("   instructions per wave and step:", {k: round(v / waves / T, 1) for k, v in sq2.items()})
j = json.load(open("profiles/pmc_traffic.json"))
j["gru_bf_fwd_kernel<128,2>"] = {
    "bytes_per_launch_mean_of_measured_shapes": round(tot_all / 2), "shapes": shapes,
    "traffic_over_algorithmic": round(tot_all / alg_all, 3),
    "source": "profiles/${TAG}_pmc_gru_bf.txt",
    "note": "the two launch kinds of the default 20-batch (1280-sequence) super-batch in their round-4 product form (Dropout + avg-pool in the "
            "epilogue, plane output): the first layer's (fused input projection, T = 300) and a K = 256 layer's (gx in, T = 150); "
            "roofline.algorithmic_bytes_per_launch in the bench line averages the four launch shapes"}
json.dump(j, open("profiles/pmc_traffic.json", "w"), indent=1)
json.dump(j, open(O + "/pmc_traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc_gru_bf_pool*
cat $O/pmc_gru_bf.txt
# the Sinc launch's SQ / memory counters (1024 sequences, whole chip)
bash tools/pmc_kernel.sh gpurun_out/$TAG/pmc_sinc wconv_sinc 1024 wconv_bf_fwd > $O/pmc_wconv_sinc.txt 2>&1
rm -rf $O/pmc_sinc
# the convolution launches at 1024 sequences on the look-ahead partition
python tools/wconv_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $O/wconv_probe.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?"
timeout 400 python bench.py --no-cpu-baseline --no-large-batch --no-side-runs > $O/bench_512.json 2> $O/bench_512.err; echo "bench512 rc=$?"
if [ "${TRACE_OTHERS:-1}" = "1" ]; then
  # fully trainable step, ASR pre-training: kernel tables of the traced runs; configs[4] shape in both arithmetics
  for w in unfreeze_all asr_pretrain; do
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -o u -- python $R/bench.py $B --workload $w --steps 100 --warmup 10 > $O/bench_${w}_under_rocprofv3.json 2> $O/bench_${w}_prof.err
    cd $R
    f=$(find $O/trace_$w -name "*kernel_trace.csv" | head -1)
    python tools/rocprof_summary.py $f 34 > $O/${w}_kernel_stats.txt
    python tools/rocprof_summary.py $f 50 --by-shape > $O/${w}_kernel_stats_by_shape.txt
    rm -rf $O/trace_$w
  done
  for d in bf16 f32; do
    timeout 300 python bench.py --dtype $d --workload unfreeze_all --seconds 10 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-table > $O/bench_cfg4_$d.json 2> $O/bench_cfg4_$d.err; echo "cfg4 $d rc=$?"
  done
fi
head -20 $O/default_kernel_stats.txt | cut -c1-160
python - <<PY
import json
for f in ("bench_20", "bench_512"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], d.get("steady_state"), {k: r.get(k) for k in ("kernel", "bound", "frac", "frac_isolated", "frac_in_loop", "prefix_traffic_over_8d", "traffic", "prefix_ms_per_super_batch_isolated")})
        for k in ("exact_fp32", "frozen_bf16x3", "host_inputs", "other_workloads", "cpu_baseline", "parity"):
            if d.get(k): print("   ", k, json.dumps(d[k])[:600])
    except Exception as e:
        print(f, "ERR", e)
PY
