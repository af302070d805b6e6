#!/bin/bash
# Round-5 measurement pass (one gpurun call): rocprofv3 kernel trace of the default bench.py command (in-loop durations of
# the DEFAULT arithmetic, bf16x3), PMC passes of the dominant kernel in its product form (HBM traffic; SQ activity), then
# the bench lines.    usage: tools/round5_profile.sh <tag>
TAG=${1:-r05_z}
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
# dominant kernel (gru_bf_fwd_kernel<128,3>), product form, 20-batch super-batch (1280 sequences): HBM traffic in separate
# passes (KiB; FETCH x 2 on gfx950, MI355X_MICROARCH.md), then SQ activity
cd /tmp
for w in gru_bf3_pool_300 gru_bf3_pool_150; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${w}_$c -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
  done
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_${w}_SQ -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_${w}_SQ2 -o p -- python $R/tools/run_one.py $w 1280 > /dev/null 2>&1
done
cd $R
python - <<PY > $O/pmc_gru_bf.txt 2>&1
import csv, glob, collections
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
B, H, D, NS = 1280, 128, 2, 3
print("gru_bf_fwd_kernel<128,3> (bf16x3, the default arithmetic of frozen layers), product form of a 1280-sequence super-batch:")
print("gx (fp32) in, Dropout(0.5) + avg-pool(2) in the epilogue, three bf16 planes of the pooled output out; whole chip, unmasked")
tot_all = alg_all = 0.0
for w, T in (("gru_bf3_pool_300", 300), ("gru_bf3_pool_150", 150)):
    fe, n = counters("pmc_%s_FETCH_SIZE" % w); wr, _ = counters("pmc_%s_WRITE_SIZE" % w)
    fe, wr = fe["FETCH_SIZE"], wr["WRITE_SIZE"]
    # algorithmic bytes of the launch: gx in, pooled planes out, keep bits in, W_hh in
    alg = 4.0 * T * B * D * 3 * H + 2.0 * NS * ((T + 1) // 2) * B * D * H + T * B * D * H / 8.0 + 4.0 * D * 3 * H * H
    tot = 2 * fe * 1024 + wr * 1024
    us = durations("pmc_%s_FETCH_SIZE" % w)
    print("T=%d: %d launches, %.1f us each; fetch x 2 = %d B, write = %d B, total %d B; algorithmic %d B; traffic / algorithmic %.3f; "
          "algorithmic HBM rate %.3f TB/s = %.3f of 8 TB/s" % (T, n, sum(us) / len(us), 2 * fe * 1024, wr * 1024, tot, alg, tot / alg,
                                                              alg / (sum(us) / len(us) * 1e-6) / 1e12, alg / (sum(us) / len(us) * 1e-6) / 8e12))
    tot_all += tot; alg_all += alg
    sq, n = counters("pmc_%s_SQ" % w); sq2, _ = counters("pmc_%s_SQ2" % w)
    us = durations("pmc_%s_SQ" % w)
    waves = (B // 16) * D * 8
    flops = 2.0 * B * H * 3 * H * D * T
    print("   algorithmic %.2f GFLOP in %.1f us = %.1f TFLOP/s fp32-equivalent = %.4f of the 2500 TFLOP/s bf16 MFMA peak (x 6 issued products: %.4f)"
          % (flops / 1e9, sum(us) / len(us), flops / (sum(us) / len(us) * 1e-6) / 1e12, flops / (sum(us) / len(us) * 1e-6) / 2.5e15,
             6 * flops / (sum(us) / len(us) * 1e-6) / 2.5e15))
    print("   SQ counters (mean of %d launches, %.1f us each, %d waves): %s" % (n, sum(us) / len(us), waves, {k: round(v) for k, v in sq.items()}))
    print("   per wave and step: wave cycles %.0f, active %.0f, VALU-active %.0f, waiting (s_waitcnt / barrier) %.0f, issue stalls %.0f; MFMA pipe busy %.0f cycles per wave-step"
          % (tuple(4 * sq[k] / waves / T for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")) + (sq["SQ_VALU_MFMA_BUSY_CYCLES"] / waves / T,)))
    print("   instructions per wave and step:", {k: round(v / waves / T, 1) for k, v in sq2.items()})
print("both shapes: traffic / algorithmic = %.3f" % (tot_all / alg_all))
PY
rm -rf $O/pmc_gru_bf3_pool*
cat $O/pmc_gru_bf.txt
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench20 rc=$?"
timeout 400 python bench.py --no-cpu-baseline --no-large-batch --no-side-runs > $O/bench_512.json 2> $O/bench_512.err; echo "bench512 rc=$?"
timeout 400 python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 --no-cpu-baseline --no-large-batch --no-side-runs --no-kernel-table > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err; echo "bench 2 ranks rc=$?"
head -24 $O/default_kernel_stats.txt | cut -c1-170
python - <<PY
import json
for f in ("bench_20", "bench_512", "bench_2ranks_one_gpu"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f, d["value"], d["ms_per_step"], d.get("steady_state"), {k: r.get(k) for k in ("kernel", "bound", "frac", "frac_hbm", "frac_mfma_algorithmic", "mfma_issued_frac", "prefix_traffic_over_8d", "prefix_ms_per_super_batch_isolated")})
        for k in ("exact_fp32", "frozen_f16x2", "host_inputs", "other_workloads", "cpu_baseline", "parity", "rccl"):
            if d.get(k): print("   ", k, json.dumps(d[k])[:700])
    except Exception as e:
        print(f, "ERR", e)
PY
