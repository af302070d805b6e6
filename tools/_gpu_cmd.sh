O=/root/repo/gpurun_out/r03_p; mkdir -p $O; R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/tools/run_one.py tn > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/pmc_tcc -o p -- python $R/tools/run_one.py tn > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $O/pmc_tcp -o p -- python $R/tools/run_one.py tn > /dev/null 2>&1
cd $R
for c in sq tcc tcp; do
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  k=$(find $O/pmc_$c -name "*kernel_trace.csv" | head -1)
  echo "== $c"; python - <<PY
import csv, collections
try:
    rows = list(csv.DictReader(open("$f")))
    agg = collections.defaultdict(list)
    for r in rows:
        if "gemm_tn_small" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        print(c, "launches", len(v), "mean %.4g" % (sum(v) / len(v)))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("$k")) if "gemm_tn_small" in r["Kernel_Name"]]
    if dur: print("kernel duration under this pass: mean %.1f us over %d launches" % (sum(dur) / len(dur), len(dur)))
except Exception as e:
    print("ERR", e)
PY
done 2>&1 | tee $O/pmc_tn.txt
rm -rf $O/pmc_sq $O/pmc_tcc $O/pmc_tcp
