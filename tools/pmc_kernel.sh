#!/bin/bash
# SQ / memory counter passes on ONE kernel of a super-batch (whole chip, separate runs per counter set).
#   usage: tools/pmc_kernel.sh <out dir> [run_one.py mode = wconv_sinc] [size = 1024] [kernel name pattern = wconv_bf_fwd]
R=$PWD; O=$R/${1:-gpurun_out/pmc_kernel}; mkdir -p $O
MODE=${2:-wconv_sinc}; SIZE=${3:-1024}; PAT=${4:-wconv_bf_fwd}
cd /tmp && export TMPDIR=/tmp
run() { timeout 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -o p -- python $R/tools/run_one.py $MODE $SIZE > /dev/null 2>&1; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run c "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM"
run d "FETCH_SIZE"
run e "WRITE_SIZE"
run f "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum"
cd $R
python - <<PY
import csv, glob, collections
O = "$O"; PAT = "$PAT"
for d in "abcdef":
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (O, d), recursive=True)
    if not f:
        print(d, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if PAT in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = glob.glob("%s/%s/**/*kernel_trace.csv" % (O, d), recursive=True)
    us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(t[0])) if PAT in r["Kernel_Name"]]
    print("pass %s (%d launches, %.1f us each):" % (d, len(us), sum(us) / max(1, len(us))), {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $O/[a-f]
