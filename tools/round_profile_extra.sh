O=/root/repo/gpurun_out/final2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1024; do
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
done
SLU_LOOKAHEAD=0 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_seq -o s -- python /root/repo/bench.py --no-cpu-baseline --no-large-batch --steps 96 --warmup 32 > $O/bench_seq_prof.json 2>/dev/null
ls $O
