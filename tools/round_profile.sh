set -x
O=/root/repo/gpurun_out/final; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3 > $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_default -o d -- python /root/repo/bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_unfrozen -o u -- python /root/repo/bench.py --workload unfreeze_all --no-cpu-baseline --no-large-batch --steps 64 --warmup 48 > $O/bench_unfrozen_prof.json 2> $O/bench_unfrozen.err
cd /root/repo
timeout 400 python bench.py > $O/bench_default_noprof.json 2>/dev/null
timeout 400 python bench.py --workload unfreeze_all --no-cpu-baseline --no-large-batch > $O/bench_unfrozen.json 2>/dev/null
SLU_LOOKAHEAD=0 timeout 400 python bench.py --no-cpu-baseline --no-large-batch --steps 128 > $O/bench_sequential.json 2>/dev/null
cd /tmp
for B in 768 64; do
timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_sq_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$B -o p -- python /root/repo/tools/run_one.py gru $B > /dev/null 2>&1
done
cd /root/repo
timeout 400 python tools/bench_kernels.py > $O/microbench_b64.txt 2>&1
timeout 400 python tools/bench_kernels.py gemm gru wconv pool --batch 768 > $O/microbench_b768.txt 2>&1
timeout 200 python tools/gru_scan.py > $O/gru_scan.txt 2>&1
ls -R $O | head -60
