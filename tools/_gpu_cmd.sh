mkdir -p gpurun_out/r03_l; cd /root/repo
bash tools/cu_split_sweep.sh 2>&1 | tee gpurun_out/r03_l/sweep2.txt
