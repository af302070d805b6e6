echo "== before (HEAD)"; SEQS=1280 SLU_HIP_LIB=end-to-end-slu_amd/lib_alt_old/libslu_hip.so python tools/bf16_bench.py wconv 2>&1 | grep -v amdgpu
echo "== packed pair split in the staging loop"; SEQS=1280 python tools/bf16_bench.py wconv 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bf16.py tests/test_hip_pcm16.py -x -q 2>&1 | tail -2
