echo "== in-place epilogue (SLU_GRU_PIPE_EPI=0)"; SLU_HIP_LIB=end-to-end-slu_amd/lib_alt_g0/libslu_hip.so python tools/gru_two_tile_probe.py 1280 2>&1 | grep -v amdgpu
echo "== epilogue of step s-1 sliced between the MFMAs of step s"; python tools/gru_two_tile_probe.py 1280 2>&1 | grep -v amdgpu
timeout 300 python -m pytest tests/test_hip_bf16.py -x -q -k "two_tiles or epilogue or exact_fp32_kernel or fused_input" 2>&1 | tail -2
