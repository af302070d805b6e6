#!/usr/bin/env python3
"""Round 6: the split-precision input-projection GEMMs of a look-ahead super-batch (the four frozen GRU layers' shapes) on
the look-ahead partition: launch times, issued-MFMA rate, and a checksum of the output (kernel variants must agree bit for
bit).    python tools/gemm_bf_probe.py [sequences = 1280] [nsplit = 3]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops, pipeline

dev = torch.device("cuda", 0)
n = pipeline.cu_split()
ncu = pipeline.n_compute_units(dev)
st = pipeline.cu_range_stream(dev, n, ncu - n)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mult = {1: 1, 2: 3, 3: 6}[ns]
tot = 0.0
for T, K in ((300, 60), (150, 256), (75, 256), (38, 256)):
    M, N = T * B, 768
    torch.manual_seed(T)
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; bias = torch.randn(N, device=dev)
    pl = ops.split_bf16(a, ns); pk = ops.gemm_bf16_pack(w, ns)
    out = torch.empty(M, N, device=dev)
    with torch.cuda.stream(st):
        ops.gemm_bf16(pl, pk, bias, N, K, out=out)
    st.synchronize()
    digest = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    ms = _timed_graph(lambda: ops.gemm_bf16(pl, pk, bias, N, K, out=out), st)
    tot += ms
    fl = 2.0 * M * N * ops.round_up(K, 32)
    print("M=%d N=%d K=%d nsplit %d on %d CUs: %.1f us | %.1f TFLOP/s fp32-equivalent | issued %.0f TFLOP/s = %.3f of the partition's bf16 peak | out %s"
          % (M, N, K, ns, ncu - n, 1e3 * ms, 2.0 * M * N * K / ms / 1e9, mult * fl / ms / 1e9, mult * fl / ms / 1e9 / (2500.0 * (ncu - n) / ncu), digest), flush=True)
print("sum of the four projections: %.1f us" % (1e3 * tot))
