#!/usr/bin/env python3
"""GEMM kernel time vs K and grid size (tuning aid): separates fixed cost from per-k-tile cost.
   usage: gemm_scan.py [small|large]   (SLU_GEMM_THR=<tiles> switches the 64/128 tile choice)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from slu_hip import ops
from bench_kernels import timeit

which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    shapes = [(M, N, K) for M, N in [(128, 128), (1216, 384), (1216, 768), (19200, 768)]
              for K in (32, 64, 128, 256, 512, 1024)]
else:
    shapes = [(38400, 768, 256), (76800, 768, 256), (153600, 768, 256), (307200, 768, 256), (614400, 768, 60),
              (8192, 8192, 1024), (4096, 4096, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    out = torch.empty(M, N, device="cuda")
    med, mn = timeit(lambda: ops.gemm(a, w.t(), None, out=out))
    print("M=%6d N=%4d K=%5d: %8.1f us  %6.1f TF" % (M, N, K, med, 2.0 * M * N * K / med / 1e6))
