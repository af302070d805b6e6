#!/usr/bin/env python3
"""GEMM kernel time vs K and grid size (tuning aid): separates fixed cost from per-k-tile cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from slu_hip import ops
from bench_kernels import timeit

for M, N in [(128, 128), (1216, 384), (1216, 768), (19200, 768)]:
    for K in (32, 64, 128, 256, 512, 1024):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        out = torch.empty(M, N, device="cuda")
        med, mn = timeit(lambda: ops.gemm(a, w.t(), None, out=out))
        print("M=%6d N=%4d K=%5d: %8.1f us  %6.1f TF" % (M, N, K, med, 2.0 * M * N * K / med / 1e6))
