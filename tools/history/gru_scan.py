#!/usr/bin/env python3
"""GRU recurrence time per step vs batch size (tuning aid for the 4- vs 16-sequence tile choice;
   SLU_GRU_TILE=4|16 forces a variant)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from slu_hip import ops
from bench_kernels import timeit

H, D, T = 128, 2, 75
wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
for B in [int(a) for a in sys.argv[1:]] or [16, 64, 256, 512, 1024, 2048, 4096]:
    gx = torch.randn(T, B, 2 * 3 * H, device="cuda")
    out, rsv = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True)
    d_out = torch.randn_like(out)
    f, _ = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, False))
    fr, _ = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True))
    b, _ = timeit(lambda: ops.gru_seq_bwd(d_out, rsv, wf, wr, T, B, H, D))
    fl = 2.0 * B * H * 3 * H * D * T
    print("B=%5d: fwd %7.2f us/step (%5.1f TF) | fwd+reserve %7.2f | bwd %7.2f us/step (%5.1f TF)"
          % (B, f / T, fl / f / 1e6, fr / T, b / T, fl / b / 1e6))
