"""The exact-fp32 GEMM on the projection / data-gradient shapes of a fully trainable B = 64 x 3 s step (whole chip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops
st = torch.cuda.Stream()
for T, I in ((300, 60), (150, 256), (75, 256), (38, 256), (19, 256)):
    M, N = T * 64, 768
    x = torch.randn(M, I, device="cuda"); w = torch.randn(N, I, device="cuda") * 0.1; b = torch.randn(N, device="cuda")
    g = torch.randn(M, N, device="cuda")
    fwd = _timed_graph(lambda: ops.gemm(x, w.t(), b), st)
    dx = _timed_graph(lambda: ops.gemm(g, w), st)                  # (M, 768) @ (768, I): B = W_ih as stored (n fast)
    wt = w.t().contiguous()
    dx_t = _timed_graph(lambda: ops.gemm(g, wt.t()), st)          # the same product with W_ih^T materialised (k fast)
    fl = 2.0 * M * N * I
    print("T=%d M=%d I=%d: projection x W^T %.1f us (%.1f TF) | dx = d_gx W: as stored %.1f us (%.1f TF), from a transposed copy %.1f us (%.1f TF)"
          % (T, M, I, 1e3 * fwd, fl / fwd / 1e9, 1e3 * dx, fl / dx / 1e9, 1e3 * dx_t, fl / dx_t / 1e9), flush=True)
