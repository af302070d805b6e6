#!/usr/bin/env python3
"""First frozen GRU layer (K = 60, T = 300) of a 16-batch super-batch on the look-ahead partition: projection GEMM +
recurrence against the recurrence with the fused input projection (bit-equality and time).  python tools/gru_fused_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops, pipeline

dev = torch.device("cuda", 0)
n = pipeline.cu_split()
st = pipeline.cu_range_stream(dev, n, pipeline.n_compute_units(dev) - n)
T, H, D, I, ns = 300, 128, 2, 60, 2
for B in (1024, 768, 37):
    torch.manual_seed(B)
    x = torch.randn(T * B, I, device=dev)
    w_ih = torch.randn(D * 3 * H, I, device=dev) * 0.1
    b_ih = torch.randn(D * 3 * H, device=dev) * 0.1
    wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
    bf, br = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
    planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
    def unfused():
        gx = ops.gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
        return ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns)[0]
    def fused():
        return ops.gru_seq_fwd_bf16(None, wf, wr, bf, br, T, B, H, D, ns, False, fused=(planes, I, packed, b_ih))[0]
    a, b = unfused(), fused()
    torch.cuda.synchronize()
    print("B=%d: fused == GEMM + recurrence bit for bit: %s (max abs diff %.3e)" % (B, torch.equal(a, b), (a - b).abs().max().item()))
    if B >= 768:
        tg = 1e3 * _timed_graph(lambda: ops.gemm_bf16(planes, packed, b_ih, D * 3 * H, I), st)
        gx = ops.gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
        tr = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns), st)
        tf = 1e3 * _timed_graph(fused, st)
        print("B=%d on %d CUs: projection GEMM %.1f us + recurrence %.1f us = %.1f us; fused %.1f us" % (B, pipeline.n_compute_units(dev) - n, tg, tr, tg + tr, tf))
