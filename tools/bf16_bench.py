#!/usr/bin/env python3
"""Times the split-precision kernels against their exact-fp32 counterparts at the launch shapes of a
12-batch look-ahead super-batch (768 sequences):  python tools/bf16_bench.py [gemm] [gru] [wconv]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops

what = sys.argv[1:] or ["gemm", "gru", "wconv"]
st = torch.cuda.Stream()
B = int(os.environ.get("SEQS", "768"))
if "gemm" in what:
    for T, K in ((300, 60), (150, 256), (75, 256), (38, 256)):
        M, N = T * B, 768
        a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        ms32 = _timed_graph(lambda: ops.gemm(a, w.t(), bias, out=out), st)
        line = "gemm M=%d N=%d K=%d: fp32 %.1f us (%.1f TF)" % (M, N, K, 1e3 * ms32, 2.0 * M * N * K / ms32 / 1e9)
        for ns in (3, 1):
            pl = ops.split_bf16(a, ns); pk = ops.gemm_bf16_pack(w, ns)
            ms = _timed_graph(lambda: ops.gemm_bf16(pl, pk, bias, N, K, out=out), st)
            byt = pl.numel() * 2 + out.numel() * 4
            line += " | nsplit=%d %.1f us (%.1f TF-equivalent, %.2f TB/s)" % (ns, 1e3 * ms, 2.0 * M * N * K / ms / 1e9, byt / ms / 1e9)
        print(line, flush=True)

if "gru" in what:
    H, D = 128, 2
    for T in (300, 150, 75, 38):
        gx = torch.randn(T, B, D * 3 * H, device="cuda")
        wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
        bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
        fl = 2.0 * B * H * 3 * H * D * T
        ms32 = _timed_graph(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, False), st)
        line = "gru T=%d B=%d: fp32 %.1f us (%.2f us/step, %.1f TF)" % (T, B, 1e3 * ms32, 1e3 * ms32 / T, fl / ms32 / 1e9)
        for ns in (3, 1):
            ms = _timed_graph(lambda: ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns), st)
            line += " | nsplit=%d %.1f us (%.2f us/step, %.1f TF-equivalent)" % (ns, 1e3 * ms, 1e3 * ms / T, fl / ms / 1e9)
        print(line, flush=True)

if "wconv" in what:
    # the three convolution blocks of the frozen encoder (split-precision kernel vs the exact-fp32 one)
    for name, L, C, Co, k, stride, do_abs, pool in (("sinc", 48000, 1, 80, 401, 80, True, 2), ("conv1", 300, 80, 60, 5, 1, False, 1),
                                                    ("conv2", 300, 60, 60, 5, 1, False, 1)):
        x = torch.randn(B, L, C, device="cuda") * 0.1
        w = torch.randn(Co, C, k, device="cuda") * 0.05
        bias = torch.randn(Co, device="cuda")
        l_conv = ops.conv_out_len(L, k, stride)
        fl = 2.0 * B * l_conv * Co * C * k
        line = "%s B=%d L=%d %d->%d k=%d:" % (name, B, L, C, Co, k)
        for ns in (3, 1):
            ms = _timed_graph(lambda: ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, False, ns), st)
            line += " nsplit=%d %.1f us (%.1f TF-equivalent) |" % (ns, 1e3 * ms, fl / ms / 1e9)
        print(line, flush=True)
