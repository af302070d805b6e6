#!/usr/bin/env python3
"""Runs one kernel shape repeatedly (for rocprofv3 --pmc passes): python tools/run_one.py gemm|gemmbig|gru [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from slu_hip import ops
what = sys.argv[1]
if what == "gemm":
    M, N, K = 128, 128, 256
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(20):
        ops.gemm(a, w.t(), None, out=out)
elif what == "gemmbig":
    M, N, K = 19200, 768, 256
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(10):
        ops.gemm(a, w.t(), None, out=out)
elif what in ("gemm_ip0", "gemm_ip1"):
    # input projections of a 20-batch look-ahead super-batch (1280 sequences): phone_rnn0 (T=300, K=60), phone_rnn1 (T=150, K=256)
    M, N, K = (300 * 768, 768, 60) if what == "gemm_ip0" else (150 * 768, 768, 256)
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        ops.gemm(a, w.t(), bias, out=out)
elif what in ("gemm_bf_ip0", "gemm_bf_ip1"):
    # split-precision input projections of a 20-batch look-ahead super-batch (1280 sequences): phone_rnn0 (T=300, K=60),
    # phone_rnn1 (T=150, K=256); three bf16 planes in, fp32 out
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
    NS = int(os.environ.get("RUN_ONE_NSPLIT", "3"))           # 2: the f16x2 scheme of the default path
    M, N, K = (300 * S, 768, 60) if what == "gemm_bf_ip0" else (150 * S, 768, 256)
    a = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.1; bias = torch.randn(N, device="cuda")
    planes = ops.split_bf16(a, NS); packed = ops.gemm_bf16_pack(w, NS); out = torch.empty(M, N, device="cuda")
    for _ in range(5):
        ops.gemm_bf16(planes, packed, bias, N, K, out=out)
elif what == "gru_bf":
    T, B, H = 300, (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 128
    NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    gx = torch.randn(T, B, 6 * H, device="cuda")
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    for _ in range(5):
        ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, 2, NS)
elif what in ("gru_bf3_pool_300_t2", "gru_bf3_pool_150_t2"):
    # round 6: the same product form on gru_bf2_fwd_kernel (two sequence tiles per workgroup), 40-batch super-batch by default
    T, B, H = (300 if "300" in what else 150), (int(sys.argv[2]) if len(sys.argv) > 2 else 2560), 128
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    keep = ops.dropout_bits(T, B, 2 * H, 0.5, 1234, 19, None, 64, "cuda")
    gx = torch.randn(T, B, 6 * H, device="cuda")
    for _ in range(5):
        ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, 2, 3, keep, 0.5, True, seq_tiles=2)
    torch.cuda.synchronize()
elif what == "gru_frozen_layers":
    # the dominant kernel of the default loop in its product form, all four frozen layers of a super-batch (bench.py's live
    # PMC passes): gx in, recurrence + Dropout(0.5) + avg-pool(2) epilogue, planes out (last layer: fp32 out); 3 rounds
    B, H = (int(sys.argv[2]) if len(sys.argv) > 2 else 1280), 128
    NS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    layers = []
    for T, planes in ((300, True), (150, True), (75, True), (38, False)):
        layers.append((T, planes, torch.randn(T, B, 6 * H, device="cuda"), ops.dropout_bits(T, B, 2 * H, 0.5, 1234, 19, None, 64, "cuda")))
    for _ in range(3):
        for T, planes, gx, keep in layers:
            ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, 2, NS, keep, 0.5, planes)
    torch.cuda.synchronize()
elif what in ("gru_bf3_pool_300", "gru_bf3_pool_150"):
    # round 5, the DEFAULT arithmetic (bf16x3): a frozen GRU layer as the look-ahead super-batch launches it — gx in (the
    # projection GEMM wrote it), recurrence + Dropout(0.5) + avg-pool(2) epilogue, three bf16 planes out; T = 300 / 150
    T, B, H = (300 if what.endswith("300") else 150), (int(sys.argv[2]) if len(sys.argv) > 2 else 1280), 128
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    keep = ops.dropout_bits(T, B, 2 * H, 0.5, 1234, 19, None, 64, "cuda")
    gx = torch.randn(T, B, 6 * H, device="cuda")
    for _ in range(5):
        ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, 2, 3, keep, 0.5, True)
    torch.cuda.synchronize()
elif what in ("gru_bf_pool_fused", "gru_bf_pool"):
    # the product form of a frozen GRU layer (round 4): recurrence + Dropout(0.5) + avg-pool(2) in one launch, plane output;
    # _fused: the first layer (K = 60, input projection inside the kernel, T = 300); else a K = 256 layer reading gx (T = 150)
    fused = what == "gru_bf_pool_fused"
    T, B, H, I = (300 if fused else 150), (int(sys.argv[2]) if len(sys.argv) > 2 else 1280), 128, 60
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    keep = ops.dropout_bits(T, B, 2 * H, 0.5, 1234, 19, None, 64, "cuda")
    if fused:
        x = torch.randn(T * B, I, device="cuda"); w_ih = torch.randn(6 * H, I, device="cuda") * 0.1; b_ih = torch.randn(6 * H, device="cuda") * 0.1
        planes, packed = ops.split_bf16(x, 2), ops.gemm_bf16_pack(w_ih, 2)
        for _ in range(5):
            ops.gru_seq_fwd_pool_bf16(None, wf, wr, bf, br, T, B, H, 2, 2, keep, 0.5, True, fused=(planes, I, packed, b_ih))
    else:
        gx = torch.randn(T, B, 6 * H, device="cuda")
        for _ in range(5):
            ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, 2, 2, keep, 0.5, True)
    torch.cuda.synchronize()
elif what in ("wconv_sinc", "wconv_conv1", "wconv_conv2"):
    # frozen CNN blocks of a super-batch on the split-precision kernel (RUN_ONE_NSPLIT: 2 = f16x2 (default here), 3 = bf16x3);
    # conv2 writes its time-major result as planes (what the first frozen GRU layer reads)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    NS = int(os.environ.get("RUN_ONE_NSPLIT", "2"))
    L, C, Co, k, stride, do_abs, pool = {"wconv_sinc": (48000, 1, 80, 401, 80, True, 2), "wconv_conv1": (300, 80, 60, 5, 1, False, 1),
                                         "wconv_conv2": (300, 60, 60, 5, 1, False, 1)}[what]
    x = torch.randn(B, L, C, device="cuda") * 0.1
    w = torch.randn(Co, C, k, device="cuda") / (C * k) ** 0.5
    bias = None if what == "wconv_sinc" else torch.randn(Co, device="cuda") * 0.1
    tm = what == "wconv_conv2"
    for _ in range(5):
        ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, tm, NS, out_planes=tm)
    torch.cuda.synchronize()
elif what == "gru_bf_fused":
    # the first frozen GRU layer of a super-batch (K = 60, T = 300): recurrence with the fused input projection
    T, B, H, I = 300, (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 128, 60
    x = torch.randn(T * B, I, device="cuda"); w_ih = torch.randn(6 * H, I, device="cuda") * 0.1; b_ih = torch.randn(6 * H, device="cuda") * 0.1
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    planes, packed = ops.split_bf16(x, 2), ops.gemm_bf16_pack(w_ih, 2)
    for _ in range(5):
        ops.gru_seq_fwd_bf16(None, wf, wr, bf, br, T, B, H, 2, 2, False, fused=(planes, I, packed, b_ih))
elif what == "tn":
    # the intent layer's three weight gradients in one batched launch (T = 19, B = 64)
    T, B, I, H, D = 19, 64, 256, 128, 2
    x = torch.randn(T * B, I, device="cuda"); g2 = torch.randn(T * B, D * 3 * H, device="cuda"); h2 = torch.randn(T * B, D * 3 * H, device="cuda")
    r2 = torch.randn(T * B, D * H, device="cuda"); n = (T - 1) * B
    dW = torch.empty(D * 3 * H, I, device="cuda"); dWf = torch.empty(3 * H, H, device="cuda"); dWr = torch.empty(3 * H, H, device="cuda")
    probs = [(g2, x, dW), (h2[B:, :3 * H], r2[:n, :H], dWf), (h2[:n, 3 * H:], r2[B:, H:], dWr)]
    for _ in range(10):
        ops.gemm_tn_batched(probs)
elif what == "tn_long":
    # a word-layer-sized GRU layer's three weight gradients in one split-K launch (T = 150, B = 64: K = 9600 rows)
    T, B, I, H, D = 150, 64, 256, 128, 2
    x = torch.randn(T * B, I, device="cuda"); g2 = torch.randn(T * B, D * 3 * H, device="cuda"); h2 = torch.randn(T * B, D * 3 * H, device="cuda")
    r2 = torch.randn(T * B, D * H, device="cuda"); n = (T - 1) * B
    dW = torch.empty(D * 3 * H, I, device="cuda"); dWf = torch.empty(3 * H, H, device="cuda"); dWr = torch.empty(3 * H, H, device="cuda")
    probs = [(g2, x, dW), (h2[B:, :3 * H], r2[:n, :H], dWf), (h2[:n, 3 * H:], r2[B:, H:], dWr)]
    for _ in range(10):
        ops.gemm_tn_batched_splitk(probs)
elif what == "gru":
    T, B, H = 300, (int(sys.argv[2]) if len(sys.argv) > 2 else 64), 128
    gx = torch.randn(T, B, 6 * H, device="cuda")
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    for _ in range(5):
        ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, 2, False)
torch.cuda.synchronize()
