#!/usr/bin/env python3
"""Times the kernels of the trainable suffix of the pipelined step (intent layer, B = 64, T = 19) ALONE on a
training-stream CU partition (20 launches back to back in one hipGraph): what each costs without the look-ahead
partition's traffic beside it.   python tools/suffix_bench.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops, pipeline

dev = torch.device("cuda", 0)
st = pipeline.cu_range_stream(dev, 0, pipeline.cu_split())      # the training stream's CU partition
T, B, I, H, D, V = 19, 64, 256, 128, 2, 24
x = torch.randn(T * B, I, device=dev)
w_ih = torch.randn(D * 3 * H, I, device=dev) * 0.05
b_ih = torch.randn(D * 3 * H, device=dev)
gx = torch.randn(T, B, D * 3 * H, device=dev)
wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
bf, br = torch.randn(3 * H, device=dev), torch.randn(3 * H, device=dev)
print("input projection gemm (1216 x 768 x 256): %.1f us" % (1e3 * _timed_graph(lambda: ops.gemm(x, w_ih.t(), b_ih), st)))
print("gru_seq_fwd (4-seq, reserve): %.1f us" % (1e3 * _timed_graph(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True), st)))
out, rsv = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True)
d_out = torch.randn(T, B, D * H, device=dev)
print("gru_seq_bwd (4-seq): %.1f us" % (1e3 * _timed_graph(lambda: ops.gru_seq_bwd(d_out, rsv, wf, wr, T, B, H, D), st)))
d_gx, d_gh, dbp = ops.gru_seq_bwd(d_out, rsv, wf, wr, T, B, H, D)
g2, h2, r2 = d_gx.view(T * B, -1), d_gh.view(T * B, -1), out.view(T * B, -1)
n = (T - 1) * B
dW = torch.empty(D * 3 * H, I, device=dev)
dWf, dWr = torch.empty(3 * H, H, device=dev), torch.empty(3 * H, H, device=dev)
probs = [(g2, x, dW), (h2[B:, :3 * H], r2[:n, :H], dWf), (h2[:n, 3 * H:], r2[B:, H:], dWr)]
print("gemm_tn_batched (3 weight gradients): %.1f us" % (1e3 * _timed_graph(lambda: ops.gemm_tn_batched(probs), st)))
db = torch.empty(dbp.shape[1:], device=dev)
print("gemm_tn_batched (3 weight gradients + bias-gradient row sums): %.1f us"
      % (1e3 * _timed_graph(lambda: ops.gemm_tn_batched(probs, (dbp, db)), st)))
for wg in (0, 96, 144, 192, 216, 288):
    with torch.cuda.stream(st):
        ops.gemm_tn_batched_splitk(probs, (dbp, db), max_wg=wg)     # (ticket words of this stream)
    torch.cuda.synchronize()
    print("gemm_tn_batched_splitk (3 weight gradients + bias row sums), workgroup budget %d: %.1f us"
          % (wg, 1e3 * _timed_graph(lambda: ops.gemm_tn_batched_splitk(probs, (dbp, db), max_wg=wg), st)))
def old():
    ops.gemm(g2.t(), x, out=dW); ops.gemm(probs[1][0].t(), probs[1][1], out=dWf); ops.gemm(probs[2][0].t(), probs[2][1], out=dWr)
print("three split-K gemms + reduces (round-1 path): %.1f us" % (1e3 * _timed_graph(old, st)))
h = torch.randn(T, B, D * H, device=dev)
cw, cb = torch.randn(V, D * H, device=dev) * 0.05, torch.randn(V, device=dev)
y = torch.stack([torch.randint(0, k, (B,)) for k in (6, 14, 4)], 1).to(dev)
print("head fwd (+reduce): %.1f us" % (1e3 * _timed_graph(lambda: ops.cls_maxpool_ce_fwd(h, cw, cb, y, (6, 14, 4), True), st)))
hw = torch.randn(V, D * H, device=dev, requires_grad=True)
_, logits, pred, argmax_t, d_logits = ops.cls_maxpool_ce_fwd(h, cw, cb, y, (6, 14, 4), True)
one = torch.ones((), device=dev)
import ctypes
from slu_hip import lib as _lib
L = _lib.load()
dh, dWc, dbc = torch.empty_like(h), torch.empty_like(cw), torch.empty_like(cb)
def head_bwd():
    _lib.check(L.slu_cls_maxpool_ce_bwd(d_logits.data_ptr(), argmax_t.data_ptr(), h.data_ptr(), cw.data_ptr(), one.data_ptr(),
                                        dh.data_ptr(), dWc.data_ptr(), dbc.data_ptr(), 0.0, 0, 0, None, T, B, D * H, V, st.cuda_stream), "bwd")
print("head bwd (d_h + d_W in one launch): %.1f us" % (1e3 * _timed_graph(head_bwd, st)))
print("dropout_pool fwd: %.1f us" % (1e3 * _timed_graph(lambda: ops.dropout_pool_fwd(out, None, 0.5, 1, 16, "none", 1), st)))
drop = (0.5, 1, 16, None)
print("head fwd with the dropout fused: %.1f us" % (1e3 * _timed_graph(lambda: ops.cls_maxpool_ce_fwd(h, cw, cb, y, (6, 14, 4), True, None, drop), st)))
def head_bwd_drop():
    _lib.check(L.slu_cls_maxpool_ce_bwd(d_logits.data_ptr(), argmax_t.data_ptr(), h.data_ptr(), cw.data_ptr(), one.data_ptr(),
                                        dh.data_ptr(), dWc.data_ptr(), dbc.data_ptr(), 0.5, 1, 16, None, T, B, D * H, V, st.cuda_stream), "bwd")
print("head bwd with the dropout fused: %.1f us" % (1e3 * _timed_graph(head_bwd_drop, st)))
