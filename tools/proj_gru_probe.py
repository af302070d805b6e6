#!/usr/bin/env python3
"""Input projection + recurrence of the intent layer (T = 19, B = 64, K = 256, H = 128, both directions) on the training
partition: two launches (slu_gemm_f32, slu_gru_seq_fwd) against one (slu_gru_proj_seq_fwd), 20 back to back in a hipGraph.
python tools/proj_gru_probe.py [T B I]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
from bench import _timed_graph
from slu_hip import ops, pipeline

dev = torch.device("cuda", 0)
n = pipeline.cu_split()
st = pipeline.cu_range_stream(dev, 0, n)
T, B, I = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (19, 64, 256)
H, D = 128, 2
x = torch.randn(T * B, I, device=dev); w_ih = torch.randn(D * 3 * H, I, device=dev) * 0.1; b_ih = torch.randn(D * 3 * H, device=dev) * 0.1
wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
bf, br = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1


def two():
    gx = ops.gemm(x, w_ih.t(), b_ih)
    return ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True)


def one():
    return ops.gru_proj_seq_fwd(x, w_ih, b_ih, wf, wr, bf, br, T, B, I, H, D, True)


with torch.cuda.stream(st):
    a, b = two()[0], one()[0]
    st.synchronize()
print("T=%d B=%d I=%d on %d CUs: equal %s | two launches %.1f us | one launch %.1f us" % (
    T, B, I, n, torch.equal(a, b), 1e3 * _timed_graph(two, st), 1e3 * _timed_graph(one, st)))
