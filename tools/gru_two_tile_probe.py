#!/usr/bin/env python3
"""Round 6: gru_bf2_fwd_kernel (two sequence tiles per workgroup, seq_tiles = 2) against gru_bf_fwd_kernel (seq_tiles = 1) in
the product form of a frozen layer (gx in, Dropout(0.5) + avg-pool(2) epilogue, planes out) on the look-ahead partition:
launch times per shape and scheme, bit equality.    python tools/gru_two_tile_probe.py [B ...]"""
import os, sys
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
H, D = 128, 2
sizes = [int(v) for v in sys.argv[1:]] or [1280, 2560]
for ns in (3, 2):
    for T in (300, 150):
        for B in sizes:
            torch.manual_seed(1)
            gx = torch.randn(T, B, D * 3 * H, device=dev)
            wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
            bf, br = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
            keep = ops.dropout_bits(T, B, D * H, 0.5, 1234, 19, None, 64, dev)
            res = {}
            for tiles in (1, 2):
                with torch.cuda.stream(st):
                    out = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True, seq_tiles=tiles).planes
                st.synchronize()
                us = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True, seq_tiles=tiles), st)
                res[tiles] = (us, out)
            same = torch.equal(res[1][1].view(torch.int16), res[2][1].view(torch.int16))
            flops = 2.0 * B * H * 3 * H * D * T
            print("nsplit %d T=%d B=%d on %d CUs: one tile %.1f us (%.3f us/step, %.1f TFLOP/s) | two tiles %.1f us (%.3f us/step, %.1f TFLOP/s) | "
                  "speed-up %.2f | bit-identical %s" % (ns, T, B, ncu - n, res[1][0], res[1][0] / T, flops / res[1][0] / 1e6,
                                                        res[2][0], res[2][0] / T, flops / res[2][0] / 1e6, res[1][0] / res[2][0], same), flush=True)
            del gx, keep, res
