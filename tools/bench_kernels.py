#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the shapes of the B=64 x 3 s workload
(torch.cuda events on the launch stream; median of N runs).  Used for A/B tuning on the GPU box:

    python tools/bench_kernels.py [gemm] [gru] [wconv] [pool] [--batch 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))

import torch  # noqa: E402
from slu_hip import ops  # noqa: E402


def timeit(fn, n=10, warm=2, reps=10):
    """median / min microseconds per call; `reps` calls are captured into one hipGraph so that the
    Python/ctypes launch path (tens of microseconds) does not pollute short kernels."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["gemm", "gru", "wconv", "pool", "head"])
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    B = a.batch
    dev = "cuda"
    H = 128
    layers = [("phone0", 300, 60), ("phone1", 150, 256), ("word0", 75, 256), ("word1", 38, 256), ("intent", 19, 256)]
    if "gemm" in a.what:
        for name, T, I in layers:
            x = torch.randn(T * B, I, device=dev)
            w = torch.randn(3 * H, I, device=dev)
            w2 = torch.randn(6 * H, I, device=dev)
            b = torch.randn(3 * H, device=dev)
            b2 = torch.randn(6 * H, device=dev)
            gx = torch.empty(T * B, 6 * H, device=dev)
            med, mn = timeit(lambda: ops.gemm(x, w.t(), b, out=gx[:, :3 * H]))
            fl = 2.0 * T * B * 3 * H * I
            med2, mn2 = timeit(lambda: ops.gemm(x, w2.t(), b2, out=gx))
            print("gemm inproj %-7s M=%6d N=384 K=%3d: %7.1f us (min %6.1f) %6.1f TF | N=768 one launch: %7.1f us %6.1f TF"
                  % (name, T * B, I, med, mn, fl / med / 1e6, med2, 2 * fl / med2 / 1e6))
            dg = torch.randn(T * B, 6 * H, device=dev)
            med3, _ = timeit(lambda: ops.gemm(dg[:, :3 * H].t(), x))
            med4, _ = timeit(lambda: ops.gemm(dg[:, :3 * H], w))
            print("     bwd dW_ih (384x%d, K=%d): %7.1f us %6.1f TF | dX (M=%d,N=%d,K=384): %7.1f us %6.1f TF"
                  % (I, T * B, med3, fl / med3 / 1e6, T * B, I, med4, fl / med4 / 1e6))
    if "gru" in a.what:
        for name, T, I in layers:
            gx = torch.randn(T, B, 6 * H, device=dev)
            wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
            bf, br = torch.randn(3 * H, device=dev), torch.randn(3 * H, device=dev)
            med, mn = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, 2, False))
            medr, _ = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, 2, True))
            out, rsv = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, 2, True)
            dout = torch.randn_like(out)
            medb, _ = timeit(lambda: ops.gru_seq_bwd(dout, rsv, wf, wr, T, B, H, 2))
            print("gru %-7s T=%3d: fwd %7.1f us = %5.2f us/step (min %6.1f) | fwd+reserve %7.1f us = %5.2f | bwd %7.1f us = %5.2f us/step"
                  % (name, T, med, med / T, mn, medr, medr / T, medb, medb / T))
    if "wconv" in a.what:
        x = 0.1 * torch.randn(B, 48000, device=dev)
        f = torch.randn(80, 1, 401, device=dev) * 0.05
        med, mn = timeit(lambda: ops.wconv_fwd(x, f, None, B, 48000, 1, 80, True, 2, 0.2, False, False))
        print("wconv sinc  : %7.1f us (min %6.1f) %6.1f TF" % (med, mn, 2.0 * B * 600 * 80 * 401 / med / 1e6))
        h = torch.randn(B, 300, 80, device=dev)
        w1, b1 = torch.randn(60, 80, 5, device=dev) * 0.05, torch.randn(60, device=dev)
        med, mn = timeit(lambda: ops.wconv_fwd(h, w1, b1, B, 300, 80, 1, False, 1, 0.2, False, False))
        print("wconv conv1 : %7.1f us (min %6.1f) %6.1f TF" % (med, mn, 2.0 * B * 300 * 60 * 400 / med / 1e6))
        h2 = torch.randn(B, 300, 60, device=dev)
        w2 = torch.randn(60, 60, 5, device=dev) * 0.05
        med, mn = timeit(lambda: ops.wconv_fwd(h2, w2, b1, B, 300, 60, 1, False, 1, 0.2, True, False))
        print("wconv conv2 : %7.1f us (min %6.1f) %6.1f TF" % (med, mn, 2.0 * B * 300 * 60 * 300 / med / 1e6))
        dc = torch.randn(B, 300, 60, device=dev)
        med, _ = timeit(lambda: ops.wconv_bwd_weight(dc, h, B, 300, 80, 60, 5, 1, True))
        medd, _ = timeit(lambda: ops.wconv_bwd_data(dc, w1, B, 300))
        print("wconv conv1 bwd: dW %7.1f us | dX %7.1f us" % (med, medd))
        dcs = torch.randn(B, 600, 80, device=dev)
        med, _ = timeit(lambda: ops.wconv_bwd_weight(dcs, x, B, 48000, 1, 80, 401, 80, False))
        print("wconv sinc bwd: dW %7.1f us" % med)
    if "pool" in a.what:
        for name, T, I in layers[:2]:
            x = torch.randn(T, B, 256, device=dev)
            med, _ = timeit(lambda: ops.dropout_pool_fwd(x, None, 0.5, 1, 2, "avg", 2))
            print("dropout+avgpool %-7s: %6.1f us (%.0f GB/s)" % (name, med, 1.5 * x.numel() * 4 / med / 1e3))


    if "head" in a.what:
        T, C, vps = 19, 256, (6, 14, 4)
        h = torch.randn(T, B, C, device=dev)
        W = torch.randn(24, C, device=dev) * 0.06
        bias = torch.randn(24, device=dev)
        y = torch.stack([torch.randint(0, n, (B,), device=dev) for n in vps], dim=1)
        med, mn = timeit(lambda: ops.cls_maxpool_ce_fwd(h, W, bias, y, vps, True))
        la, lg, pr, am, dl = ops.cls_maxpool_ce_fwd(h, W, bias, y, vps, True)
        g = torch.ones((), device=dev)
        from slu_hip import lib as _l
        L = _l.load()
        dh, dW, db = torch.empty_like(h), torch.empty_like(W), torch.empty_like(bias)
        s_ = lambda: torch.cuda.current_stream().cuda_stream
        medb, _ = timeit(lambda: L.slu_cls_maxpool_ce_bwd(dl.data_ptr(), am.data_ptr(), h.data_ptr(), W.data_ptr(), g.data_ptr(),
                                                          dh.data_ptr(), dW.data_ptr(), db.data_ptr(), 0.0, 0, 0, None, T, B, C, 24, s_()))
        print("head fwd (+reduce): %6.1f us (min %5.1f) | bwd (dh + dW): %6.1f us" % (med, mn, medb))


if __name__ == "__main__":
    main()
