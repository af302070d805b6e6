#!/usr/bin/env python3
"""The three frozen CNN blocks of a 16-batch super-batch (1024 x 3 s) on the look-ahead partition: split-precision
convolution kernels (f16x2 / bf16x3) against the exact fp32 kernel, and launch times.  python tools/wconv_probe.py"""
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
B = int(os.environ.get("PROBE_B", "1024"))
torch.manual_seed(3)
# (name, l_in, c_in, c_out, k, stride, do_abs, pool, time_major)
stages = [("sinc", 48000, 1, 80, 401, 80, True, 2, False), ("conv1", 300, 80, 60, 5, 1, False, 1, False),
          ("conv2", 300, 60, 60, 5, 1, False, 1, True)]
def r3_conv(x, w, bias, B, L, C, stride, do_abs, pool, tm, ns=2):
    """The round-3 kernel (tools/probes/slu_wconv_bf16_r3.hip) when the loaded library carries it: same-box baseline."""
    import ctypes
    from slu_hip import lib as _lib
    Lb = _lib.load()
    if not hasattr(Lb, "slu_wconv_fwd_bf16_r3"):
        return None
    vp, i64, ci, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    fn = Lb.slu_wconv_fwd_bf16_r3
    fn.restype = ci
    fn.argtypes = [vp, vp, i64, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, ci, ci, f32, i64, i64, vp, i64, vp, ctypes.c_size_t, ci, ci, vp]
    wsf = Lb.slu_wconv_bf16_workspace_bytes_r3
    wsf.restype = ctypes.c_size_t
    wsf.argtypes = [i64, i64, i64, ci]
    Co, _, k = w.shape
    l_conv = ops.conv_out_len(L, k, stride)
    l_out = -(-l_conv // pool)
    out = torch.empty((l_out, B, Co) if tm else (B, l_out, Co), device=dev)
    sb, sl = (Co, B * Co) if tm else (l_out * Co, Co)
    nb = wsf(Co, C, k, ns)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    rc = fn(x.data_ptr(), None, 0, w.data_ptr(), None if bias is None else bias.data_ptr(), out.data_ptr(), None, B, L, C, Co, k, stride,
            int(do_abs), pool, 0.2, sb, sl, None, 0, ws.data_ptr(), nb, 0, ns, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out


for name, L, C, Co, k, stride, do_abs, pool, tm in stages:
    x = torch.randn(B, L, C, device=dev) * 0.1
    w = torch.randn(Co, C, k, device=dev) / (C * k) ** 0.5
    bias = None if name == "sinc" else torch.randn(Co, device=dev) * 0.05
    ref = ops.wconv_fwd(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, tm, False)
    ref = ref[0] if isinstance(ref, tuple) else ref
    span = ref.abs().max().item()
    line = "%s B=%d: " % (name, B)
    for ns in (2, 3):
        out = ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, tm, ns)
        err = (out - ref).abs().max().item()
        t = 1e3 * _timed_graph(lambda: ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, tm, ns), st)
        l_conv = ops.conv_out_len(L, k, stride)
        tf = 2.0 * B * l_conv * Co * k * C / (t * 1e-6) / 1e12
        line += "nsplit %d: %.1f us (%.1f TFLOP/s fp32-equivalent), max dev from exact fp32 %.2e of range %.2f | " % (ns, t, tf, err, span)
    if r3_conv(x, w, bias, B, L, C, stride, do_abs, pool, tm) is not None:
        t = 1e3 * _timed_graph(lambda: r3_conv(x, w, bias, B, L, C, stride, do_abs, pool, tm), st)
        o3 = r3_conv(x, w, bias, B, L, C, stride, do_abs, pool, tm)
        line += "ROUND-3 kernel f16x2, this box: %.1f us (incl. its per-call filter pack), dev %.2e | " % (t, (o3 - ref).abs().max().item())
    if name == "conv2":
        pl = ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, True, 2, out_planes=True)
        t = 1e3 * _timed_graph(lambda: ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, do_abs, pool, 0.2, True, 2, out_planes=True), st)
        line += "planes out %.1f us" % t
    print(line, flush=True)
print("(round 3 in-loop: sinc 364 us, conv1 / conv2 ~ 190 / 160 us at 1024 sequences on 128 CUs)")
