#!/usr/bin/env python3
"""Round-4 recurrence kernel (csrc/slu_gru_bf16.hip): every scheduling / arithmetic variant (SLU_GRU_VARIANT) against the
exact fp32 kernel, the fused Dropout + avg-pool epilogue against the two-launch path (bit equality), and launch times on
the look-ahead partition.  python tools/gru_variants.py [--quick]"""
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
H, D, ns = 128, 2, 2
VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "0,1,2,4").split(",")]


def r3_kernel(gx, wf, wr, bf, br, T, B, planes=None, I=0, packed=None, b_ih=None):
    """The round-3 kernel (tools/probes/slu_gru_bf16_r3.hip), when the loaded library carries it (tools/build_alt.sh with
    EXTRA_UNITS): same-box baseline."""
    import ctypes
    from slu_hip import lib as _lib
    L = _lib.load()
    if not hasattr(L, "slu_gru_seq_fwd_bf16_r3"):
        return None
    fn = L.slu_gru_seq_fwd_bf16_r3
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    fn.restype = ctypes.c_int
    fn.argtypes = [vp] * 8 + [i64, i64, vp, vp, i64, i64, i64, i64, ctypes.c_int, vp]
    out = torch.empty(T, B, D * H, device=dev)
    xa = (planes.data_ptr(), planes.stride(0), I, packed.data_ptr(), b_ih.data_ptr()) if planes is not None else (None, 0, 0, None, None)
    rc = fn(None if planes is not None else gx.data_ptr(), wf.data_ptr(), wr.data_ptr(), bf.data_ptr(), br.data_ptr(), out.data_ptr(),
            None, *xa, T, B, H, D, ns, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out


def weights(I, seed):
    torch.manual_seed(seed)
    w_ih = torch.randn(D * 3 * H, I, device=dev) * 0.1
    b_ih = torch.randn(D * 3 * H, device=dev) * 0.1
    wf, wr = torch.randn(3 * H, H, device=dev) * 0.08, torch.randn(3 * H, H, device=dev) * 0.08
    bf, br = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
    return w_ih, b_ih, wf, wr, bf, br


def check(T, B, I, sub):
    w_ih, b_ih, wf, wr, bf, br = weights(I, T * 1000 + B)
    x = torch.randn(T * B, I, device=dev)
    planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
    gx = ops.gemm_bf16(planes, packed, b_ih, D * 3 * H, I).view(T, B, D * 3 * H)
    ref, _ = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, False)            # exact fp32 MFMA kernel
    base = None
    for v in VARIANTS:
        os.environ["SLU_GRU_VARIANT"] = str(v)
        out = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns)[0]
        line = "T=%d B=%d I=%d var %2d: vs exact fp32 %.2e" % (T, B, I, v, (out - ref).abs().max().item())
        if base is None:
            base = out
        else:
            line += " | vs var 0 %.2e" % (out - base).abs().max().item()
        if ops.gru_fused_input_ok(I, H, D, ns):
            fo = ops.gru_seq_fwd_bf16(None, wf, wr, bf, br, T, B, H, D, ns, False, fused=(planes, I, packed, b_ih))[0]
            line += " | fused input == GEMM + recurrence: %s (%.1e)" % (torch.equal(fo, out), (fo - out).abs().max().item())
        # Dropout(0.5) + avg-pool(2): fused epilogue against recurrence + dropout_pool launch
        for p in (0.5, 0.0):
            keep = ops.dropout_bits(T, B, D * H, p, 1234, 7 * 16 + 3, None, sub, dev) if p > 0 else None
            two_f = ops.dropout_pool_fwd(out, None, p, 1234, 7 * 16 + 3, "avg", 2, None, sub, keep_bits=keep)
            two_p = ops.dropout_pool_fwd_planes(out, None, p, 1234, 7 * 16 + 3, "avg", 2, ns, None, sub, keep_bits=keep).planes
            one_f = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, p, False)
            one_p = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, p, True).planes
            line += " | pool p=%.1f fp32 %s planes %s" % (p, torch.equal(one_f, two_f), torch.equal(one_p.view(torch.int16), two_p.view(torch.int16)))
            if not torch.equal(one_p.view(torch.int16), two_p.view(torch.int16)):
                bad = (one_p.view(torch.int16) != two_p.view(torch.int16))
                idx = bad.nonzero()[:4]
                line += " (planes: %d of %d differ, e.g. %s -> fused %s vs two-launch %s)" % (
                    int(bad.sum()), bad.numel(), idx.tolist(), [hex(int(one_p.view(torch.int16)[tuple(i)]) & 0xffff) for i in idx],
                    [hex(int(two_p.view(torch.int16)[tuple(i)]) & 0xffff) for i in idx])
            if not torch.equal(one_f, two_f):
                d = (one_f - two_f).abs()
                line += " (max %.2e, %d of %d differ)" % (d.max().item(), int((d > 0).sum()), d.numel())
        print(line, flush=True)
    os.environ["SLU_GRU_VARIANT"] = "0"
    out3 = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, 3)[0]
    print("T=%d B=%d bf16x3 (transposed kernel) vs exact fp32 %.2e" % (T, B, (out3 - ref).abs().max().item()), flush=True)


def timing(T, B, I):
    w_ih, b_ih, wf, wr, bf, br = weights(I, 7)
    x = torch.randn(T * B, I, device=dev)
    planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
    gx = torch.randn(T, B, D * 3 * H, device=dev)
    keep = ops.dropout_bits(T, B, D * H, 0.5, 1234, 19, None, 64, dev)
    raw = torch.randn(T, B, D * H, device=dev)
    t_bits = 1e3 * _timed_graph(lambda: ops.dropout_bits(T, B, D * H, 0.5, 1234, 19, None, 64, dev), st)
    t_pool = 1e3 * _timed_graph(lambda: ops.dropout_pool_fwd_planes(raw, None, 0.5, 1234, 19, "avg", 2, ns, None, 64), st)
    print("T=%d B=%d on %d CUs: dropout_bits %.1f us; dropout_pool_fwd_planes (two-launch path) %.1f us" %
          (T, B, pipeline.n_compute_units(dev) - n, t_bits, t_pool), flush=True)
    fus = (planes, I, packed, b_ih) if ops.gru_fused_input_ok(I, H, D, ns) else None
    if r3_kernel(gx, wf, wr, bf, br, T, B) is not None:
        t0 = 1e3 * _timed_graph(lambda: r3_kernel(gx, wf, wr, bf, br, T, B), st)
        line = "  ROUND-3 kernel, this box: plain %.1f us (%.3f us/step)" % (t0, t0 / T)
        if fus is not None:
            t2 = 1e3 * _timed_graph(lambda: r3_kernel(None, wf, wr, bf, br, T, B, planes, I, packed, b_ih), st)
            line += " | fused input %.1f us (%.3f us/step)" % (t2, t2 / T)
        print(line, flush=True)
    for v in VARIANTS:
        os.environ["SLU_GRU_VARIANT"] = str(v)
        t0 = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns), st)
        t1 = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True), st)
        line = "  var %2d: plain %.1f us (%.3f us/step) | + dropout/pool epilogue (planes) %.1f us" % (v, t0, t0 / T, t1)
        if fus is not None:
            t2 = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_bf16(None, wf, wr, bf, br, T, B, H, D, ns, False, fused=fus), st)
            t3 = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_pool_bf16(None, wf, wr, bf, br, T, B, H, D, ns, keep, 0.5, True, fused=fus), st)
            line += " | fused input %.1f us (%.3f us/step) | fused input + epilogue %.1f us" % (t2, t2 / T, t3)
        print(line, flush=True)
    os.environ["SLU_GRU_VARIANT"] = "0"
    t3 = 1e3 * _timed_graph(lambda: ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, 3), st)
    print("  bf16x3 plain %.1f us (%.3f us/step)" % (t3, t3 / T), flush=True)


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    check(75, 128, 60, 64)          # odd T (partial last pooling window), fused input
    check(38, 37, 256, 0)           # ragged tile, K = 256
    if not quick:
        check(300, 1024, 60, 64)
    timing(300, 1024, 60)
    timing(150, 1024, 256)
    print("(round 3, same shapes on 128 CUs: plain T=300 385 us, fused input 511 us; T=150 ~ 190 us)")
