"""Tensor-level wrappers over the C ABI (include/slu_hip.h) and the autograd Functions built on them.

Layouts used between kernels (the host mirror in models.py converts at the module boundary):
  waveform (B, T); CNN activations channels-last (B, L, C); everything from the last CNN layer
  on is TIME-MAJOR (T, B, C), which is also the memory order ATen's batch_first GRU uses.

torch is plumbing here: it owns the device buffers, the stream and the autograd tape.  All
arithmetic of the hot path happens in libslu_hip.so; there is no fallback path.
"""
import contextlib
import os

import torch

from . import lib as _lib

METHODS = {"none": 0, "avg": 1, "max": 2}


def bf16_mode():
    """SLU_DTYPE=bf16 (BASELINE configs[4]): every forward contraction — convolutions, GRU input projections and
    recurrences — and the backward contractions of the GRU layers (data gradients, and the weight gradients of layers
    with more than TN_SMALL_ROWS rows on slu_gemm_tn_bf16) take bf16 operands on v_mfma_f32_16x16x32_bf16 with fp32
    accumulation; convolution weight gradients, BPTT gate math, loss math, master weights and Adam stay fp32."""
    return os.environ.get("SLU_DTYPE", "f32") == "bf16"


_TRAIN_MATH = {"split": (2, 3), "bf16x3": (3, 3), "fp32": (0, 0)}


def train_nsplit(grad=False):
    """Split scheme (csrc/slu_bf16.h) of the GEMM-shaped contractions of TRAINABLE layers — GRU input projections, their
    data and weight gradients, the convolution blocks' data gradients, the ASR heads' products (SLU_TRAIN_MATH):
      "fp32" (default): exact fp32 MFMA everywhere;
      "split": products of activations and weights (grad=False) on f16x2 — fp32 operands as two fp16 terms,
          three fp16 MFMA products, 3/16 of the fp32-MFMA cycles; products with a GRADIENT operand (grad=True: d_gx W,
          d_gx^T x, ...) on bf16x3 — three bf16 terms, six products, 6/16 — because gradient entries are far below
          fp16's smallest normal number (softmax gradients of a 10 000-word head: 1e-7), where f16x2 keeps 11 bits
          only; bf16 has fp32's exponent.  Both are fp32-class: 1e-7 of sum |a b| against float64 (tests/test_hip_bf16.py);
      "bf16x3": bf16x3 for both;
      SLU_DTYPE=bf16 (BASELINE configs[4]) overrides: plain bf16 operands (1).
    "split" measured on MI355X (B = 64, 3 s, same box): unfreeze_all 2.73 vs 2.79 ms/step, ASR pre-training 3.43 vs
    3.05 (the 10 000-word head's products do not suit the 128 x 64 tiles) — the weight-gradient GEMMs run beside the
    next layer's BPTT and the recurrences dominate the step, so it stays opt-in.
    Always exact fp32: the recurrences (forward and BPTT) — bf16 mode's forward recurrences excepted —, the FORWARD pass of
    trainable convolution blocks (|.| and LeakyReLU have kinks: the sign of a pre-activation within round-off of zero
    decides a whole gradient term, and the exact kernel keeps those decisions where the reference's are), the convolutions'
    weight gradients, every reduction, the loss and the optimizer."""
    if bf16_mode():
        return 1
    mode = os.environ.get("SLU_TRAIN_MATH", "fp32")
    if mode not in _TRAIN_MATH:
        raise ValueError("SLU_TRAIN_MATH=%r: expected one of %s" % (mode, sorted(_TRAIN_MATH)))
    return _TRAIN_MATH[mode][1 if grad else 0]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _f32c(t, what):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError("%s must be a float32 CUDA tensor (got %s on %s)" % (what, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


# Auxiliary streams for independent work inside one backward stage (the weight-gradient GEMMs of a GRU
# layer do not depend on each other nor on the data-gradient GEMM): forked from and joined back into
# the current stream.
_AUX = {}


def _aux_streams(device, n):
    key = (device.type, device.index)
    pool = _AUX.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


class _Fork:
    """with _Fork(device, i): ...   runs the body on auxiliary stream i after the current stream's work;
    _Fork.join(device) makes the current stream wait for every auxiliary stream used since."""
    _used = {}

    capture_forks = False     # set by pipeline.StepGraph while it captures a step with nothing running beside it
    # Round 5: set by training.Trainer._iterate_full_steps for the steps of a FULLY trainable loop (nothing runs beside them):
    # the batched weight-gradient launch of a GRU layer with thousands of rows goes to an auxiliary stream / graph branch
    # (ops.wgrad_branch: joined at the end of the layer's backward, or — mode "pass" — left open until the trainer's single
    # join after loss.backward()).
    defer = False

    def __init__(self, device, i):
        # Inside a captured step of the look-ahead pipeline the branches land on extra hardware queues that
        # compete with the look-ahead streams (measured: -17 % on the pipelined step): there forking is
        # eager-mode only.  A fully trainable step has the chip to itself: its captured graph keeps the
        # branches (weight-gradient GEMMs beside the next layer's BPTT, which fills 32 of 256 CUs).
        self.active = _Fork.capture_forks or not torch.cuda.is_current_stream_capturing()
        if self.active:
            self.cur = torch.cuda.current_stream(device)
            self.side = _aux_streams(device, i + 1)[i]
            self.ctx = torch.cuda.stream(self.side)
            _Fork._used.setdefault((device.type, device.index), set()).add(i)

    def __enter__(self):
        if self.active:
            self.side.wait_stream(self.cur)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc) if self.active else False

    @staticmethod
    def join(device):
        cur = torch.cuda.current_stream(device)
        key = (device.type, device.index)
        for i in sorted(_Fork._used.pop(key, ())):
            cur.wait_stream(_AUX[key][i])

    # Why the join sits at the end of each layer's backward function and not at the end of the whole backward pass
    # (round 4, tried: one join queued with autograd's queue_callback, branch operands kept alive until then, so that a
    # layer's weight-gradient GEMMs run beside the LOWER layer's BPTT, which fills 32 of 256 CUs for hundreds of
    # microseconds): (1) it is slower — fully trainable step 2.79 -> 3.04 ms, ASR pre-training 3.23 -> 3.52 ms
    # (profiles/r04_aa_fork_join.txt): full-chip GEMMs beside the latency-bound recurrence lengthen its dependent steps
    # by more than the GEMMs' own time (the likely cause; the measurement is the fact); (2) it is not safe under autograd as is: AccumulateGrad CLONES a gradient it
    # cannot steal (a view of a stacked buffer, or a tensor someone else still references) on the main stream, i.e.
    # before the branch has written it.


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# functional wrappers (no autograd)
# ------------------------------------------------------------------------------------------------

def sinc_filters(b1, band, filt_dim, fs):
    L = _lib.load()
    n = b1.numel()
    out = torch.empty(n, filt_dim, dtype=torch.float32, device=b1.device)
    _lib.check(L.slu_sinc_filters_fwd(b1.data_ptr(), band.data_ptr(), out.data_ptr(), n, filt_dim,
                                      float(fs), _stream()), "slu_sinc_filters_fwd")
    return out


def sinc_filters_bwd(b1, band, d_filters, filt_dim, fs):
    L = _lib.load()
    n = b1.numel()
    d_filters = _f32c(d_filters, "d_filters")
    db1 = torch.empty_like(b1)
    dband = torch.empty_like(band)
    _lib.check(L.slu_sinc_filters_bwd(b1.data_ptr(), band.data_ptr(), d_filters.data_ptr(),
                                      db1.data_ptr(), dband.data_ptr(), n, filt_dim, float(fs),
                                      _stream()), "slu_sinc_filters_bwd")
    return db1, dband


def conv_out_len(l_in, k_t, stride):
    return (l_in + 2 * (k_t // 2) - k_t) // stride + 1


def wconv_fwd(x, weight, bias, B, l_in, c_in, stride, do_abs, pool, slope, time_major, want_route):
    """x: contiguous (B, l_in, c_in) [or (B, T) with c_in = 1]; weight (c_out, c_in, k_t)."""
    L = _lib.load()
    x = _f32c(x, "x")
    weight = _f32c(weight, "weight")
    c_out, _, k_t = weight.shape
    l_conv = conv_out_len(l_in, k_t, stride)
    l_out = -(-l_conv // pool)
    if time_major:
        out = torch.empty(l_out, B, c_out, dtype=torch.float32, device=x.device)
        sb, sl = c_out, B * c_out
    else:
        out = torch.empty(B, l_out, c_out, dtype=torch.float32, device=x.device)
        sb, sl = l_out * c_out, c_out
    route = torch.empty(B, l_out, c_out, dtype=torch.uint8, device=x.device) if want_route else None
    wsb = L.slu_wconv_workspace_bytes(c_out, c_in, k_t)
    ws = _workspace(wsb, x.device)
    _lib.check(L.slu_wconv_fwd(x.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(),
                               _ptr(route), B, l_in, c_in, c_out, k_t, stride, int(do_abs), pool,
                               float(slope), sb, sl, ws.data_ptr(), wsb, _stream()), "slu_wconv_fwd")
    return out, route, l_conv


def wconv_bwd_act(dy, y, route, B, l_conv, c_out, do_abs, pool, slope, time_major):
    L = _lib.load()
    dy = _f32c(dy, "dy")
    l_out = -(-l_conv // pool)
    sb, sl = (c_out, B * c_out) if time_major else (l_out * c_out, c_out)
    d_conv = torch.empty(B, l_conv, c_out, dtype=torch.float32, device=dy.device)
    _lib.check(L.slu_wconv_bwd_act(dy.data_ptr(), y.data_ptr(), _ptr(route), d_conv.data_ptr(), B,
                                   l_conv, c_out, int(do_abs), pool, float(slope), sb, sl,
                                   _stream()), "slu_wconv_bwd_act")
    return d_conv


def wconv_bwd_data(d_conv, weight, B, l_in):
    L = _lib.load()
    c_out, c_in, k_t = weight.shape
    d_in = torch.empty(B, l_in, c_in, dtype=torch.float32, device=d_conv.device)
    wsb = L.slu_wconv_workspace_bytes(c_out, c_in, k_t)
    ws = _workspace(wsb, d_conv.device)
    _lib.check(L.slu_wconv_bwd_data(d_conv.data_ptr(), weight.data_ptr(), d_in.data_ptr(), B, l_in,
                                    c_in, c_out, k_t, ws.data_ptr(), wsb, _stream()),
               "slu_wconv_bwd_data")
    return d_in


def wconv_bwd_weight(d_conv, x, B, l_in, c_in, c_out, k_t, stride, want_bias, out=None):
    L = _lib.load()
    if out is not None:
        dW, db = out
    else:
        dW = torch.empty(c_out, c_in, k_t, dtype=torch.float32, device=x.device)
        db = torch.empty(c_out, dtype=torch.float32, device=x.device) if want_bias else None
    wsb = L.slu_wconv_bwd_weight_workspace_bytes(B, l_in, c_in, c_out, k_t, stride)
    ws = _workspace(wsb, x.device)
    _lib.check(L.slu_wconv_bwd_weight(d_conv.data_ptr(), x.data_ptr(), dW.data_ptr(), _ptr(db), B,
                                      l_in, c_in, c_out, k_t, stride, ws.data_ptr(), wsb, _stream()),
               "slu_wconv_bwd_weight")
    return dW, db


def gemm(a, b, bias=None, out=None, accumulate=False):
    """out(M,N) = [out] + a(M,K) @ b(K,N) + bias(N); a, b, out are 2-D views with ANY strides."""
    L = _lib.load()
    M, K = a.shape
    K2, N = b.shape
    assert K == K2, (a.shape, b.shape)
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    wsb = L.slu_gemm_workspace_bytes(M, N, K)
    ws = _workspace(wsb, a.device) if wsb else None
    _lib.check(L.slu_gemm_f32(a.data_ptr(), a.stride(0), a.stride(1), b.data_ptr(), b.stride(0),
                              b.stride(1), out.data_ptr(), out.stride(0), out.stride(1), _ptr(bias),
                              M, N, K, int(accumulate), _ptr(ws), wsb, _stream()), "slu_gemm_f32")
    return out


def _small_ok(a, b, mode):
    return (a.dtype == b.dtype == torch.float32 and a.dim() == b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
            and a.shape[1] % 4 == 0 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0
            and (mode == 1 or (b.stride(0) % 4 == 0 and b.data_ptr() % 16 == 0)))


def gemm_small_batched(problems):
    """problems: [(A (M, K), B, bias or None, C (M, N), mode, accumulate)] — mode 0: B is (N, K) (C = A B^T + bias), mode 1:
    B is (K, N) (C = A B + bias); row-strided fp32 views with unit column stride.  Up to four problems per launch of
    slu_gemm_small_batched (latency-bound small-M products); shapes the kernel does not take go through gemm()."""
    import ctypes
    L = _lib.load()
    batch = []

    def flush():
        if not batch:
            return
        n = len(batch)
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        arr = lambda ty, vals: (ty * n)(*vals)
        _lib.check(L.slu_gemm_small_batched(
            arr(vp, [p[0].data_ptr() for p in batch]), arr(i64, [p[0].stride(0) for p in batch]),
            arr(vp, [p[1].data_ptr() for p in batch]), arr(i64, [p[1].stride(0) for p in batch]),
            arr(ci, [p[4] for p in batch]), arr(vp, [p[3].data_ptr() for p in batch]), arr(i64, [p[3].stride(0) for p in batch]),
            arr(vp, [_ptr(p[2]) or None for p in batch]), arr(ci, [int(p[5]) for p in batch]),
            arr(i64, [p[0].shape[0] for p in batch]), arr(i64, [p[3].shape[1] for p in batch]), arr(i64, [p[0].shape[1] for p in batch]),
            n, _stream()), "slu_gemm_small_batched")
        batch.clear()

    for A, B, bias, C, mode, acc in problems:
        assert C.stride(1) == 1 and C.shape[0] == A.shape[0]
        assert (B.shape == (C.shape[1], A.shape[1])) if mode == 0 else (B.shape == (A.shape[1], C.shape[1]))
        if _small_ok(A, B, mode):
            batch.append((A, B, bias, C, mode, acc))
            if len(batch) == 4:
                flush()
        else:
            flush()
            gemm(A, B.t() if mode == 0 else B, bias, out=C, accumulate=bool(acc))
    flush()


def stage_inputs(pairs, set_tensor=None, set_value=0):
    """One launch refreshing static input buffers: pairs = [(dst contiguous, src)] with src contiguous or strided
    along dim 0 only; optionally set_tensor[0] = set_value (int64).  Returns the pairs it could not take."""
    import ctypes
    L = _lib.load()
    segs, rest = [], []
    for dst, src in pairs:
        ok = src.is_cuda and src.dtype == dst.dtype and tuple(src.shape) == tuple(dst.shape) and dst.is_contiguous()
        if ok and src.is_contiguous():
            segs.append((src.data_ptr(), dst.data_ptr(), 1, src.numel() * src.element_size(), 0))
        elif ok and src.dim() >= 2 and src[0].is_contiguous():
            row = src[0].numel() * src.element_size()
            segs.append((src.data_ptr(), dst.data_ptr(), src.shape[0], row, src.stride(0) * src.element_size()))
        else:
            rest.append((dst, src))
    while len(segs) > 4:
        rest.append(None)       # never happens for the step inputs (<= 3 tensors); guard for other callers
        segs.pop()
    n = len(segs)
    arr = lambda k, ty: (ty * max(n, 1))(*([sg[k] for sg in segs] or [0]))
    _lib.check(L.slu_stage_inputs(arr(0, ctypes.c_void_p), arr(1, ctypes.c_void_p), arr(2, ctypes.c_int64),
                                  arr(3, ctypes.c_int64), arr(4, ctypes.c_int64), n, _ptr(set_tensor), int(set_value),
                                  _stream()), "slu_stage_inputs")
    return [r for r in rest if r is not None]


def copy_multi(pairs):
    """[(dst, src)] contiguous same-sized tensors (whole 4-byte words): all copies in ceil(n / 32) launches."""
    import ctypes
    L = _lib.load()
    step = L.slu_multi_max()
    for i in range(0, len(pairs), step):
        part = pairs[i:i + step]
        n = len(part)
        for dst, src in part:
            assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == src.numel() * src.element_size()
        src = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in part])
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in part])
        nb = (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in part])
        _lib.check(L.slu_copy_multi(src, dst, nb, n, _stream()), "slu_copy_multi")


def scale_multi(tensors, scale_dev):
    """x *= scale_dev[0] for every contiguous fp32 tensor of the list (one launch per 32 tensors)."""
    import ctypes
    if not tensors:
        return
    L = _lib.load()
    scale_dev = scale_dev.reshape(-1)
    assert scale_dev.dtype == torch.float32 and scale_dev.is_cuda
    step = L.slu_multi_max()
    for i in range(0, len(tensors), step):
        part = tensors[i:i + step]
        n = len(part)
        assert all(t.dtype == torch.float32 and t.is_contiguous() for t in part)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in part])
        numel = (ctypes.c_int64 * n)(*[t.numel() for t in part])
        _lib.check(L.slu_scale_multi(ptrs, numel, n, scale_dev.data_ptr(), _stream()), "slu_scale_multi")


class PoolActFn(torch.autograd.Function):
    """[Abs ->] MaxPool1d(pool, ceil) -> LeakyReLU / ReLU for pool widths the convolution epilogue does not fuse
    (cnn_max_pool_len > 2; reference models.py:163-168, :205, :211-213).  x channels-last (B, L, C) ->
    (B, L_out, C), or time-major (L_out, B, C)."""

    @staticmethod
    def forward(ctx, x, pool, do_abs, slope, time_major):
        L = _lib.load()
        x = _f32c(x, "x")
        B, Lin, C = x.shape
        l_out = -(-Lin // pool)
        if time_major:
            y = torch.empty(l_out, B, C, dtype=torch.float32, device=x.device)
            sb, sl = C, B * C
        else:
            y = torch.empty(B, l_out, C, dtype=torch.float32, device=x.device)
            sb, sl = l_out * C, C
        need = ctx.needs_input_grad[0]
        route = torch.empty(B, l_out, C, dtype=torch.uint8, device=x.device) if need else None
        _lib.check(L.slu_pool_act_fwd(x.data_ptr(), y.data_ptr(), _ptr(route), B, Lin, C, pool, int(do_abs), float(slope),
                                      sb, sl, _stream()), "slu_pool_act_fwd")
        if need:
            ctx.save_for_backward(y, route)
            ctx.cfg = (B, Lin, C, pool, slope, sb, sl)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _lib.load()
        y, route = ctx.saved_tensors
        B, Lin, C, pool, slope, sb, sl = ctx.cfg
        dy = _f32c(dy, "dy")
        dx = torch.empty(B, Lin, C, dtype=torch.float32, device=dy.device)
        _lib.check(L.slu_pool_act_bwd(dy.data_ptr(), y.data_ptr(), route.data_ptr(), dx.data_ptr(), B, Lin, C, pool,
                                      float(slope), sb, sl, _stream()), "slu_pool_act_bwd")
        return dx, None, None, None, None


# -- split-precision (16-bit MFMA) path of the frozen stages: see csrc/slu_bf16.h ------------------------
def round_up(n, m):
    return -(-n // m) * m


def plane_dtype(nsplit):
    """Element type of the planes of split scheme `nsplit`: 2 = f16x2 (two fp16 terms), 1 / 3 = bf16 terms."""
    return torch.float16 if nsplit == 2 else torch.bfloat16


def split_bf16(x2d, nsplit):
    """fp32 (rows, K) with unit column stride -> (nsplit, rows, round_up(K, 32)) planes of 16-bit terms, zero padded
    (nsplit 1 / 3: bf16 terms, 2: the fp16 pair of the f16x2 scheme)."""
    L = _lib.load()
    rows, K = x2d.shape
    assert x2d.dtype == torch.float32 and x2d.stride(1) == 1
    planes = torch.empty(nsplit, rows, round_up(K, 32), dtype=plane_dtype(nsplit), device=x2d.device)
    _lib.check(L.slu_split_bf16(x2d.data_ptr(), x2d.stride(0), planes.data_ptr(), planes.stride(0), rows, K,
                                nsplit, _stream()), "slu_split_bf16")
    return planes


def gemm_bf16_pack(w, nsplit):
    """(N, K) fp32 weights — any strides, e.g. the .t() view of a weight — -> the scheme's 16-bit planes in MFMA
    B-fragment order (opaque byte tensor)."""
    L = _lib.load()
    N, K = w.shape
    assert w.dtype == torch.float32
    packed = torch.empty(L.slu_gemm_bf16_pack_bytes(N, K, nsplit), dtype=torch.uint8, device=w.device)
    _lib.check(L.slu_gemm_bf16_pack(w.data_ptr(), w.stride(0), w.stride(1), packed.data_ptr(), N, K, nsplit, _stream()),
               "slu_gemm_bf16_pack")
    return packed


def gemm_bf16(planes, packed, bias, N, K, out=None):
    """out (M, N) fp32 = A W^T + bias; planes (nsplit, M, ld) from split_bf16 / a split-writing stage."""
    L = _lib.load()
    nsplit, M, ld = planes.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=planes.device)
    _lib.check(L.slu_gemm_bf16(planes.data_ptr(), planes.stride(0), ld, packed.data_ptr(), _ptr(bias), out.data_ptr(),
                               out.stride(0), M, N, K, nsplit, _stream()), "slu_gemm_bf16")
    return out


def gemm_a32_ok(a, N, K):
    """Shapes slu_gemm_bf16_a32 takes: a (M, K) fp32 with unit column stride, 16-byte aligned rows; N, K % 4 == 0."""
    return (a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.stride(0) >= K
            and a.data_ptr() % 16 == 0 and N % 4 == 0 and K % 4 == 0)


def gemm_a32(a, packed, bias, N, nsplit, out=None):
    """out (M, N) fp32 = a W^T + bias with the fp32 matrix a (M, K) split on the fly inside the kernel and W packed by
    gemm_bf16_pack (W (N, K) or a transposed view of a weight): the GEMMs of trainable layers (slu_gemm_bf16_a32)."""
    L = _lib.load()
    M, K = a.shape
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _lib.check(L.slu_gemm_bf16_a32(a.data_ptr(), a.stride(0), packed.data_ptr(), _ptr(bias), out.data_ptr(), out.stride(0),
                                   M, N, K, nsplit, _stream()), "slu_gemm_bf16_a32")
    return out


def gemm_nt(a, w, bias=None, out=None, grad=False, bf16_ok=True):
    """a (M, K) @ w^T (w (N, K), any strides) + bias in the arithmetic of the trainable layers (train_nsplit; grad: a
    is a gradient): the split-precision kernel where the shape allows it, else the exact fp32 GEMM.
    bf16_ok=False: this product stays fp32 in bf16 mode (the ASR heads: BASELINE configs[4] is the SLU model)."""
    ns = train_nsplit(grad)
    if ns == 1 and not bf16_ok:
        ns = 0
    N, K = w.shape
    if ns and a.is_cuda and gemm_a32_ok(a, N, K) and (out is None or (out.stride(1) == 1 and out.stride(0) % 4 == 0)):
        return gemm_a32(a, gemm_bf16_pack(w.detach(), ns), bias, N, ns, out)
    return gemm(a, w.t(), bias, out=out)


def wconv_bf16_supported(c_in, stride, pool, k_t=None, nsplit=3):
    """Shapes slu_wconv_fwd_bf16 takes (else the exact fp32 kernel runs).  With k_t given the launcher's LDS limit
    is mirrored too: nsplit planes of (64 frames + window overhang) rows of stride * c_pad + 8 bf16 must fit 160 KiB
    (the launcher falls back from 128- to 64-frame tiles before giving up)."""
    if not (pool in (1, 2) and ((c_in == 1 and stride % 8 == 0) or (c_in > 1 and stride == 1))):
        return False
    if k_t is None:
        return True
    c_pad = 1 if c_in == 1 else round_up(c_in, 8)
    S = stride * c_pad
    kc = -(-(k_t * c_pad) // 32)
    nrows = 64 + -(-(kc * 32) // S) + 1
    return nsplit * nrows * (S + 8) * 2 <= 160 * 1024


class RowTable:
    """The input of a look-ahead super-batch WITHOUT a concatenation copy: `ptrs` = int64 device tensor of P base
    addresses (one per batch of `rows` equally long fp32 rows); stands for a (P * rows, T) float32 tensor that only
    slu_wconv_fwd_bf16 (in_table) can read.  The address table is refreshed with store_u64 before each replay."""
    requires_grad = False

    def __init__(self, ptrs, rows, T, dtype=torch.float32):
        self.ptrs, self.rows = ptrs, rows
        self.shape = (ptrs.numel() * rows, T)
        self.device = ptrs.device
        self.dtype = dtype            # float32, or int16: PCM16 samples the first block scales by PCM16_SCALE itself

    def dim(self):
        return 2

    def float(self):
        return self

    def to(self, *args, **kwargs):
        return self


PCM16_SCALE = 1.0 / 32768.0        # int16 sample -> [-1, 1): the sox / soundfile convention of the reference's loaders


def pcm16_to_f32(x):
    """int16 PCM samples -> fp32 sample / 32768 (slu_pcm16_to_f32): for a first block on the exact fp32 kernels."""
    L = _lib.load()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    _lib.check(L.slu_pcm16_to_f32(x.data_ptr(), out.data_ptr(), x.numel(), PCM16_SCALE, _stream()), "slu_pcm16_to_f32")
    return out


def store_u64(dst, values):
    """dst (int64 device tensor)[:len(values)] = values, in one launch with the values in the kernel arguments."""
    import ctypes
    L = _lib.load()
    n = len(values)
    assert 1 <= n <= 64 and dst.dtype == torch.int64 and dst.numel() >= n and dst.is_contiguous()
    arr = (ctypes.c_uint64 * n)(*[int(v) & 0xFFFFFFFFFFFFFFFF for v in values])
    _lib.check(L.slu_store_u64(dst.data_ptr(), arr, n, _stream()), "slu_store_u64")


def wconv_bf16_planes_ok(c_out, pool):
    """Can slu_wconv_fwd_bf16 write its result as split-precision planes?  (pool 1; the kernel's channel tiling
    — 1, 2, 4, 5 or 8 tiles of 16 — has to cover round_up(c_out, 32) columns)"""
    need = -(-c_out // 16)
    nt = next((n for n in (1, 2, 4, 5, 8) if n >= need), None)
    return pool == 1 and nt is not None and nt * 16 >= round_up(c_out, 32)


def wconv_fwd_bf16(x, weight, bias, B, l_in, c_in, stride, do_abs, pool, slope, time_major, nsplit, out_planes=False,
                   pack_cache=None, want_route=False, absmax=None):
    """wconv_fwd of a FROZEN block on the split-precision kernels (no route): x contiguous (B, l_in, c_in).
    out_planes: return a SplitAct (time-major rows, bf16 planes) for the next frozen GRU layer instead of fp32.
    pack_cache: a dict of the caller's (one per frozen block and weight version): the packed filters are built by
    the first call and reused by the following ones (no pack launch per super-batch).
    absmax: None, or the device address of this launch's f16x2 range word (slu_hip/guard.py)."""
    L = _lib.load()
    pcm16 = x.dtype == torch.int16          # PCM16 waveform: scaled by 1 / 32768 inside the kernel (frozen first block)
    if isinstance(x, RowTable):
        x_ptr, tab, tab_rows = None, x.ptrs.data_ptr(), x.rows
        assert c_in == 1 and x.shape == (B, l_in)
    elif pcm16:
        assert c_in == 1 and x.is_cuda and not want_route
        x = x.contiguous()
        x_ptr, tab, tab_rows = x.data_ptr(), None, 0
    else:
        x = _f32c(x, "x")
        x_ptr, tab, tab_rows = x.data_ptr(), None, 0
    pcm = (1, PCM16_SCALE) if pcm16 else (0, 1.0)
    weight = _f32c(weight, "weight")
    c_out, _, k_t = weight.shape
    l_conv = conv_out_len(l_in, k_t, stride)
    l_out = -(-l_conv // pool)
    if out_planes:
        assert time_major and wconv_bf16_planes_ok(c_out, pool)
        planes = torch.empty(nsplit, l_out * B, round_up(c_out, 32), dtype=plane_dtype(nsplit), device=x.device)
        ws, wsb, valid = _wconv_pack_ws(L, pack_cache, c_out, c_in, k_t, nsplit, x.device)
        _lib.check(L.slu_wconv_fwd_bf16(x_ptr, tab, tab_rows, weight.data_ptr(), _ptr(bias), None, None, B, l_in, c_in, c_out,
                                        k_t, stride, int(do_abs), pool, float(slope), 0, 0, planes.data_ptr(),
                                        planes.stride(0), ws.data_ptr(), wsb, valid, nsplit, absmax, *pcm, _stream()),
                   "slu_wconv_fwd_bf16")
        return SplitAct(planes, l_out, B, c_out)
    if time_major:
        out = torch.empty(l_out, B, c_out, dtype=torch.float32, device=x.device)
        sb, sl = c_out, B * c_out
    else:
        out = torch.empty(B, l_out, c_out, dtype=torch.float32, device=x.device)
        sb, sl = l_out * c_out, c_out
    ws, wsb, valid = _wconv_pack_ws(L, pack_cache, c_out, c_in, k_t, nsplit, x.device)
    route = torch.empty(B, l_out, c_out, dtype=torch.uint8, device=x.device) if want_route else None
    _lib.check(L.slu_wconv_fwd_bf16(x_ptr, tab, tab_rows, weight.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(route), B, l_in,
                                    c_in, c_out, k_t, stride, int(do_abs), pool, float(slope), sb, sl, None, 0, ws.data_ptr(), wsb,
                                    valid, nsplit, absmax, *pcm, _stream()), "slu_wconv_fwd_bf16")
    return (out, route, l_conv) if want_route else out


def _wconv_pack_ws(L, cache, c_out, c_in, k_t, nsplit, device):
    """-> (workspace, bytes, packed_valid) of slu_wconv_fwd_bf16: a fresh workspace, or the caller's cached one whose
    filter pack the first call built (valid from the second call on)."""
    wsb = L.slu_wconv_bf16_workspace_bytes(c_out, c_in, k_t, nsplit)
    if cache is None:
        return _workspace(wsb, device), wsb, 0
    key = (c_out, c_in, k_t, nsplit, str(device))
    ws = cache.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return _workspace(wsb, device), wsb, 0      # memory allocated under capture belongs to the graph: no caching
        cache[key] = ws = _workspace(wsb, device)
        return ws, wsb, 0
    return ws, wsb, 1


class SplitAct:
    """A time-major activation (T, B, C) in the split-precision format: `planes` = (nsplit, T*B, round_up(C, 32))
    bf16 (csrc/slu_bf16.h).  Travels only between FROZEN stages (no autograd)."""

    def __init__(self, planes, T, B, C):
        self.planes, self.T, self.B, self.C = planes, T, B, C


def dropout_pool_fwd_planes(x, mask, p, seed, offset, method, factor, nsplit, offset_dev=None, sub_batch=0, keep_bits=None):
    """dropout_pool_fwd whose result is written as split-precision planes (C % 32 == 0) -> SplitAct.
    keep_bits: the 1-bit mask of dropout_bits instead of in-kernel Philox draws (frozen layers)."""
    L = _lib.load()
    T, B, C = x.shape
    T_out = -(-T // factor)
    planes = torch.empty(nsplit, T_out * B, C, dtype=plane_dtype(nsplit), device=x.device)
    mp, mst, msb = _mask_args(mask, T, B, C)
    _lib.check(L.slu_dropout_pool_fwd_planes(x.data_ptr(), mp, mst, msb, _ptr(keep_bits), float(p), int(seed), int(offset),
                                             _ptr(offset_dev), int(sub_batch), 16, METHODS[method], factor,
                                             planes.data_ptr(), planes.stride(0), nsplit, T, B, C, _stream()),
               "slu_dropout_pool_fwd_planes")
    return SplitAct(planes, T_out, B, C)


def dropout_bits(T, B, C, p, seed, offset, offset_dev=None, sub_batch=0, device=None):
    """The dropout mask of a FROZEN layer as a bit stream (int32 (T, B, C // 32), bit c % 32 of word c // 32 = keep):
    slu_dropout_bits.  Consumed by gru_seq_fwd_pool_bf16 and by dropout_pool_fwd[_planes](keep_bits=...)."""
    L = _lib.load()
    bits = torch.empty(T, B, C // 32, dtype=torch.int32, device=device)
    _lib.check(L.slu_dropout_bits(bits.data_ptr(), float(p), int(seed), int(offset), _ptr(offset_dev), int(sub_batch), 16,
                                  T, B, C, _stream()), "slu_dropout_bits")
    return bits


def gru_pool_fused_ok(H, D, T, p, mask, method, factor):
    """Can the recurrence apply the layer's Dropout + Downsample in its epilogue (slu_gru_seq_fwd_pool_bf16)?  Average
    pooling over two frames with an in-kernel Philox mask (or no dropout): every layer of the reference cfgs.  Injected
    masks (parity tests) and other pooling modes take the two-launch path.  SLU_FUSE_GRU_POOL=0: off."""
    return (method == "avg" and factor == 2 and mask is None and (D * H) % 32 == 0 and T <= 65535 and 0.0 <= p < 1.0
            and os.environ.get("SLU_FUSE_GRU_POOL", "1") != "0")


# CUs a CU-masked stream may use (pipeline.cu_range_stream registers its streams here): what decides between one and two
# sequence tiles per recurrence workgroup
STREAM_CUS = {}


def gru_seq_tiles(B, H, D, fused=None, reserve=False, device=None):
    """Sequence tiles (16 sequences each) per workgroup of the split-precision recurrence (slu_gru_seq_fwd[_pool]_bf16's
    seq_tiles).  1 (default): gru_bf_fwd_kernel.  SLU_GRU_TILES=2: gru_bf2_fwd_kernel — two tiles half a step apart, the gate
    arithmetic of one interleaved with the other's MFMAs; =auto: two tiles as soon as the one-tile grid (ceil(B / 16) x D
    workgroups, one per CU) would not fit the stream's CUs in one round.  Both kernels produce the same bits.  Measured
    (profiles/r06_a_gru_two_tiles.txt): 8 % faster on bf16x3 at 2560 sequences, slower on f16x2 — gfx950 overlaps a wave's MFMAs
    with its neighbour's VALU work far less than the design assumed (csrc/slu_gru_bf16.hip) — hence opt-in."""
    if H != 128 or fused is not None or reserve:
        return 1
    env = os.environ.get("SLU_GRU_TILES", "1")
    if env in ("1", "2"):
        return int(env)
    cus = STREAM_CUS.get(torch.cuda.current_stream(device).cuda_stream)
    if cus is None:
        cus = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device).multi_processor_count
    return 2 if -(-B // 16) * D > cus else 1


def gru_seq_fwd_pool_bf16(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, keep, p, out_planes, fused=None,
                          seq_tiles=None):
    """Recurrence + Dropout(p; keep = dropout_bits or None) + avg-pool(2) in one launch -> SplitAct (out_planes) or fp32
    (ceil(T/2), B, D*H).  fused as gru_seq_fwd_bf16.  seq_tiles: None = gru_seq_tiles()."""
    L = _lib.load()
    if seq_tiles is None:
        seq_tiles = gru_seq_tiles(B, H, D, fused, False, w_hh_f.device)
    dev = w_hh_f.device
    T_out = -(-T // 2)
    if out_planes:
        planes = torch.empty(nsplit, T_out * B, D * H, dtype=plane_dtype(nsplit), device=dev)
        oa = (None, planes.data_ptr(), planes.stride(0))
    else:
        out = torch.empty(T_out, B, D * H, dtype=torch.float32, device=dev)
        oa = (out.data_ptr(), None, 0)
    if fused is not None:
        xp, K, packed, b_ih = fused
        assert gx is None and xp.shape == (nsplit, T * B, round_up(K, 32)) and xp.stride(1) == xp.shape[2]
        xa = (xp.data_ptr(), xp.stride(0), K, packed.data_ptr(), b_ih.data_ptr())
    else:
        xa = (None, 0, 0, None, None)
    _lib.check(L.slu_gru_seq_fwd_pool_bf16(_ptr(gx), w_hh_f.data_ptr(), _ptr(w_hh_r), b_hh_f.data_ptr(), _ptr(b_hh_r), *oa,
                                           _ptr(keep), float(p), *xa, T, B, H, D, nsplit, int(seq_tiles), _stream()),
               "slu_gru_seq_fwd_pool_bf16")
    return SplitAct(planes, T_out, B, D * H) if out_planes else out


def gru_layer_frozen(x, w_ih, b_ih, packed_ih, w_hh_f, b_hh_f, w_hh_r, b_hh_r, p, mask, seed, offset, method, factor,
                     nsplit, out_planes):
    """A FROZEN GRU layer + Dropout + Downsample on the split-precision kernels, outside autograd.
    x: fp32 (T, B, I) or a SplitAct; out_planes: return a SplitAct for the next frozen layer (when the pooled
    channel count allows it) instead of fp32 (T_out, B, D*H)."""
    H = w_hh_f.shape[1]
    D = 1 if w_hh_r is None else 2
    if isinstance(x, SplitAct):
        T, B, I, planes = x.T, x.B, x.C, x.planes
    else:
        x = x.contiguous()
        T, B, I = x.shape
        planes = split_bf16(x.view(T * B, I), nsplit)
    packed = packed_ih if packed_ih is not None else gemm_bf16_pack(w_ih, nsplit)
    # the mask of a frozen layer is a bit stream (dropout_bits) whichever kernel applies it; injected float masks (parity
    # tests) and channel counts without whole 32-bit words keep the in-kernel Philox / float-mask forms
    off, off_dev, sub = offset if isinstance(offset, tuple) else (offset, None, 0)
    keep = None
    if p > 0.0 and mask is None and (D * H) % 32 == 0 and T <= 65535:
        keep = dropout_bits(T, B, D * H, p, seed, off, off_dev, sub, w_hh_f.device)
    if gru_pool_fused_ok(H, D, T, p, mask, method, factor):
        # Dropout + Downsample(avg, 2) in the recurrence's epilogue: no fp32 (T, B, D*H) output, no pool launch
        to_planes = bool(out_planes) and -(-T // factor) <= 65535
        if gru_fused_input_ok(I, H, D, nsplit):
            return gru_seq_fwd_pool_bf16(None, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, keep, p, to_planes,
                                         fused=(planes, I, packed, b_ih))
        gx = gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
        return gru_seq_fwd_pool_bf16(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, keep, p, to_planes)
    if gru_fused_input_ok(I, H, D, nsplit):
        # the first GRU layer (K = 60): the recurrence computes x W_ih^T + b_ih itself — no projection launch, no gx
        raw, _ = gru_seq_fwd_bf16(None, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, False,
                                  fused=(planes, I, packed, b_ih))
    else:
        gx = gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
        raw, _ = gru_seq_fwd_bf16(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, False)
    if out_planes and (D * H) % 32 == 0 and -(-T // factor) <= 65535:
        return dropout_pool_fwd_planes(raw, mask, p, seed, off, method, factor, nsplit, off_dev, sub, keep_bits=keep)
    if p == 0.0 and factor == 1:
        return raw
    return dropout_pool_fwd(raw, mask, p, seed, off, method, factor, off_dev, sub, keep_bits=keep)


def absmax_into(t, word_ptr):
    """atomicMax of the IEEE bit pattern of max |t| into the uint32 device word at `word_ptr` (the f16x2 range guard's
    words, slu_hip/guard.py): one small launch on the current stream."""
    import ctypes
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    assert t.dtype == torch.float32 and t.is_cuda
    ptrs = (ctypes.c_void_p * 1)(t.data_ptr())
    numel = (ctypes.c_int64 * 1)(t.numel())
    _lib.check(_lib.load().slu_absmax_multi(ptrs, numel, 1, word_ptr, _stream()), "slu_absmax_multi")


def split_path_supported(H, D):
    """Shapes the split-precision kernels are instantiated for (else the exact fp32 kernels run)."""
    return H in (64, 128) and (D * 3 * H) % 64 == 0


def gru_fused_input_ok(I, H, D, nsplit):
    """Layers whose input projection slu_gru_seq_fwd_bf16 computes itself (x_planes): at most 64 input channels, H = 128,
    the f16x2 scheme — the first GRU layer of the reference architecture (K = 60).  SLU_FUSE_GRU_INPUT=0: off.
    Measured on MI355X (tools/gru_fused_probe.py, T = 300, 1024 sequences, 128 CUs): projection GEMM 275 us + recurrence
    385 us = 660 us against 511 us fused, bit-identical output, 1.9 GB less HBM traffic per super-batch (the fp32 gx of the
    largest layer is never written); the pipelined loop: 343-346 k against 325-336 k utt/s (same box).  The fused step costs
    1.70 instead of 1.27 us: its 18 extra MFMAs per wave run back to back at the end of the step (interleaving them with
    the gate math by scheduling hints made the step 2.2 us)."""
    if not (H == 128 and 1 <= I <= 64 and os.environ.get("SLU_FUSE_GRU_INPUT", "1") != "0"):
        return False
    # bf16x3 (round 5): instantiated and bit-identical to GEMM + recurrence (tests/test_hip_bf16.py), but with W_hh (144
    # registers) AND the W_ih fragments (72 for K = 60) resident the kernel spills 40-49 VGPRs to scratch at two waves per
    # SIMD, and the loop gets SLOWER: 160.3-160.7 against 190.7-196.8 k utt/s for the driver's 20-step command, 292-293
    # against 345 k steady (same box, alternating runs).  Only when SLU_FUSE_GRU_INPUT3=1 asks for it.
    return nsplit == 2 or (nsplit == 3 and os.environ.get("SLU_FUSE_GRU_INPUT3", "0") == "1")


def gru_seq_fwd_bf16(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, want_reserve=False, fused=None, seq_tiles=None):
    """Recurrence on the split-precision MFMA kernels -> (out (T, B, D*H) fp32, reserve or None).
    fused = (planes (nsplit, T*B, round_up(K, 32)), K, packed W_ih, b_ih) with gx None: the kernel computes the input
    projection itself (gru_fused_input_ok shapes; bit-identical to gemm_bf16 + this call)."""
    L = _lib.load()
    dev = w_hh_f.device
    if seq_tiles is None:
        seq_tiles = gru_seq_tiles(B, H, D, fused, want_reserve, dev)
    out = torch.empty(T, B, D * H, dtype=torch.float32, device=dev)
    reserve = None
    if want_reserve:
        reserve = torch.empty(L.slu_gru_reserve_bytes(T, B, H, D) // 4, dtype=torch.float32, device=dev)
    if fused is not None:
        planes, K, packed, b_ih = fused
        assert gx is None and planes.shape == (nsplit, T * B, round_up(K, 32)) and planes.stride(1) == planes.shape[2]
        xa = (planes.data_ptr(), planes.stride(0), K, packed.data_ptr(), b_ih.data_ptr())
    else:
        xa = (None, 0, 0, None, None)
    _lib.check(L.slu_gru_seq_fwd_bf16(_ptr(gx), w_hh_f.data_ptr(), _ptr(w_hh_r), b_hh_f.data_ptr(), _ptr(b_hh_r),
                                      out.data_ptr(), _ptr(reserve), *xa, T, B, H, D, nsplit, int(seq_tiles), _stream()),
               "slu_gru_seq_fwd_bf16")
    return out, reserve


TN_SMALL_ROWS = 4096      # weight gradients of a GRU layer with T*B up to this many rows go through ONE batched launch


def gemm_tn_batched(problems, rowsum=None):
    """problems: [(A (K, M), B (K, N), C (M, N))] 2-D fp32 views with unit column stride (row strides free);
    C_q = A_q^T B_q for up to four problems in one launch (slu_gemm_tn_batched).
    rowsum: optional (src (R, ...) contiguous fp32, dst (numel of a row)) — dst = src.sum(0) in the same launch."""
    import ctypes
    L = _lib.load()
    n = len(problems)
    for A, B, C in problems:
        assert A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1 and A.shape[0] == B.shape[0]
        assert C.shape == (A.shape[1], B.shape[1])
    if rowsum:
        src, dst = rowsum
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32
        assert src.numel() == src.shape[0] * dst.numel()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    arr = lambda ty, vals: (ty * n)(*vals)
    _lib.check(L.slu_gemm_tn_batched(arr(vp, [p[0].data_ptr() for p in problems]), arr(i64, [p[0].stride(0) for p in problems]),
                                     arr(vp, [p[1].data_ptr() for p in problems]), arr(i64, [p[1].stride(0) for p in problems]),
                                     arr(vp, [p[2].data_ptr() for p in problems]), arr(i64, [p[2].stride(0) for p in problems]),
                                     arr(i64, [p[2].shape[0] for p in problems]), arr(i64, [p[2].shape[1] for p in problems]),
                                     arr(i64, [p[0].shape[0] for p in problems]), n,
                                     rowsum[0].data_ptr() if rowsum else None, rowsum[0].shape[0] if rowsum else 0,
                                     rowsum[1].numel() if rowsum else 0, rowsum[1].data_ptr() if rowsum else None,
                                     _stream()), "slu_gemm_tn_batched")


_TN_TICKETS = {}          # (device, stream) -> zeroed ticket words of slu_gemm_tn_batched_splitk (the kernel leaves them zero)
_TN_RETIRED = []          # outgrown ticket buffers, kept alive for the graphs that captured their addresses
TN_SPLITK_MIN_ROWS = 2048


def tn_tickets(dev, tiles=0):
    """The zeroed ticket words of the CURRENT stream for slu_gemm_tn_batched_splitk (the kernel leaves them zero, so one
    buffer serves every launch of a stream).  Created outside hipGraph capture only: a buffer allocated while capturing
    would belong to that graph's private memory pool and die with it while this cache still hands it out —
    pipeline.StepGraph calls this before it starts capturing."""
    key = (dev.index, _stream())
    tk = _TN_TICKETS.get(key)
    if tk is None or tk.numel() < tiles:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("slu_gemm_tn_batched_splitk: no ticket buffer for this stream yet and the stream is capturing; "
                               "call slu_hip.ops.tn_tickets(device) on this stream before the capture starts")
        if tk is not None:
            _TN_RETIRED.append(tk)      # a hipGraph captured earlier may still address the smaller buffer: never freed
        tk = _TN_TICKETS[key] = torch.zeros(max(1024, tiles), dtype=torch.int32, device=dev)
    return tk


def wgrad_branch():
    """-> (mode, workgroup budget) of the batched weight-gradient launch of a long GRU layer (>= 2048 rows).  The BUDGET
    applies to every such launch (the split count decides the summation order: it must not depend on the loop kind); the
    BRANCH only inside a fully trainable loop (GRULayerFn.backward under _Fork.defer), elsewhere the launch stays in line.
    SLU_WGRAD_BRANCH:
      layer (default)  on an auxiliary stream / graph branch beside the layer's data-gradient GEMM, joined at the end of the
                       layer's backward, with a budget of SLU_WGRAD_WGS = 216 workgroups (of the 512 a full round has);
      pass             the same launch left open until the whole backward pass has ended (beside the BPTT of the layer
                       below; the trainer joins once) — bit-identical, measured slower;
      0                in line on the main stream, full round (rounds 3-4).
    Measured on MI355X (profiles/r05_hi_wgrad_branch.txt; B = 64 x 3 s, every layer trainable, ms per step, same box): in line
    2.780 | layer 2.620 | pass 2.697 (budget 144: 2.707, 288: 2.687, 512: 2.838); second box: in line 2.945 | layer with
    budget 72 / 144 / 216 / 288 / 360 / 512: 2.979 / 2.855 / 2.804 / 2.939 / 2.968 / 2.969.  Beside the data-gradient GEMM the
    weight gradients are free as long as they do not take every CU; beside the BPTT they still lengthen its dependent steps
    by more than they save, budget or not — and not by sharing its CUs: with BPTT workgroups that own their CU (a probe
    build claiming all 512 registers per lane) the open form measured 2.686 against 2.683 ms."""
    if _WGRAD[0] is None:
        resolve_wgrad()
    return _WGRAD[0]


_WGRAD = [None]


def resolve_wgrad():
    """Read SLU_WGRAD_BRANCH / SLU_WGRAD_WGS ONCE — training.Trainer calls this when it is built — instead of on every
    backward call: the budget sets the split-K factor, i.e. the summation order; an environment change in mid-run would
    silently lose bit-identity between steps (and between a run and its resumption).  dp.make_comm compares
    wgrad_signature() across the ranks."""
    mode = os.environ.get("SLU_WGRAD_BRANCH", "layer")
    if mode not in ("layer", "pass", "0"):
        raise ValueError("SLU_WGRAD_BRANCH=%r: expected layer, pass or 0" % mode)
    _WGRAD[0] = (mode, 0 if mode == "0" else int(os.environ.get("SLU_WGRAD_WGS", "216")))
    return _WGRAD[0]


def wgrad_signature():
    """What must be EQUAL on every data-parallel rank for the replicas to stay bit-identical: the weight-gradient launch's
    (mode, workgroup budget) and the arithmetic modes."""
    mode, budget = wgrad_branch()
    return "%s/%d/%s/%s/%s" % (mode, budget, os.environ.get("SLU_TRAIN_MATH", "fp32"), os.environ.get("SLU_FROZEN_MATH", "bf16x3"),
                               os.environ.get("SLU_DTYPE", "f32"))


def gemm_tn_splitk_ok(operands):
    """operands: [(A (K, M), B (K, N)), ...] — shapes slu_gemm_tn_batched_splitk takes: every M, N and row stride a
    multiple of 4, unit column strides, 16-byte aligned operands."""
    return all(A.shape[1] % 4 == 0 and B.shape[1] % 4 == 0 and A.stride(0) % 4 == 0 and B.stride(0) % 4 == 0
               and A.stride(1) == 1 and B.stride(1) == 1 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0
               for A, B in operands)


def gemm_tn_batched_splitk(problems, rowsum=None, max_wg=0):
    """gemm_tn_batched for long k ranges (the weight gradients of a GRU layer with thousands of rows): one launch, the k
    range split over workgroups, partial tiles folded in a fixed order by each tile's last workgroup."""
    import ctypes
    L = _lib.load()
    n = len(problems)
    for A, B, C in problems:
        assert A.shape[0] == B.shape[0] and C.shape == (A.shape[1], B.shape[1])
    assert gemm_tn_splitk_ok([(A, B) for A, B, _ in problems]) and all(C.stride(1) == 1 for _, _, C in problems)
    if rowsum:
        src, dst = rowsum
        assert src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype == torch.float32
        assert src.numel() == src.shape[0] * dst.numel()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    arr = lambda ty, vals: (ty * n)(*vals)
    Ms, Ns, Ks = (arr(i64, [p[2].shape[0] for p in problems]), arr(i64, [p[2].shape[1] for p in problems]),
                  arr(i64, [p[0].shape[0] for p in problems]))
    dev = problems[0][0].device
    wsb = L.slu_gemm_tn_splitk_workspace_bytes_wg(Ms, Ns, Ks, n, int(max_wg))    # max_wg: workgroup budget (0 = a full round)
    ws = _workspace(wsb, dev)
    tiles = sum(-(-p[2].shape[0] // 64) * -(-p[2].shape[1] // 64) for p in problems)
    tk = tn_tickets(dev, tiles)
    _lib.check(L.slu_gemm_tn_batched_splitk_wg(arr(vp, [p[0].data_ptr() for p in problems]), arr(i64, [p[0].stride(0) for p in problems]),
                                            arr(vp, [p[1].data_ptr() for p in problems]), arr(i64, [p[1].stride(0) for p in problems]),
                                            arr(vp, [p[2].data_ptr() for p in problems]), arr(i64, [p[2].stride(0) for p in problems]),
                                            Ms, Ns, Ks, n,
                                            rowsum[0].data_ptr() if rowsum else None, rowsum[0].shape[0] if rowsum else 0,
                                            rowsum[1].numel() if rowsum else 0, rowsum[1].data_ptr() if rowsum else None,
                                            ws.data_ptr(), wsb, tk.data_ptr(), tk.numel(), int(max_wg), _stream()),
               "slu_gemm_tn_batched_splitk")


def colsum(x2d, out=None, accumulate=False):
    L = _lib.load()
    M, N = x2d.shape
    assert x2d.stride(1) == 1
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    _lib.check(L.slu_colsum_f32(x2d.data_ptr(), x2d.stride(0), out.data_ptr(), M, N, int(accumulate),
                                _stream()), "slu_colsum_f32")
    return out


def gru_seq_fwd(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, want_reserve):
    L = _lib.load()
    out = torch.empty(T, B, D * H, dtype=torch.float32, device=gx.device)
    reserve = None
    if want_reserve:
        reserve = torch.empty(L.slu_gru_reserve_bytes(T, B, H, D) // 4, dtype=torch.float32, device=gx.device)
    _lib.check(L.slu_gru_seq_fwd(gx.data_ptr(), w_hh_f.data_ptr(), _ptr(w_hh_r), b_hh_f.data_ptr(),
                                 _ptr(b_hh_r), out.data_ptr(), _ptr(reserve), T, B, H, D, _stream()),
               "slu_gru_seq_fwd")
    return out, reserve


_PROJ_STATE = {}          # (device, stream, T, B) -> hand-off counters of slu_gru_proj_seq_fwd (zeroed once, then launch-owned)


def gru_proj_fused_ok(x2, w_ih, T, B, I, H, D):
    """Can the layer's input projection and recurrence share one launch (slu_gru_proj_seq_fwd)?  Exact-fp32 training
    arithmetic, a shape the kernel takes, plain row-major operands — and, while a hipGraph is being captured, hand-off
    counters that already exist for this (shape, stream) (they are created by the eager steps that precede every capture;
    a buffer allocated during the capture would belong to the graph's private pool).
    OPT-IN (SLU_FUSE_PROJ_GRU=1): bit-identical to the two launches, but measured SLOWER on MI355X — intent layer (T = 19,
    B = 64) 57.7 us against 51.0 for projection + recurrence on the 96-CU training partition, T = 300: 523 against 427
    (profiles/r04_ap_proj_gru_fused.txt): the consumer's write-through loads and counter reads cost more per step than the
    overlap with the projection returns.  Kept as the tested starting point of that fusion."""
    if os.environ.get("SLU_FUSE_PROJ_GRU", "0") != "1" or not x2.is_cuda or train_nsplit(False) != 0:
        return False
    if not (x2.stride(1) == 1 and w_ih.stride(1) == 1 and x2.dtype == w_ih.dtype == torch.float32):
        return False
    if not _lib.load().slu_gru_proj_supported(T, B, I, H, D):
        return False
    from . import pipeline as _pl
    if 0 < _pl.cu_split() < 64:         # consumers (<= 32 workgroups) + producers must be co-resident on the training partition
        return False
    key = (x2.device.index, _stream(), T, B)
    return key in _PROJ_STATE or not torch.cuda.is_current_stream_capturing()


def gru_proj_seq_fwd(x2, w_ih, b_ih, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, I, H, D, want_reserve):
    """gx = x2 w_ih^T + b_ih and the recurrence over it in ONE launch; -> (out (T, B, D*H), reserve or None).  Bit-identical
    to gemm + gru_seq_fwd."""
    L = _lib.load()
    dev = x2.device
    key = (dev.index, _stream(), T, B)
    st = _PROJ_STATE.get(key)
    if st is None:
        st = _PROJ_STATE[key] = torch.zeros(L.slu_gru_proj_state_words(T, B), dtype=torch.int32, device=dev)
    gx = torch.empty(T * B, D * 3 * H, dtype=torch.float32, device=dev)
    out = torch.empty(T, B, D * H, dtype=torch.float32, device=dev)
    reserve = None
    if want_reserve:
        reserve = torch.empty(L.slu_gru_reserve_bytes(T, B, H, D) // 4, dtype=torch.float32, device=dev)
    _lib.check(L.slu_gru_proj_seq_fwd(x2.data_ptr(), x2.stride(0), w_ih.data_ptr(), w_ih.stride(0), _ptr(b_ih), gx.data_ptr(),
                                      w_hh_f.data_ptr(), _ptr(w_hh_r), b_hh_f.data_ptr(), _ptr(b_hh_r), out.data_ptr(),
                                      _ptr(reserve), T, B, I, H, D, st.data_ptr(), st.numel(), _stream()), "slu_gru_proj_seq_fwd")
    return out, reserve


def gru_seq_bwd(d_out, reserve, w_hh_f, w_hh_r, T, B, H, D):
    L = _lib.load()
    d_out = _f32c(d_out, "d_out")
    dev = d_out.device
    d_gx = torch.empty(T, B, D * 3 * H, dtype=torch.float32, device=dev)
    d_gh = torch.empty(T, B, D * 3 * H, dtype=torch.float32, device=dev)
    nbt = int(L.slu_gru_bias_tiles(T, B, H, D))
    d_bias_part = torch.empty(nbt, D, 6 * H, dtype=torch.float32, device=dev)
    _lib.check(L.slu_gru_seq_bwd(d_out.data_ptr(), reserve.data_ptr(), w_hh_f.data_ptr(), _ptr(w_hh_r),
                                 d_gx.data_ptr(), d_gh.data_ptr(), d_bias_part.data_ptr(), T, B, H, D,
                                 _stream()), "slu_gru_seq_bwd")
    return d_gx, d_gh, d_bias_part


def _mask_args(mask, T, B, C):
    """mask: None or a float32 {0,1} tensor of logical shape (T,B,C) with unit channel stride."""
    if mask is None:
        return 0, 0, 0
    assert mask.dtype == torch.float32 and tuple(mask.shape) == (T, B, C) and mask.stride(2) == 1
    return mask.data_ptr(), mask.stride(0), mask.stride(1)


def dropout_pool_fwd(x, mask, p, seed, offset, method, factor, offset_dev=None, sub_batch=0, keep_bits=None):
    """offset_dev: optional 1-element int64 CUDA tensor added to `offset` on the device;
    sub_batch > 0: B is sub-batches of that size, sub-batch k uses offset + 16 k (consecutive steps);
    keep_bits: the 1-bit mask of dropout_bits instead of in-kernel Philox draws (frozen layers)."""
    L = _lib.load()
    T, B, C = x.shape
    T_out = -(-T // factor)
    y = torch.empty(T_out, B, C, dtype=torch.float32, device=x.device)
    mp, mst, msb = _mask_args(mask, T, B, C)
    _lib.check(L.slu_dropout_pool_fwd(x.data_ptr(), mp, mst, msb, _ptr(keep_bits), float(p), int(seed), int(offset),
                                      _ptr(offset_dev), int(sub_batch), 16, METHODS[method], factor, y.data_ptr(),
                                      T, B, C, _stream()),
               "slu_dropout_pool_fwd")
    return y


def dropout_pool_bwd(dy, x, mask, p, seed, offset, method, factor, offset_dev=None):
    L = _lib.load()
    T, B, C = x.shape
    dy = _f32c(dy, "dy")
    dx = torch.empty(T, B, C, dtype=torch.float32, device=x.device)
    mp, mst, msb = _mask_args(mask, T, B, C)
    _lib.check(L.slu_dropout_pool_bwd(dy.data_ptr(), x.data_ptr(), 0, mp, mst, msb, float(p), int(seed),
                                      int(offset), _ptr(offset_dev), 0, 16, METHODS[method], factor, dx.data_ptr(), T, B, C,
                                      _stream()), "slu_dropout_pool_bwd")
    return dx


_TICKETS = {}


def _ticket(dev):
    """A persistent zeroed device word per (device, stream) for kernels whose last workgroup finishes a reduction
    (it leaves the word at zero).  None while a hipGraph capture is running and the word does not exist yet (memory
    allocated during a capture belongs to the graph): the caller then takes the separate-launch path."""
    if os.environ.get("SLU_HEAD_TICKET", "1") == "0":
        return None
    key = (dev.index, _stream())
    t = _TICKETS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        t = _TICKETS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def cls_maxpool_ce_fwd(h, weight, bias, y, values_per_slot, want_grad, epoch_sums=None, drop=None):
    """-> (loss_acc (2) or None, logits (B,V), pred (B,S), argmax_t (B,V) int32, d_logits or None[, h_drop])
    epoch_sums: optional float64 (2) device tensor; B * (loss, acc) is added to it by the same launch.
    drop: None, or (p, seed, offset, offset_dev) — the Dropout in front of the classifier fused into this launch (h is
    then the raw GRU output; the dropped activations are returned as a sixth value)."""
    import ctypes
    L = _lib.load()
    T, B, C = h.shape
    S = len(values_per_slot)
    V = int(sum(values_per_slot))
    dev = h.device
    logits = torch.empty(B, V, dtype=torch.float32, device=dev)
    argmax_t = torch.empty(B, V, dtype=torch.int32, device=dev)
    pred = torch.empty(B, S, dtype=torch.int64, device=dev)
    d_logits = torch.empty(B, V, dtype=torch.float32, device=dev) if (want_grad and y is not None) else None
    row_stats = torch.empty(B, 2, dtype=torch.float32, device=dev) if y is not None else None
    loss_acc = torch.empty(2, dtype=torch.float32, device=dev) if y is not None else None
    vps = (ctypes.c_int64 * S)(*[int(v) for v in values_per_slot])
    p, seed, offset, offset_dev = drop if drop else (0.0, 0, 0, None)
    h_drop = torch.empty_like(h) if drop else None
    _lib.check(L.slu_cls_maxpool_ce_fwd(h.data_ptr(), weight.data_ptr(), bias.data_ptr(), _ptr(y), vps, S,
                                        logits.data_ptr(), argmax_t.data_ptr(), pred.data_ptr(), _ptr(d_logits),
                                        _ptr(row_stats), _ptr(loss_acc), _ptr(epoch_sums),
                                        _ptr(_ticket(dev)) if y is not None else 0,
                                        float(p), int(seed), int(offset), _ptr(offset_dev), _ptr(h_drop),
                                        T, B, C, _stream()),
               "slu_cls_maxpool_ce_fwd")
    if drop:
        return loss_acc, logits, pred, argmax_t, d_logits, h_drop
    return loss_acc, logits, pred, argmax_t, d_logits


def head_dropout_fusable(weight, p, mask, method, factor, h=None):
    """The Dropout between the last intent GRU layer and the classifier can be drawn inside the head kernels: Philox
    masks (no injected mask tensor), no Downsample, four-channel alignment OF WHAT THE HEAD READS — the classifier's
    in_features (= the last GRU layer's D * H; the layer's output is a fresh contiguous tensor).  The decision is taken
    before that layer runs, so it must not look at the layer's INPUT (round 3 did: an encoder width of 256 in front of a
    unidirectional 50-unit intent layer passed the gate and the head kernel then refused the 50-channel rows).
    h: optionally the head's actual input, checked as well."""
    ok = (p > 0.0 and mask is None and factor == 1 and weight.dim() == 2 and weight.shape[1] % 4 == 0
          and weight.data_ptr() % 16 == 0 and os.environ.get("SLU_FUSE_HEAD_DROPOUT", "1") != "0")
    if ok and h is not None:
        ok = h.shape[-1] == weight.shape[1] and h.is_contiguous() and h.data_ptr() % 16 == 0
    return ok


# ------------------------------------------------------------------------------------------------
# autograd Functions (one per fused stage of the encoder)
# ------------------------------------------------------------------------------------------------
class IntentHeadFn(torch.autograd.Function):
    """final_classifier Linear -> FinalPool (max over time) -> per-slot cross-entropy / accuracy
    (reference models.py:709, :112-123, :811-821).  h time-major (T,B,C), y (B,S) int64.
    Returns (loss, acc, logits (B,V), pred (B,S)); only `loss` carries a gradient."""
    last_loss_acc = None
    epoch_sums = None       # float64 (2) device tensor while a training loop wants B * (loss, acc) accumulated in-kernel

    @staticmethod
    def forward(ctx, h, weight, bias, y, values_per_slot, drop=None):
        """drop: None, or (p, seed, offset, offset_dev): h is the RAW output of the last intent GRU layer and the
        Dropout in front of the classifier (models.py:700) is drawn inside the head kernels (forward and backward)."""
        h = h.contiguous()
        y = y.contiguous()
        need = any(ctx.needs_input_grad[:3])
        sums = IntentHeadFn.epoch_sums
        if sums is not None:
            assert sums.dtype == torch.float64 and sums.numel() >= 2 and sums.is_contiguous() and sums.device == h.device
        res = cls_maxpool_ce_fwd(h, weight, bias, y, values_per_slot, need, sums, drop)
        loss_acc, logits, pred, argmax_t, d_logits = res[:5]
        ctx.drop = drop
        if need:
            ctx.save_for_backward(res[5] if drop else h, weight, argmax_t, d_logits)
        ctx.set_materialize_grads(False)       # no zero-filled gradients for acc / logits / pred
        ctx.mark_non_differentiable(logits, pred)
        acc = loss_acc[1]
        ctx.mark_non_differentiable(acc)
        IntentHeadFn.last_loss_acc = loss_acc  # (2,) float32 [loss, acc]: one-kernel metric accumulation
        return loss_acc[0], acc, logits, pred

    @staticmethod
    def backward(ctx, d_loss, _d_acc, _d_logits, _d_pred):
        L = _lib.load()
        if d_loss is None:
            return None, None, None, None, None, None
        h, weight, argmax_t, d_logits = ctx.saved_tensors
        p, seed, offset, offset_dev = ctx.drop if ctx.drop else (0.0, 0, 0, None)
        T, B, C = h.shape
        V = weight.shape[0]
        g = d_loss.contiguous().float()
        dh = torch.empty_like(h) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dW = torch.empty_like(weight) if need_w else None
        db = torch.empty(V, dtype=torch.float32, device=h.device) if need_w else None
        _lib.check(L.slu_cls_maxpool_ce_bwd(d_logits.data_ptr(), argmax_t.data_ptr(), h.data_ptr(),
                                            weight.data_ptr(), g.data_ptr(), _ptr(dh), _ptr(dW), _ptr(db),
                                            float(p), int(seed), int(offset), _ptr(offset_dev),
                                            T, B, C, V, _stream()), "slu_cls_maxpool_ce_bwd")
        return dh, dW, db, None, None, None



class FrameHeadFn(torch.autograd.Function):
    """Linear -> F.cross_entropy(ignore_index=-1) + frame accuracy of the ASR pre-training heads
    (reference models.py:291-331: phoneme_linear / word_linear on every frame).  h time-major (T,B,C),
    y (B,T) int64 with -1 = unlabelled frame.  Returns (loss, acc); only `loss` carries a gradient.
    The (N, V) logits buffer (N = T*B, V up to 10 000) is overwritten in place with d loss / d logits."""

    @staticmethod
    def forward(ctx, h, weight, bias, y):
        L = _lib.load()
        T, B, C = h.shape
        V = weight.shape[0]
        hn = _f32c(h, "h").reshape(T * B, C)
        y_tm = y.t().contiguous().reshape(-1)                       # row t*B + b, like hn
        if y_tm.dtype != torch.int64 or y_tm.numel() != T * B:
            raise TypeError("FrameHeadFn: y must be int64 of shape (B, T)")
        need = any(ctx.needs_input_grad[:3])
        logits = gemm_nt(hn, weight, bias, bf16_ok=False)          # train_nsplit arithmetic (fp32 in bf16 mode)
        row_stats = torch.empty(2 * T * B, dtype=torch.float32, device=h.device)
        out3 = torch.empty(3, dtype=torch.float32, device=h.device)
        _lib.check(L.slu_frame_ce_fwd(logits.data_ptr(), y_tm.data_ptr(), T * B, V, -1, int(need),
                                      row_stats.data_ptr(), out3.data_ptr(), _stream()), "slu_frame_ce_fwd")
        if need:
            ctx.save_for_backward(hn, weight, logits)
        ctx.set_materialize_grads(False)
        ctx.shape = (T, B, C)
        acc = out3[1]
        ctx.mark_non_differentiable(acc)
        return out3[0], acc

    @staticmethod
    def backward(ctx, d_loss, _d_acc):
        if d_loss is None:
            return None, None, None, None
        hn, weight, d_logits = ctx.saved_tensors
        T, B, C = ctx.shape
        g = d_loss.float()
        dh = dW = db = None
        if ctx.needs_input_grad[0]:
            dh = gemm_nt(d_logits, weight.t(), grad=True, bf16_ok=False).view(T, B, C)
        if ctx.needs_input_grad[1]:
            dW = _wgrad(d_logits, hn, None, bf16_ok=False)
        if ctx.needs_input_grad[2]:
            db = colsum(d_logits)
        scale_multi([t for t in (dh, dW, db) if t is not None], g)      # d loss upstream (a device scalar), one launch
        return dh, dW, db, None


class SincBlockFn(torch.autograd.Function):
    """SincLayer -> Abs -> MaxPool1d(ceil) -> LeakyReLU  (models.py:77-110, :163-168, :205, :211).
    x (B,T) -> (B, L_out, N_filt) channels-last, or (L_out, B, N_filt) when time_major."""

    @staticmethod
    def forward(ctx, x, b1, band, filt_dim, fs, stride, pool, slope, time_major, do_abs=True):
        B, T = x.shape
        filters = sinc_filters(b1, band, filt_dim, fs)
        need = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        ns = 1 if bf16_mode() else 0           # the forward pass of a TRAINABLE block is exact fp32 (train_nsplit: kinks)
        if ns and pool in (1, 2) and wconv_bf16_supported(1, stride, pool, filt_dim, ns):
            # bf16 operands on the MFMA (waveform and filterbank rounded to bf16), the epilogue and the route bits as
            # the fp32 kernel's; the backward below is the exact fp32 one on the saved fp32 input
            res = wconv_fwd_bf16(x, filters.view(-1, 1, filt_dim), None, B, T, 1, stride, do_abs, pool, slope, time_major, ns,
                                 want_route=need and (do_abs or pool != 1))
            out, route, l_conv = res if isinstance(res, tuple) else (res, None, conv_out_len(T, filt_dim, stride))
        else:
            out, route, l_conv = wconv_fwd(x, filters.view(-1, 1, filt_dim), None, B, T, 1, stride, do_abs,
                                           pool, slope, time_major, need and (do_abs or pool != 1))
        ctx.cfg = (B, T, filt_dim, fs, stride, pool, slope, time_major, l_conv, do_abs)
        if need:
            ctx.save_for_backward(x, b1, band, out, route)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, T, filt_dim, fs, stride, pool, slope, time_major, l_conv, do_abs = ctx.cfg
        x, b1, band, out, route = ctx.saved_tensors
        n = b1.numel()
        d_conv = wconv_bwd_act(dy, out, route, B, l_conv, n, do_abs, pool, slope, time_major)
        dW, _ = wconv_bwd_weight(d_conv, x, B, T, 1, n, filt_dim, stride, False)
        db1, dband = sinc_filters_bwd(b1, band, dW.view(n, filt_dim), filt_dim, fs)
        return None, db1, dband, None, None, None, None, None, None, None


class ConvBlockFn(torch.autograd.Function):
    """Conv1d -> [Abs] -> MaxPool1d(ceil) -> LeakyReLU/ReLU  (models.py:190/200, :205, :211-213).
    x channels-last (B, L, Cin) -> (B, L_out, Cout) or time-major (L_out, B, Cout)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, do_abs, pool, slope, time_major):
        B, l_in, c_in = x.shape
        x = x.contiguous()
        need_route = (pool != 1 or do_abs) and any(ctx.needs_input_grad[:3])
        k_t = weight.shape[2]
        ns = 1 if bf16_mode() else 0           # the forward pass of a TRAINABLE block is exact fp32 (train_nsplit: kinks)
        if ns and pool in (1, 2) and wconv_bf16_supported(c_in, stride, pool, k_t, ns) and weight.shape[0] <= 128:
            res = wconv_fwd_bf16(x, weight.detach(), None if bias is None else bias.detach(), B, l_in, c_in, stride, do_abs,
                                 pool, slope, time_major, ns, want_route=need_route)
            out, route, l_conv = res if isinstance(res, tuple) else (res, None, conv_out_len(l_in, k_t, stride))
        else:
            out, route, l_conv = wconv_fwd(x, weight, bias, B, l_in, c_in, stride, do_abs, pool, slope,
                                           time_major, need_route)
        ctx.cfg = (B, l_in, c_in, stride, do_abs, pool, slope, time_major, l_conv)
        ctx.save_for_backward(x, weight, out, route)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        B, l_in, c_in, stride, do_abs, pool, slope, time_major, l_conv = ctx.cfg
        x, weight, out, route = ctx.saved_tensors
        c_out, _, k_t = weight.shape
        d_conv = wconv_bwd_act(dy, out, route, B, l_conv, c_out, do_abs, pool, slope, time_major)
        dx = dW = db = None
        dev = x.device
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW = torch.empty(c_out, c_in, k_t, dtype=torch.float32, device=dev)
            db = torch.empty(c_out, dtype=torch.float32, device=dev) if ctx.has_bias else None
            with _Fork(dev, 0):                        # independent of the data gradient below
                wconv_bwd_weight(d_conv, x, B, l_in, c_in, c_out, k_t, stride, ctx.has_bias, out=(dW, db))
        if ctx.needs_input_grad[0]:
            if stride != 1:
                # a strided layer that is not the first one (the reference accepts any cnn_stride, models.py:200): its data
                # gradient is the stride-1 data gradient of d_conv with stride - 1 zeros inserted between frames — the zeros
                # add nothing, so the result is exact; the kernel does `stride` times the necessary work, which is fine for
                # a geometry no shipped cfg uses.  (zeros + strided copy: data movement only)
                l_conv1 = l_in + 2 * (k_t // 2) - k_t + 1
                d_up = torch.zeros(B, l_conv1, c_out, dtype=torch.float32, device=dev)
                d_up[:, :(l_conv - 1) * stride + 1:stride] = d_conv
                d_conv_dx, l_dx = d_up, l_conv1
            else:
                d_conv_dx, l_dx = d_conv, l_conv
            ns = train_nsplit(True)
            if ns and k_t % 2 == 1 and wconv_bf16_supported(c_out, 1, 1, k_t, ns) and c_in <= 128:
                # data gradient = the same windowed contraction with the filters transposed and reversed in time, on
                # split-precision operands (the flip / transpose is a 72 KB copy).  Odd kernel sizes only: for an even k_t
                # the "same"-padded forward convolution is one frame longer than its input and is not this transpose
                w_t = weight.detach().transpose(0, 1).flip(2).contiguous()
                dx = wconv_fwd_bf16(d_conv_dx, w_t, None, B, l_dx, c_out, 1, False, 1, 1.0, False, ns)
            else:
                dx = wconv_bwd_data(d_conv_dx, weight, B, l_in)
        _Fork.join(dev)
        return dx, dW, db, None, None, None, None, None


def gemm_tn_bf16_ok(a, b):
    """Shapes slu_gemm_tn_bf16 takes: A (K, M), B (K, N) fp32 views with unit column stride."""
    return (a.dtype == b.dtype == torch.float32 and a.shape[0] == b.shape[0] and a.stride(1) == 1 and b.stride(1) == 1
            and a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0 and a.stride(0) % 4 == 0 and b.stride(0) % 4 == 0
            and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0)


def gemm_tn_bf16(a, b, out=None, nsplit=1):
    """out (M, N) = a^T b with a (K, M), b (K, N) split (nsplit 2 / 3) or rounded to bf16 (1) on the way into the MFMA,
    fp32 accumulation: the weight gradients of the trainable layers (slu_gemm_tn_bf16)."""
    L = _lib.load()
    K, M = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1
    wsb = L.slu_gemm_tn_bf16_workspace_bytes(M, N, K)
    ws = _workspace(wsb, a.device) if wsb else None
    _lib.check(L.slu_gemm_tn_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                  M, N, K, nsplit, _ptr(ws), wsb, _stream()), "slu_gemm_tn_bf16")
    return out


def _wgrad(a, b, out, bf16_ok=True):
    """out = a^T b (a (K, M) gradients, b (K, N) activations) in the arithmetic of the trainable layers (train_nsplit):
    the split-precision TN kernel where it takes the shape, else the exact fp32 GEMM on the transposed view."""
    ns = train_nsplit(True)
    if ns == 1 and not bf16_ok:
        ns = 0
    if ns and gemm_tn_bf16_ok(a, b) and (out is None or out.stride(1) == 1):
        return gemm_tn_bf16(a, b, out, ns)
    return gemm(a.t(), b, out=out)


def _gru_dx(g2, w_ih, T, B, I, H, D):
    """dx = d_gx W_ih (K = D * 3H) in the arithmetic of the trainable layers (train_nsplit): d_gx split on the fly, W_ih^T
    packed in fragment order in place — or the exact fp32 MFMA GEMM."""
    return gemm_nt(g2, w_ih.t(), grad=True).view(T, B, I)     # (N = I, K = D*3H) view of the stacked weight


class GRULayerFn(torch.autograd.Function):
    """nn.GRU (1 layer, h0 = 0) -> RNNSelect -> Dropout -> Downsample  (models.py:232-253).
    x time-major (T, B, I) -> (T_out, B, D*H).
    w_ih (D*3H, I) / b_ih (D*3H) are the direction-stacked STORAGE the parameters weight_ih_l0[_reverse] /
    bias_ih_l0[_reverse] are views of (models.GRU links them), so that the input projection of both
    directions is ONE GEMM launch (N = 768) and so are its two gradients, without any concatenation;
    the four parameters are passed as well, only to receive their slices of those gradients."""

    @staticmethod
    def forward(ctx, x, w_ih, b_ih, w_ih_f, w_ih_r, b_ih_f, b_ih_r, w_hh_f, b_hh_f, w_hh_r, b_hh_r,
                p, mask, seed, offset, method, factor, nsplit=0, packed_ih=None):
        """nsplit: 0 = exact fp32 MFMA; 3 / 1 = the two contractions of the FORWARD pass (x W_ih^T and the
        recurrence's h W_hh^T) on the split-precision bf16 MFMA kernels (csrc/slu_bf16.h) — 3: fp32-class (used
        for frozen layers), 1: plain bf16 (BASELINE configs[4]); the backward pass is exact fp32 either way."""
        x = x.contiguous()
        T, B, I = x.shape
        H = w_hh_f.shape[1]
        D = 1 if w_hh_r is None else 2
        need = any(ctx.needs_input_grad[:11])
        if nsplit and split_path_supported(H, D):
            planes = split_bf16(x.view(T * B, I), nsplit)
            # packed_ih: the owner's cached bf16 planes of a FROZEN W_ih (models.GRU keeps them per weight version)
            packed = packed_ih if packed_ih is not None else gemm_bf16_pack(w_ih, nsplit)
            gx = gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
            raw, reserve = gru_seq_fwd_bf16(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, nsplit, need)
        else:
            x2 = x.view(T * B, I)
            if gru_proj_fused_ok(x2, w_ih, T, B, I, H, D):
                # projection tiles and recurrence in one launch: the recurrence starts as soon as its first rows exist
                raw, reserve = gru_proj_seq_fwd(x2, w_ih, b_ih, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, I, H, D, need)
            else:
                gx = gemm_nt(x2, w_ih, b_ih)                              # (T*B, D*3H); train_nsplit arithmetic
                raw, reserve = gru_seq_fwd(gx, w_hh_f, w_hh_r, b_hh_f, b_hh_r, T, B, H, D, need)
        offset, offset_dev, sub_batch = offset if isinstance(offset, tuple) else (offset, None, 0)
        if p == 0.0 and (factor == 1):
            y = raw
        else:
            y = dropout_pool_fwd(raw, mask, p, seed, offset, method, factor, offset_dev, sub_batch)
        if need:
            ctx.save_for_backward(x, raw, reserve, mask, w_ih, w_hh_f, w_hh_r)
            assert sub_batch == 0, "sub-batched dropout streams are for no-grad (frozen) stages"
            ctx.offset_dev = offset_dev          # not a graph tensor: a persistent 1-element buffer
            ctx.cfg = (T, B, I, H, D, p, seed, offset, method, factor)
        return y

    @staticmethod
    def backward(ctx, dy):
        T, B, I, H, D, p, seed, offset, method, factor = ctx.cfg
        x, raw, reserve, mask, w_ih, w_hh_f, w_hh_r = ctx.saved_tensors
        if p == 0.0 and factor == 1:
            d_raw = dy
        else:
            d_raw = dropout_pool_bwd(dy, raw, mask, p, seed, offset, method, factor, ctx.offset_dev)
        d_gx, d_gh, dbp = gru_seq_bwd(d_raw, reserve, w_hh_f, w_hh_r, T, B, H, D)
        ng = ctx.needs_input_grad
        # positions: 0 x | 1 w_ih 2 b_ih (storage, no grad) | 3 w_ih_f 4 w_ih_r 5 b_ih_f 6 b_ih_r |
        #            7 w_hh_f 8 b_hh_f 9 w_hh_r 10 b_hh_r
        need_bias = ng[5] or ng[6] or ng[8] or ng[10]
        x2 = x.view(T * B, I)
        g2 = d_gx.view(T * B, D * 3 * H)
        h2 = d_gh.view(T * B, D * 3 * H)
        r2 = raw.view(T * B, D * H)
        grads = [None] * 19
        dev = x.device
        small = (T * B <= TN_SMALL_ROWS and T > 1 and (ng[3] or ng[4]) and all(ng[7 + 2 * d] for d in range(D))
                 and I % 4 == 0 and H % 4 == 0)
        # long layers (thousands of rows) in exact-fp32 training: the same ONE launch with the k range split over workgroups
        # (slu_gemm_tn_batched_splitk) instead of three k-slow GEMMs + three reduce launches + a column sum
        long_rows = (not small and T > 1 and T * B >= TN_SPLITK_MIN_ROWS and (ng[3] or ng[4]) and all(ng[7 + 2 * d] for d in range(D))
                     and I % 4 == 0 and H % 4 == 0 and train_nsplit(True) == 0
                     and os.environ.get("SLU_TN_SPLITK", "1") != "0")
        if long_rows:
            n = (T - 1) * B
            long_rows = gemm_tn_splitk_ok([(g2, x2)] + [
                ((h2[:, d * 3 * H:(d + 1) * 3 * H][B:], r2[:n, :H]) if d == 0
                 else (h2[:, d * 3 * H:(d + 1) * 3 * H][:n], r2[B:, H:])) for d in range(D)])
        small = small or long_rows
        if need_bias and not small:
            # (tiles, D, 6H) per-tile partials -> (D, 6H): [d(b_ih) (3H) | d(b_hh) (3H)]  (slu_colsum_f32, deterministic)
            dbp = colsum(dbp.view(dbp.shape[0], D * 6 * H)).view(D, 6 * H)
        if small:
            # a few thousand rows (the intent layer of the look-ahead pipeline): every weight gradient of the layer
            # in ONE launch (no split-K workspaces, no reduce launches).  The choice depends on the shape only.
            n = (T - 1) * B
            # Round 5, fully trainable loops (_Fork.defer; wgrad_branch has the modes and the measurements): the launch goes to
            # an auxiliary stream / graph branch with a workgroup budget, beside this layer's data-gradient GEMM — or, mode
            # "pass", stays open until the trainer's single join after the whole backward pass.  What makes the open form safe:
            #  * every gradient it writes is a tensor of its own (not a view of a stacked buffer): AccumulateGrad takes such
            #    a tensor over without reading it — a view it would CLONE, on this stream, before the branch has written it;
            #  * the bias sums (views of one small buffer) are formed on THIS stream by slu_colsum_f32;
            #  * the operands are recorded on the branch's stream, so the allocator does not hand their memory out again
            #    before the branch has read them.
            # The budget (and with it the split count = the summation order) depends on the SHAPE only, so that every loop
            # kind — captured, eager, look-ahead with trainable long layers — produces the same bits; the branch itself only
            # where nothing runs beside the step (_Fork.defer).
            mode, budget = wgrad_branch() if long_rows else ("0", 0)
            branch = bool(budget) and _Fork.defer
            join_here = False
            outs = []
            if budget:
                dWs = [torch.empty(3 * H, I, dtype=torch.float32, device=dev) for _ in range(D)]
                probs = [(g2[:, d * 3 * H:(d + 1) * 3 * H], x2, dWs[d]) for d in range(D)]
            else:
                dW = torch.empty(D * 3 * H, I, dtype=torch.float32, device=dev)
                probs = [(g2, x2, dW)]
            for d in range(D):
                hd = h2[:, d * 3 * H:(d + 1) * 3 * H]
                ga, hp = (hd[B:], r2[:n, :H]) if d == 0 else (hd[:n], r2[B:, H:])
                dWh = torch.empty(3 * H, H, dtype=torch.float32, device=dev)
                probs.append((ga, hp, dWh))
                outs.append(dWh)
            rowsum = None
            if need_bias and budget:
                dbp = colsum(dbp.view(dbp.shape[0], D * 6 * H)).view(D, 6 * H)
            elif need_bias:                                # the per-tile bias partial sums, summed by the same launch
                db = torch.empty(dbp.shape[1:], dtype=torch.float32, device=dev)
                rowsum = (dbp.contiguous(), db)
                dbp = db
            if budget:
                fork = _Fork(dev, 0) if branch else contextlib.nullcontext()
                with fork:
                    gemm_tn_batched_splitk(probs, None, max_wg=budget)
                if branch and fork.active:
                    # operands AND results: the results are written on the branch's stream (in "pass" mode long after this
                    # function has returned), the allocator must not recycle them for the main stream before that
                    for t in (d_gx, d_gh, x, raw, *dWs, *outs):
                        t.record_stream(fork.side)
                    if mode == "layer":
                        join_here = True                 # after the data-gradient GEMM below
                grads[3] = dWs[0]
                if D == 2:
                    grads[4] = dWs[1]
            else:
                (gemm_tn_batched_splitk if long_rows else gemm_tn_batched)(probs, rowsum)
                grads[3] = dW[:3 * H]
                if D == 2:
                    grads[4] = dW[3 * H:]
            for d in range(D):
                grads[7 + 2 * d] = outs[d]
                if ng[8 + 2 * d]:
                    grads[8 + 2 * d] = dbp[d, 3 * H:]
            if ng[0]:
                grads[0] = _gru_dx(g2, w_ih, T, B, I, H, D)
            if ng[5]:
                grads[5] = dbp[0, :3 * H]
            if D == 2 and ng[6]:
                grads[6] = dbp[1, :3 * H]
            if join_here:
                _Fork.join(dev)
            return tuple(grads)
        # The weight-gradient GEMMs are independent of each other and of the data-gradient GEMM: they
        # run on auxiliary streams (graph branches under capture) while dx proceeds on this one.
        if ng[3] or ng[4]:                                 # dW_ih = d_gx^T x, one GEMM for both directions
            dW = torch.empty(D * 3 * H, I, dtype=torch.float32, device=dev)
            with _Fork(dev, 0):
                _wgrad(g2, x2, dW)
            grads[3] = dW[:3 * H]
            if D == 2:
                grads[4] = dW[3 * H:]
        for d in range(D):
            wpos, bpos = 7 + 2 * d, 8 + 2 * d              # (w_hh, b_hh) of direction d
            if ng[wpos]:                                   # dW_hh = d_gh^T h_{t-1}
                if T > 1:
                    n = (T - 1) * B
                    hd = h2[:, d * 3 * H:(d + 1) * 3 * H]
                    if d == 0:     # h_{t-1} = raw[t-1]: gradient rows t >= 1 against raw rows t-1
                        ga, hp = hd[B:], r2[:n, :H]
                    else:          # reverse scan: h_prev(t) = raw[t+1]
                        ga, hp = hd[:n], r2[B:, H:]
                    dWh = torch.empty(3 * H, H, dtype=torch.float32, device=dev)
                    with _Fork(dev, 1 + d):
                        _wgrad(ga, hp, dWh)
                    grads[wpos] = dWh
                else:
                    grads[wpos] = torch.zeros(3 * H, H, dtype=torch.float32, device=dev)
            if ng[bpos]:
                grads[bpos] = dbp[d, 3 * H:]
        if ng[0]:                                          # dx = d_gx W_ih (both directions, K = D*3H)
            grads[0] = _gru_dx(g2, w_ih, T, B, I, H, D)
        if ng[5]:
            grads[5] = dbp[0, :3 * H]
        if D == 2 and ng[6]:
            grads[6] = dbp[1, :3 * H]
        _Fork.join(dev)
        return tuple(grads)


# ------------------------------------------------------------------------------------------------
# seq2seq intent decoder (reference models.py:418-557): step kernels + one autograd Function with a manual BPTT
# ------------------------------------------------------------------------------------------------
def gru_cell_fwd(gi, gh, h_prev, h_out, save, drop_out, mask, p, seed, offset, offset_dev, idx_base):
    L = _lib.load()
    B, H = h_prev.shape
    assert gi.is_contiguous() and gh.is_contiguous() and h_prev.stride(1) == 1 and h_out.stride(1) == 1
    _lib.check(L.slu_gru_cell_fwd(gi.data_ptr(), gh.data_ptr(), h_prev.data_ptr(), h_prev.stride(0), h_out.data_ptr(),
                                  h_out.stride(0), _ptr(save), _ptr(drop_out), _ptr(mask), float(p), int(seed), int(offset),
                                  _ptr(offset_dev), int(idx_base), B, H, _stream()), "slu_gru_cell_fwd")


def gru_cell_bwd(d_h, d_drop, save, h_prev, d_gi, d_gh, d_h_prev, mask, p, seed, offset, offset_dev, idx_base):
    L = _lib.load()
    B, H = h_prev.shape
    _lib.check(L.slu_gru_cell_bwd(d_h.data_ptr(), d_h.stride(0), _ptr(d_drop), save.data_ptr(), h_prev.data_ptr(),
                                  h_prev.stride(0), d_gi.data_ptr(), d_gh.data_ptr(), d_h_prev.data_ptr(), d_h_prev.stride(0),
                                  _ptr(mask), float(p), int(seed), int(offset), _ptr(offset_dev), int(idx_base), B, H,
                                  _stream()), "slu_gru_cell_bwd")


def attention_fwd(keys, values, query, ctx, weights, inv_scale):
    """keys (T, B, Kd) / values (T, B, Vd) time-major contiguous; query (B, Kd), ctx (B, Vd) row-strided views."""
    L = _lib.load()
    T, B, Kd = keys.shape
    Vd = values.shape[2]
    _lib.check(L.slu_attention_fwd(keys.data_ptr(), keys.stride(0), keys.stride(1), values.data_ptr(), values.stride(0),
                                   values.stride(1), query.data_ptr(), query.stride(0), ctx.data_ptr(), ctx.stride(0),
                                   weights.data_ptr(), float(inv_scale), B, T, Kd, Vd, _stream()), "slu_attention_fwd")


def attention_bwd(keys, values, query, d_ctx, weights, d_keys, d_values, d_query, inv_scale):
    L = _lib.load()
    T, B, Kd = keys.shape
    Vd = values.shape[2]
    assert d_keys.stride() == keys.stride() and d_values.stride() == values.stride()
    _lib.check(L.slu_attention_bwd(keys.data_ptr(), keys.stride(0), keys.stride(1), values.data_ptr(), values.stride(0),
                                   values.stride(1), query.data_ptr(), query.stride(0), d_ctx.data_ptr(), d_ctx.stride(0),
                                   weights.data_ptr(), d_keys.data_ptr(), d_values.data_ptr(), d_query.data_ptr(),
                                   d_query.stride(0), float(inv_scale), B, T, Kd, Vd, _stream()), "slu_attention_bwd")


def logsoftmax_dot_fwd(logits, y_u, logp_acc, lse):
    L = _lib.load()
    B, V = logits.shape
    assert logits.is_contiguous() and y_u.stride(1) == 1
    _lib.check(L.slu_logsoftmax_dot_fwd(logits.data_ptr(), y_u.data_ptr(), y_u.stride(0), logp_acc.data_ptr(), _ptr(lse),
                                        B, V, _stream()), "slu_logsoftmax_dot_fwd")


def logsoftmax_dot_bwd(logits, y_u, lse, g, d_logits):
    L = _lib.load()
    B, V = logits.shape
    _lib.check(L.slu_logsoftmax_dot_bwd(logits.data_ptr(), y_u.data_ptr(), y_u.stride(0), lse.data_ptr(), g.data_ptr(), 1,
                                        d_logits.data_ptr(), B, V, _stream()), "slu_logsoftmax_dot_bwd")


def broadcast_rows(src, dst2d):
    L = _lib.load()
    rows, n = dst2d.shape
    assert src.numel() == n and src.is_contiguous() and dst2d.stride(1) == 1
    _lib.check(L.slu_broadcast_rows_f32(src.data_ptr(), dst2d.data_ptr(), dst2d.stride(0), rows, n, _stream()),
               "slu_broadcast_rows_f32")


def decoder_step(P, keys, values, state_prev, state_next, y_prev, q, inp0, att_w, gi, gh, save, drop, logits, step,
                 drop_cfg):
    """One decoding step (models.py:528-536) on the HIP kernels, shared by the teacher-forced forward and the beam
    search: attention on the top layer's state, embedding of the previous label, the GRUCell stack (+ dropout between
    the cells), output logits.  P: parameter dict (detached tensors); state_* (B, L, Dd); save / drop: per-layer
    buffers or None (inference); drop_cfg = (p, masks or None, seed, offset, offset_dev, B * Dd).
    y_prev None: the embedding half of inp0 is already filled; logits None: the caller computes them later (teacher
    forcing knows every input label up front and needs the logits only after the loop: both become ONE GEMM over all
    steps instead of one small launch per step)."""
    Lc, Dd = state_prev.shape[1], state_prev.shape[2]
    E = P["embed.weight"].shape[0]
    p, masks, seed, offset, offset_dev, bd = drop_cfg
    # everything that depends on the previous state only, in ONE grouped launch: the attention query and every cell's
    # hidden-to-hidden product (gh is (Lc, B, 3 Dd))
    group = [(state_prev[:, Lc - 1], P["query.weight"], P["query.bias"], q, 0, 0)]
    group += [(state_prev[:, l], P["w_hh%d" % l], P["b_hh%d" % l], gh[l], 0, 0) for l in range(Lc)]
    gemm_small_batched(group)
    attention_fwd(keys, values, q, inp0[:, E:], att_w, P["inv_scale"])
    if y_prev is not None:
        gemm(y_prev, P["embed.weight"].t(), P["embed.bias"], out=inp0[:, :E])
    x_in = inp0
    for l in range(Lc):
        gemm_small_batched([(x_in, P["w_ih%d" % l], P["b_ih%d" % l], gi, 0, 0)])
        last = l == Lc - 1
        mask = None if (masks is None or last) else masks["decoder_dropout_u%d_l%d" % (step, l)]
        gru_cell_fwd(gi, gh[l], state_prev[:, l], state_next[:, l], None if save is None else save[l],
                     None if last else drop[l], mask, 0.0 if last else p, seed, offset, offset_dev,
                     (step * Lc + l) * bd)
        if not last:
            x_in = drop[l]
    if logits is not None:
        gemm_small_batched([(state_next[:, Lc - 1], P["linear.weight"], P["linear.bias"], logits, 0, 0)])


class Seq2SeqDecoderFn(torch.autograd.Function):
    """Seq2SeqDecoder.forward (reference models.py:504-557) + the loss of Model.forward's seq2seq branch (:825-828):
    teacher-forced log p(y | x) of every utterance and loss = -mean.  enc time-major (T, B, 2 * encoder_dim); y
    (B, U, V) float (one-hot rows, padded with <eos>).  One autograd node for the whole decoder: the backward is a
    hand-written BPTT over the U steps on the same kernels (every Linear's weight gradient is ONE GEMM over the
    (U * B)-row history).  Returns (loss_acc (2) = [loss, 0], log_p (B)); only loss_acc[0] carries a gradient."""
    NAMES = None    # set per call: parameter names in argument order

    @staticmethod
    def forward(ctx, enc, y, meta, *params):
        names, SOS, p, masks, seed, offset, offset_dev = meta
        P = {n: t.detach() for n, t in zip(names, params)}
        dev = enc.device
        enc = _f32c(enc, "encoder output")
        y = _f32c(y.to(dev), "y")
        T, B, E2 = enc.shape
        _, U, V = y.shape
        Lc = sum(1 for n in names if n.startswith("w_ih"))
        Dd = P["w_hh0"].shape[1]
        E, Kd, Vd = P["embed.weight"].shape[0], P["key.weight"].shape[0], P["value.weight"].shape[0]
        P["inv_scale"] = 1.0 / float(torch.sqrt(torch.tensor(Kd).float()))          # models.py:421
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        enc2 = enc.view(T * B, E2)
        keys = gemm(enc2, P["key.weight"].t(), P["key.bias"]).view(T, B, Kd)
        values = gemm(enc2, P["value.weight"].t(), P["value.bias"]).view(T, B, Vd)
        state = f(U + 1, B, Lc, Dd)
        broadcast_rows(P["initial_state"].contiguous().view(-1), state[0].view(B, Lc * Dd))
        ycat = f(U + 1, B, V)                                # time-major labels behind a <sos> row:
        ycat[0].zero_()                                      #   yprev[u] = ycat[u]     = the label fed at step u
        ycat[0, :, SOS] = 1.0                                #   y_tm[u]  = ycat[u + 1] = the label scored at step u
        ycat[1:].copy_(y.transpose(0, 1))
        yprev, y_tm = ycat[:U], ycat[1:]
        q, att_w, inp0 = f(U, B, Kd), f(U, B, T), f(U, B, E + Vd)
        save = f(U, Lc, 4, B, Dd)
        drop = f(max(Lc - 1, 1), U, B, Dd)
        logits, lse = f(U, B, V), f(U, B)
        logp = torch.zeros(B, dtype=torch.float32, device=dev)
        gi, gh = f(B, 3 * Dd), f(Lc, B, 3 * Dd)
        dcfg = (p, masks, seed, offset, offset_dev, B * Dd)
        # teacher forcing: the embeddings of ALL steps' input labels in one GEMM (rows (u, b)), straight into inp0
        gemm(yprev.view(U * B, V), P["embed.weight"].t(), P["embed.bias"], out=inp0.view(U * B, E + Vd)[:, :E])
        for u in range(U):
            decoder_step(P, keys, values, state[u], state[u + 1], None, q[u], inp0[u], att_w[u], gi, gh, save[u],
                         [drop[l][u] for l in range(Lc - 1)], None, u, dcfg)
        # ... and the output layer of all steps in one GEMM over the state history; the per-step scores are then added
        # in step order (the reference's running sum, models.py:540)
        top = slice((Lc - 1) * Dd, Lc * Dd)
        gemm(state[1:].view(U * B, Lc * Dd)[:, top], P["linear.weight"].t(), P["linear.bias"], out=logits.view(U * B, V))
        lp = torch.zeros(U, B, dtype=torch.float32, device=dev)               # log p(y_u | ...) of every (step, utterance)
        logsoftmax_dot_fwd(logits.view(U * B, V), y_tm.reshape(U * B, V), lp.view(-1), lse.view(-1))
        colsum(lp, out=logp)                                                  # log p(y | x) = sum over the steps
        loss_acc = torch.zeros(2, dtype=torch.float32, device=dev)
        L = _lib.load()
        _lib.check(L.slu_neg_mean_f32(logp.data_ptr(), loss_acc.data_ptr(), B, _stream()), "slu_neg_mean_f32")
        ctx.meta = (names, p, masks, seed, offset, offset_dev, (T, B, E2, U, V, Lc, Dd, E, Kd, Vd), P["inv_scale"])
        ctx.save_for_backward(enc, ycat, yprev, keys, values, state, q, att_w, inp0, save, drop, logits, lse, *params)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(logp)
        return loss_acc, logp

    @staticmethod
    def backward(ctx, d_loss_acc, _d_logp):
        names, p, masks, seed, offset, offset_dev, dims, inv_scale = ctx.meta
        T, B, E2, U, V, Lc, Dd, E, Kd, Vd = dims
        n_par = len(names)
        if d_loss_acc is None:
            return (None,) * (3 + n_par)
        saved = ctx.saved_tensors
        enc, ycat, yprev, keys, values, state, q, att_w, inp0, save, drop, logits, lse = saved[:13]
        y_tm = ycat[1:]
        P = {n: t.detach() for n, t in zip(names, saved[13:])}
        dev = enc.device
        L = _lib.load()
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        g = _f32c(d_loss_acc.float(), "d_loss")                       # (2): only [0] matters
        dlogp = f(U * B)                                             # d loss / d log p(y_u | ...) = -g / B for every row
        _lib.check(L.slu_fill_scaled_f32(dlogp.data_ptr(), U * B, g.data_ptr(), -1.0 / B, _stream()), "slu_fill_scaled_f32")
        d_state = torch.zeros(B, Lc, Dd, dtype=torch.float32, device=dev)
        d_keys, d_values = torch.zeros_like(keys), torch.zeros_like(values)
        d_gi, d_gh = f(Lc, U, B, 3 * Dd), f(Lc, U, B, 3 * Dd)
        d_logits, d_q, d_inp0 = f(U, B, V), f(U, B, Kd), f(U, B, E + Vd)
        d_x = f(B, Dd)                                              # gradient w.r.t. the dropped input of layer l + 1
        # every step's d loss / d logits, and its way into the top layer's state, in ONE GEMM over all steps (the loop then
        # adds slice u where the reference's graph adds it: as an extra gradient of the top cell's output at step u)
        logsoftmax_dot_bwd(logits.view(U * B, V), y_tm.reshape(U * B, V), lse.view(-1), dlogp, d_logits.view(U * B, V))
        g_top = gemm(d_logits.view(U * B, V), P["linear.weight"]).view(U, B, Dd)
        for u in range(U - 1, -1, -1):
            for l in range(Lc - 1, -1, -1):
                last = l == Lc - 1
                mask = None if (masks is None or last) else masks["decoder_dropout_u%d_l%d" % (u, l)]
                # top cell: d_drop = this step's logits gradient (keep-factor 1: there is no dropout after the last cell)
                gru_cell_bwd(d_state[:, l], g_top[u] if last else d_x, save[u, l], state[u][:, l], d_gi[l, u], d_gh[l, u],
                             d_state[:, l], mask, 0.0 if last else p, seed, offset, offset_dev, (u * Lc + l) * B * Dd)
                # one grouped launch: the recurrent part of d h_{u-1} (accumulated) and the gradient of the cell's input
                gemm_small_batched([(d_gh[l, u], P["w_hh%d" % l], None, d_state[:, l], 1, 1),
                                    (d_gi[l, u], P["w_ih%d" % l], None, d_x if l > 0 else d_inp0[u], 1, 0)])
            attention_bwd(keys, values, q[u], d_inp0[u][:, E:], att_w[u], d_keys, d_values, d_q[u], inv_scale)
            gemm_small_batched([(d_q[u], P["query.weight"], None, d_state[:, Lc - 1], 1, 1)])
        grads = {}
        st_prev = state[:U].view(U * B, Lc * Dd)                     # rows (u, b): the state BEFORE step u
        st_next = state[1:].view(U * B, Lc * Dd)
        top = slice((Lc - 1) * Dd, Lc * Dd)
        dl2, dq2, di2 = d_logits.view(U * B, V), d_q.view(U * B, Kd), d_inp0.view(U * B, E + Vd)
        grads["linear.weight"], grads["linear.bias"] = gemm(dl2.t(), st_next[:, top]), colsum(dl2)
        grads["query.weight"], grads["query.bias"] = gemm(dq2.t(), st_prev[:, top]), colsum(dq2)
        grads["embed.weight"], grads["embed.bias"] = gemm(di2[:, :E].t(), yprev.view(U * B, V)), colsum(di2[:, :E])
        for l in range(Lc):
            gi2, gh2 = d_gi[l].view(U * B, 3 * Dd), d_gh[l].view(U * B, 3 * Dd)
            x_l = inp0.view(U * B, E + Vd) if l == 0 else drop[l - 1].view(U * B, Dd)
            grads["w_ih%d" % l], grads["b_ih%d" % l] = gemm(gi2.t(), x_l), colsum(gi2)
            grads["w_hh%d" % l], grads["b_hh%d" % l] = gemm(gh2.t(), st_prev[:, l * Dd:(l + 1) * Dd]), colsum(gh2)
        grads["initial_state"] = colsum(d_state.view(B, Lc * Dd)).view(Lc, Dd)
        dk2, dv2, enc2 = d_keys.view(T * B, Kd), d_values.view(T * B, Vd), enc.view(T * B, E2)
        grads["key.weight"], grads["key.bias"] = gemm(dk2.t(), enc2), colsum(dk2)
        grads["value.weight"], grads["value.bias"] = gemm(dv2.t(), enc2), colsum(dv2)
        d_enc = None
        if ctx.needs_input_grad[0]:
            d_enc = gemm(dk2, P["key.weight"])
            gemm(dv2, P["value.weight"], out=d_enc, accumulate=True)
            d_enc = d_enc.view(T, B, E2)
        return (d_enc, None, None) + tuple(grads[n] if ctx.needs_input_grad[3 + i] else None for i, n in enumerate(names))
