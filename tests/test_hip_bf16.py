"""GPU: the split-precision (16-bit MFMA) kernels of the frozen encoder stages (csrc/slu_bf16.h) against
float64 references.  nsplit = 3 (bf16x3: three bf16 terms, six products) and nsplit = 2 (f16x2: two fp16 terms,
three products, the default of the frozen stages) must be fp32-class (the error of an exact fp32 fmaf chain is
~1e-7 of sum |a b|; a few 2^-24 | 2^-22 per product here), nsplit = 1 is plain bf16 (2^-9 per operand)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))


@pytest.fixture(scope="module")
def ops():
    from slu_hip import lib, ops as _ops
    lib.require_gfx950()
    return _ops


def test_split_planes_are_exact(ops):
    torch.manual_seed(0)
    x = (torch.randn(37, 60) * torch.logspace(-3, 3, 60)).cuda()
    pl = ops.split_bf16(x, 3)
    assert pl.shape == (3, 37, 64) and pl.dtype == torch.bfloat16
    assert torch.equal(pl[:, :, 60:].float(), torch.zeros(3, 37, 4, device="cuda"))
    back = (pl[0, :, :60].double() + pl[1, :, :60].double() + pl[2, :, :60].double())
    assert torch.equal(back.float(), x)                      # three bf16 terms carry all 24 bits
    one = ops.split_bf16(x, 1)
    assert torch.equal(one[0, :, :60], x.to(torch.bfloat16))  # round to nearest even


def test_split_f16x2_planes(ops):
    """f16x2: x = hi + 2^-11 lo to 2^-22 |x| wherever hi is a normal fp16; below fp16's smallest normal hi is zero and
    lo alone carries the value (absolute error <= 2^-26); beyond 65504 the scheme is out of range (documented)."""
    torch.manual_seed(1)
    x = (torch.randn(53, 40) * torch.logspace(-3, 3, 40)).cuda()
    x[0, :8] = torch.tensor([0.0, 1.0, -1.0, 6.1e-5, -3.0e-5, 1.0e-7, 65000.0, -2.5e-9], device="cuda")
    pl = ops.split_bf16(x, 2)
    assert pl.shape == (2, 53, 64) and pl.dtype == torch.float16
    assert torch.equal(pl[:, :, 40:].float(), torch.zeros(2, 53, 24, device="cuda"))
    hi, lo = pl[0, :, :40].double(), pl[1, :, :40].double()
    assert torch.isfinite(hi).all() and torch.isfinite(lo).all()
    back = hi + lo / 2048.0
    xd = x.double()
    err = (back - xd).abs()
    normal = xd.abs() >= 2.0 ** -14
    assert (err[normal] <= xd.abs()[normal] * 2.0 ** -22).all()
    assert (err[~normal] <= 2.0 ** -26).all()
    assert (hi[~normal] == 0).all()                          # nothing rests on fp16 denormals in the dominant term
    assert ((hi.abs() >= 2.0 ** -14) | (hi == 0)).all()


@pytest.mark.parametrize("M,N,K", [(1000, 768, 60), (4097, 768, 256), (130, 384, 256), (64, 128, 33), (333, 192, 20), (129, 64, 60), (140000, 128, 100), (131073, 192, 256)])
@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_gemm_bf16_vs_float64(ops, M, N, K, nsplit):
    torch.manual_seed(M + K)
    a = torch.randn(M, K)
    w = torch.randn(N, K) * 0.2
    bias = torch.randn(N)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.abs().double() @ w.abs().double().t()).max().item()
    out = ops.gemm_bf16(ops.split_bf16(a.cuda(), nsplit), ops.gemm_bf16_pack(w.cuda(), nsplit), bias.cuda(), N, K)
    err = (out.cpu().double() - ref).abs().max().item() / scale
    exact = (a.cuda() @ w.cuda().t() + bias.cuda()).cpu().double()
    err_f32 = (exact - ref).abs().max().item() / scale
    print("gemm_bf16 nsplit=%d M=%d N=%d K=%d: rel err %.2e (torch fp32 GEMM: %.2e)" % (nsplit, M, N, K, err, err_f32))
    assert err <= {3: 4e-7, 2: 6e-7, 1: 2e-2}[nsplit]


@pytest.mark.parametrize("M,N,K,wt", [(1216, 768, 256, False), (19200, 768, 60, False), (19200, 60, 768, True), (130, 100, 36, False),
                                      (4097, 1000, 256, False), (300, 256, 10000, True), (64, 64, 4, True)])
@pytest.mark.parametrize("nsplit", [2, 3, 1])
def test_gemm_a32_vs_float64(ops, M, N, K, wt, nsplit):
    """slu_gemm_bf16_a32 (GEMMs of trainable layers: fp32 A split on the fly, packed W or W^T in place) against float64:
    forward projection (K = 60 / 256), data gradient through the transposed weight view (N = 60, K = 768), an ASR head
    (N = 1000 | K = 10 000), ragged M / N / K, A as a column slice of a wider matrix."""
    torch.manual_seed(M + N)
    abig = torch.randn(M, K + 8)
    a = abig[:, 4:4 + K]                                       # row stride K + 8, 16-byte aligned
    w = torch.randn(N, K) * 0.2
    bias = torch.randn(N)
    ref = a.double() @ w.double().t() + bias.double()
    scale = (a.abs().double() @ w.abs().double().t()).max().item()
    wd = w.cuda() if not wt else w.t().contiguous().cuda().t()   # (N, K) view of a (K, N) tensor: k is the slow index
    ad = abig.cuda()[:, 4:4 + K]
    assert ops.gemm_a32_ok(ad, N, K)
    out = ops.gemm_a32(ad, ops.gemm_bf16_pack(wd, nsplit), bias.cuda(), N, nsplit)
    torch.cuda.synchronize()
    err = (out.cpu().double() - ref).abs().max().item() / scale
    print("gemm_a32 nsplit=%d M=%d N=%d K=%d%s: rel err %.2e" % (nsplit, M, N, K, " (W^T view)" if wt else "", err))
    assert err <= {3: 4e-7, 2: 6e-7, 1: 2e-2}[nsplit]
    assert not ops.gemm_a32_ok(ad[:, 1:], N, K - 1)


@pytest.mark.parametrize("K,M,N", [(32000, 768, 256), (9568, 384, 128), (300, 64, 64), (4097, 128, 192), (31, 64, 128),
                                   (1900, 1000, 256), (700, 60, 36)])
@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_gemm_tn_bf16_vs_float64(ops, K, M, N, nsplit):
    """slu_gemm_tn_bf16 (weight gradients of trainable layers): C = A^T B with k-major fp32 operands split (f16x2, the
    default) or rounded to bf16 (bf16 mode) in the staging, fp32 accumulation, deterministic split-K — against float64
    on the fp32 operands and, for bf16, on the bf16-rounded operands (exact up to fp32 accumulation), incl. column-slice
    views (row stride > width) and M / N that are not multiples of the 64 x 64 tile."""
    torch.manual_seed(K + M)
    abig = torch.randn(K + 3, M + 64, device="cuda")
    bbig = torch.randn(K + 3, N + 128, device="cuda")
    a, b = abig[3:, 64:], bbig[:K, 128:]                       # offset / strided views, 16-byte aligned
    assert ops.gemm_tn_bf16_ok(a, b)
    out = ops.gemm_tn_bf16(a, b, None, nsplit)
    out2 = ops.gemm_tn_bf16(a, b, None, nsplit)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                              # deterministic
    ref = a.double().t() @ b.double()
    scale = (a.abs().double().t() @ b.abs().double()).max().item()
    e = (out.double() - ref).abs().max().item() / scale
    if nsplit == 1:
        ar, br = a.to(torch.bfloat16).double(), b.to(torch.bfloat16).double()
        e_r = (out.double() - ar.t() @ br).abs().max().item() / scale
        print("gemm_tn_bf16 K=%d M=%d N=%d: %.2e of sum|a||b| vs the bf16-rounded operands, %.2e vs fp32 operands" % (K, M, N, e_r, e))
        assert e_r <= 2e-6 and e <= 1e-2
    else:
        print("gemm_tn nsplit=%d K=%d M=%d N=%d: %.2e of sum|a||b| vs float64" % (nsplit, K, M, N, e))
        assert e <= 6e-7
    assert not ops.gemm_tn_bf16_ok(a[:, :M - 3], b)


@pytest.mark.parametrize("T,B,H", [(40, 64, 128), (23, 37, 128), (9, 5, 64), (300, 768, 128)])
@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_gru_bf16_vs_exact_fp32_kernel(ops, T, B, H, nsplit):
    """slu_gru_seq_fwd_bf16 against the exact-fp32 persistent kernel (itself held to the oracle in
    test_hip_ops / test_hip_bench_path): nsplit = 3 must agree to fp32 round-off accumulated over T steps."""
    torch.manual_seed(T + B)
    D = 2
    gx = torch.randn(T, B, D * 3 * H).cuda()
    k = 1.0 / H ** 0.5
    wf, wr = ((torch.rand(3 * H, H) * 2 - 1) * k).cuda(), ((torch.rand(3 * H, H) * 2 - 1) * k).cuda()
    bf, br = ((torch.rand(3 * H) * 2 - 1) * k).cuda(), ((torch.rand(3 * H) * 2 - 1) * k).cuda()
    ref, _ = ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, False)
    out, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit)
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    print("gru bf16 nsplit=%d T=%d B=%d H=%d: max-abs deviation from the fp32 kernel %.2e" % (nsplit, T, B, H, err))
    assert err <= (5e-6 if nsplit >= 2 else 5e-2)


@pytest.mark.parametrize("ns", [2, 3])
@pytest.mark.parametrize("T,B,I", [(40, 64, 60), (23, 37, 60), (7, 16, 32), (300, 1024, 60)])
def test_gru_fused_input_projection_equals_gemm_plus_recurrence(ops, T, B, I, ns):
    """slu_gru_seq_fwd_bf16(x_planes): the recurrence computes x W_ih^T + b_ih itself (first GRU layer, K <= 64, f16x2) with
    the projection GEMM's accumulation order — bit-identical to slu_gemm_bf16 followed by the plain recurrence, for both
    directions, ragged batches (B not a multiple of the 16-sequence tile) and the 1024-sequence super-batch."""
    torch.manual_seed(T + B)
    H, D = 128, 2
    x = torch.randn(T * B, I, device="cuda")
    w_ih = torch.randn(D * 3 * H, I, device="cuda") * 0.1
    b_ih = torch.randn(D * 3 * H, device="cuda") * 0.1
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda") * 0.1, torch.randn(3 * H, device="cuda") * 0.1
    planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
    gx = ops.gemm_bf16(planes, packed, b_ih, D * 3 * H, I)
    ref, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, ns)
    out, _ = ops.gru_seq_fwd_bf16(None, wf, wr, bf, br, T, B, H, D, ns, False, fused=(planes, I, packed, b_ih))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    exact, _ = ops.gru_seq_fwd(ops.gemm(x, w_ih.t(), b_ih), wf, wr, bf, br, T, B, H, D, False)
    assert (out - exact).abs().max().item() <= 5e-6


@pytest.mark.parametrize("T,B,H,D,sub", [(75, 128, 128, 2, 64), (38, 37, 128, 2, 0), (300, 1024, 128, 2, 64), (9, 16, 64, 2, 0),
                                          (20, 48, 128, 1, 16)])
@pytest.mark.parametrize("nsplit", [2, 3])
def test_gru_dropout_pool_epilogue_equals_the_two_launch_path(ops, T, B, H, D, sub, nsplit):
    """slu_gru_seq_fwd_pool_bf16: Dropout(p) + Downsample("avg", 2) of a frozen layer (models.py:246-251, 26-46) applied in the
    recurrence's epilogue from the 1-bit mask of slu_dropout_bits — bit-identical to recurrence + slu_dropout_pool_fwd /
    _fwd_planes with the same (seed, offset, sub-batch) Philox stream: odd T (partial last window), ragged tiles, both
    directions and a unidirectional layer, eval mode (p = 0), the 1024-sequence super-batch, fp32 and plane outputs."""
    torch.manual_seed(T * 7 + B)
    gx = torch.randn(T, B, D * 3 * H, device="cuda")
    wf, bf = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1
    wr, br = (torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1) if D == 2 else (None, None)
    raw, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit)
    off_dev = torch.tensor([5 * 16], dtype=torch.int64, device="cuda")
    for p, offset, odev in ((0.5, 7 * 16 + 3, None), (0.25, 3, off_dev), (0.0, 0, None)):
        keep = ops.dropout_bits(T, B, D * H, p, 1234, offset, odev, sub, gx.device) if p > 0 else None
        two_f = ops.dropout_pool_fwd(raw, None, p, 1234, offset, "avg", 2, odev, sub, keep_bits=keep)
        two_p = ops.dropout_pool_fwd_planes(raw, None, p, 1234, offset, "avg", 2, nsplit, odev, sub, keep_bits=keep).planes
        one_f = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, False)
        one_p = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, True).planes
        torch.cuda.synchronize()
        assert torch.equal(one_f, two_f), "fp32 output, p = %g" % p
        assert torch.equal(one_p.view(torch.int16), two_p.view(torch.int16)), "plane output, p = %g" % p
        if p > 0:
            # the bit stream is the mask itself: element (t, b, c) is dropped in the two-launch output iff its bit is clear
            bits = keep.view(T, B, D * H // 32, 1).bitwise_right_shift(torch.arange(32, device="cuda")).bitwise_and(1)
            kept = bits.view(T, B, D * H).bool()
            full = ops.dropout_pool_fwd(raw, None, p, 1234, offset, "none", 1, odev, sub, keep_bits=keep)
            assert torch.equal(full != 0, kept & (raw != 0))
            assert abs(kept.float().mean().item() - (1 - p)) < 0.01
            if sub:      # every sub-batch (training step) draws its own mask, and so does every frame
                assert not torch.equal(kept[:, :sub], kept[:, sub:2 * sub]) and not torch.equal(kept[0], kept[1])
                # the mask of sub-batch k is the mask a stand-alone batch would draw on stream offset + 16 k
                alone = ops.dropout_bits(T, sub, D * H, p, 1234, offset + 16, odev, 0, gx.device)
                assert torch.equal(alone, keep[:, sub:2 * sub])


@pytest.mark.parametrize("T,B,D", [(75, 128, 2), (38, 37, 2), (1, 70, 2), (2, 33, 1), (301, 2560, 2), (150, 1296, 2)])
@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_gru_two_tiles_per_workgroup_equals_one_tile(ops, T, B, D, nsplit):
    """gru_bf2_fwd_kernel (round 6: two 16-sequence tiles per workgroup half a step apart, gx and keep bits by LDS-DMA through a
    ring of three chunk buffers; slu_gru_seq_fwd[_pool]_bf16(seq_tiles = 2)) against gru_bf_fwd_kernel (seq_tiles = 1): same
    products in the same order, same gate formulas, same epilogue -> BIT-IDENTICAL outputs, for all three output forms
    (fp32 every step; Dropout + avg-pool -> fp32; -> planes), odd T (partial last pooling window), T = 1 / 2 (no steady-state
    loop iteration), ragged batches (the last workgroup's second tile partly or wholly past B), a unidirectional layer, and
    the 40-batch super-batch of the benchmarked loop (2560 sequences, T = 301)."""
    torch.manual_seed(T * 11 + B)
    H = 128
    gx = torch.randn(T, B, D * 3 * H, device="cuda")
    wf, bf = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1
    wr, br = (torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, device="cuda") * 0.1) if D == 2 else (None, None)
    one, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, seq_tiles=1)
    two, _ = ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, seq_tiles=2)
    torch.cuda.synchronize()
    assert torch.equal(one, two), "fp32 output of every step: %.3e" % (one - two).abs().max().item()
    for p, offset in ((0.5, 7 * 16 + 3), (0.0, 0)):
        keep = ops.dropout_bits(T, B, D * H, p, 1234, offset, None, 64 if B % 64 == 0 else 0, gx.device) if p > 0 else None
        a_f = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, False, seq_tiles=1)
        b_f = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, False, seq_tiles=2)
        a_p = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, True, seq_tiles=1).planes
        b_p = ops.gru_seq_fwd_pool_bf16(gx, wf, wr, bf, br, T, B, H, D, nsplit, keep, p, True, seq_tiles=2).planes
        torch.cuda.synchronize()
        assert torch.equal(a_f, b_f), "pooled fp32 output, p = %g: %.3e" % (p, (a_f - b_f).abs().max().item())
        assert torch.equal(a_p.view(torch.int16), b_p.view(torch.int16)), "plane output, p = %g" % p
    # opt-in: the default is one tile; SLU_GRU_TILES=auto takes two as soon as the one-tile grid exceeds the device's CUs
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert ops.gru_seq_tiles(B, H, D) == 1
    os.environ["SLU_GRU_TILES"] = "auto"
    try:
        assert ops.gru_seq_tiles(B, H, D) == (2 if -(-B // 16) * D > cus else 1)
        assert ops.gru_seq_tiles(B, 64, D) == 1 and ops.gru_seq_tiles(B, H, D, reserve=True) == 1
    finally:
        del os.environ["SLU_GRU_TILES"]


def test_bf16_mode_full_model_vs_fp32_oracle(tmp_path, monkeypatch):
    """BASELINE configs[4] arithmetic (SLU_DTYPE=bf16: the GRU layers' forward contractions on bf16 MFMA with
    fp32 accumulation and gate math, exact-fp32 backward on the saved gates).  The reference has no reduced
    precision path (plain fp32 nn.GRU, models.py:232/:262/:686), so the yardstick is the fp32 oracle:
    predicted intents must agree, the logit deviation is reported and bounded empirically, gradients must point
    the same way."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import slu_oracle as O
    import data
    import models
    cfg = O.OracleConfig(pretraining_type=0)                  # full-size no_unfreezing architecture, all trainable
    cfg.folder = str(tmp_path)
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
    torch.manual_seed(5)
    monkeypatch.setenv("SLU_DTYPE", "bf16")
    model = models.Model(cfg)
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x = 0.1 * torch.randn(8, 16000, generator=g)
    y = torch.stack([torch.randint(0, n, (8,), generator=g) for n in cfg.values_per_slot], dim=1)
    masks = O.draw_dropout_masks(cfg, x, seed=21)
    models.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    try:
        model.train()
        loss, acc = model(x, y)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        models.set_dropout_masks(None)
    rloss, racc, rlogits, rpred = O.slu_forward(sd, x, y, cfg, masks, explicit_gru=False)
    rloss.backward()
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
        _, _, elogits, epred = O.slu_forward({k: v.detach() for k, v in sd.items()}, x, y, cfg, None, explicit_gru=False)
    dev = (logits.cpu() - elogits).abs().max().item()
    print("bf16 mode: eval logits max-abs deviation vs fp32 oracle %.3e (logit range %.2f), train loss %.5f vs %.5f"
          % (dev, elogits.abs().max().item(), loss.item(), rloss.item()))
    assert torch.equal(pred.cpu(), epred)                      # intent decisions unchanged
    assert dev <= 2e-2 and abs(loss.item() - rloss.item()) <= 2e-2
    worst, worst_name, dots, na, nb = 1.0, "", 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        if sd[k].grad is None or p.grad is None:
            continue
        a, b = p.grad.cpu().double().flatten(), sd[k].grad.double().flatten()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        if cos < worst:
            worst, worst_name = cos, k
        dots, na, nb = dots + (a @ b).item(), na + (a @ a).item(), nb + (b @ b).item()
    total = dots / (na ** 0.5 * nb ** 0.5)
    print("bf16 mode: gradient cosine vs fp32 oracle: whole model %.5f, worst tensor %.5f (%s)" % (total, worst, worst_name))
    assert total >= 0.97 and worst >= 0.9                      # bounds set empirically (measured 0.985 / 0.963)


def test_bf16_mode_at_configs4_shape_vs_fp32_oracle(tmp_path, monkeypatch):
    """BASELINE configs[4] at ITS shape: 10 s utterances (GRU lengths 1000 / 500 / 250 / 125 / 63), B = 32 per GPU, every
    layer trainable (unfreeze_all_layers end state), SLU_DTYPE=bf16 — bf16 operands in every forward contraction
    (Sinc / conv blocks, input projections, recurrences) and in the data-gradient contractions, fp32 accumulation, gate
    math, weight gradients.  The reference is fp32 only (models.py:190-286), so the yardstick is the fp32 oracle with
    the same dropout masks: predicted intents identical, eval logits / train loss within the bound bf16's 2^-9 operand
    rounding gives over 1000 recurrent steps (measured, printed), gradients pointing the same way tensor by tensor."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import slu_oracle as O
    import data
    import models
    from slu_hip import ops as _ops
    cfg = O.OracleConfig(pretraining_type=0)
    cfg.folder = str(tmp_path)
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
    torch.manual_seed(6)
    monkeypatch.setenv("SLU_DTYPE", "bf16")
    assert _ops.bf16_mode()
    model = models.Model(cfg)
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    B, T = 32, 160000
    x = 0.1 * torch.randn(B, T, generator=g)
    y = torch.stack([torch.randint(0, n, (B,), generator=g) for n in cfg.values_per_slot], dim=1)
    masks = O.draw_dropout_masks(cfg, x, seed=22)
    models.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    calls = {"conv_bf16": 0, "gemm_bf16": 0, "gemm_a32": 0, "tn_bf16": 0}
    real_conv, real_gemm, real_tn, real_a32 = _ops.wconv_fwd_bf16, _ops.gemm_bf16, _ops.gemm_tn_bf16, _ops.gemm_a32
    monkeypatch.setattr(_ops, "gemm_a32", lambda *a, **k: (calls.__setitem__("gemm_a32", calls["gemm_a32"] + 1), real_a32(*a, **k))[1])
    tn_args = []
    monkeypatch.setattr(_ops, "gemm_tn_bf16", lambda *a, **k: (calls.__setitem__("tn_bf16", calls["tn_bf16"] + 1), tn_args.append((a, k)), real_tn(*a, **k))[2])
    monkeypatch.setattr(_ops, "wconv_fwd_bf16", lambda *a, **k: (calls.__setitem__("conv_bf16", calls["conv_bf16"] + 1), real_conv(*a, **k))[1])
    monkeypatch.setattr(_ops, "gemm_bf16", lambda *a, **k: (calls.__setitem__("gemm_bf16", calls["gemm_bf16"] + 1), real_gemm(*a, **k))[1])
    try:
        model.train()
        loss, acc = model(x, y)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        models.set_dropout_masks(None)
    # 3 conv forwards + 2 conv data gradients; 5 input projections (bf16 planes); 5 GRU data gradients (d_gx split on
    # the fly); weight gradients on the TN kernel: dW_hh of both directions of the three layers with more than 4096 rows
    # (1000 / 500 / 250 steps x 32) + their dW_ih (the 4000- and 2016-row layers take the batched fp32 launch)
    assert all(a[-1] == 1 or k.get("nsplit") == 1 for a, k in tn_args), tn_args
    assert calls == {"conv_bf16": 5, "gemm_bf16": 5, "gemm_a32": 5, "tn_bf16": 9}, calls
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    rloss, racc, rlogits, rpred = O.slu_forward(sd, x, y, cfg, masks, explicit_gru=False)
    rloss.backward()
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
        _, _, elogits, epred = O.slu_forward({k: v.detach() for k, v in sd.items()}, x, y, cfg, None, explicit_gru=False)
    dev = (logits.cpu() - elogits).abs().max().item()
    margin = torch.stack([t.topk(2)[0] for t in elogits.split(cfg.values_per_slot, dim=1)], 0)
    margin = (margin[..., 0] - margin[..., 1]).min().item()
    print("bf16 mode, 32 x 10 s: eval logits max-abs deviation vs fp32 oracle %.3e (logit range %.2f, smallest top-2 margin %.3e); "
          "train loss %.5f vs %.5f" % (dev, elogits.abs().max().item(), margin, loss.item(), rloss.item()))
    assert (pred.cpu() != epred).sum().item() == 0 or dev >= 0.5 * margin       # a decision may only move where the margin is inside the bf16 error
    # measured on MI355X: 3.3e-4 / 3e-5; the bounds leave a factor of ~6 (bf16 rounding is data dependent)
    assert dev <= 2e-3 and abs(loss.item() - rloss.item()) <= 1e-3
    worst, worst_name, dots, na, nb = 1.0, "", 0.0, 0.0, 0.0
    for k, p in model.named_parameters():
        if sd[k].grad is None or p.grad is None:
            continue
        a, b = p.grad.cpu().double().flatten(), sd[k].grad.double().flatten()
        cos = (a @ b / (a.norm() * b.norm() + 1e-30)).item()
        if cos < worst:
            worst, worst_name = cos, k
        dots, na, nb = dots + (a @ b).item(), na + (a @ a).item(), nb + (b @ b).item()
    total = dots / (na ** 0.5 * nb ** 0.5)
    print("bf16 mode, 32 x 10 s: gradient cosine vs fp32 oracle: whole model %.5f, worst tensor %.5f (%s)" % (total, worst, worst_name))
    assert total >= 0.999 and worst >= 0.99                 # measured 0.99968 / 0.9974 (the float64 Sinc parameters)


@pytest.mark.parametrize("case", ["sinc", "conv1"])
def test_wconv_bf16_route_equals_fp32_kernel(ops, case):
    """The route bits (pool pick, sign) the bf16 convolution kernel writes for TRAINABLE blocks in bf16 mode equal the
    exact kernel's wherever the two forward results agree on the decision (pre-activations further than the bf16
    error from a tie / from zero): the backward (slu_wconv_bwd_act) reads them."""
    torch.manual_seed(5)
    if case == "sinc":
        B, l_in, c_in, c_out, k, stride, do_abs, pool = 4, 15930, 1, 80, 401, 80, True, 2
        x = (0.1 * torch.randn(B, l_in)).cuda()
        w = (torch.randn(c_out, 1, k) * 0.05).cuda()
        bias = None
    else:
        B, l_in, c_in, c_out, k, stride, do_abs, pool = 5, 301, 80, 60, 5, 1, False, 1
        x = torch.randn(B, l_in, c_in).abs().cuda()
        w = (torch.randn(c_out, c_in, k) * 0.05).cuda()
        bias = (torch.randn(c_out) * 0.1).cuda()
    ref, route_ref, l_conv = ops.wconv_fwd(x, w, bias, B, l_in, c_in, stride, do_abs, pool, 0.2, False, True)
    out, route, l_conv2 = ops.wconv_fwd_bf16(x, w, bias, B, l_in, c_in, stride, do_abs, pool, 0.2, False, 1, want_route=True)
    torch.cuda.synchronize()
    assert l_conv == l_conv2 and route.shape == route_ref.shape and route.dtype == torch.uint8
    agree = (route == route_ref).float().mean().item()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print("wconv bf16 route (%s): %.3f %% of the route bytes equal the exact kernel's; output deviation %.2e of the range" % (case, 100 * agree, err))
    assert agree >= 0.97 and err <= 2e-2
    # the exact-fp32 backward accepts the bf16 forward's (out, route)
    d = ops.wconv_bwd_act(torch.ones_like(out), out, route, B, l_conv, c_out, do_abs, pool, 0.2, False)
    assert torch.isfinite(d).all() and tuple(d.shape) == (B, l_conv, c_out)


@pytest.mark.parametrize("case", ["sinc", "sinc_odd", "conv1", "conv2", "conv2_tm"])
@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_wconv_bf16_vs_exact_fp32_kernel(ops, case, nsplit):
    """slu_wconv_fwd_bf16 (frozen CNN blocks) against the exact-fp32 windowed-conv kernel, epilogue included: the
    Sinc layer (1 -> 80 channels, 401 taps, stride 80, abs + max-pool 2 + LeakyReLU; odd length = partial pool window),
    conv1 (80 -> 60, k = 5) and conv2 (60 -> 60: input channels padded to 64 inside the kernel; time-major output)."""
    torch.manual_seed(11)
    if case.startswith("sinc"):
        B, l_in, c_in, c_out, k, stride, do_abs, pool, tm = 5, (16000 if case == "sinc" else 15930), 1, 80, 401, 80, True, 2, False
        x = (0.1 * torch.randn(B, l_in)).cuda()
        w = (torch.randn(c_out, 1, k) * 0.05).cuda()
        bias = None
    else:
        c_in, c_out = (80, 60) if case == "conv1" else (60, 60)
        B, l_in, k, stride, do_abs, pool, tm = 7, 301, 5, 1, False, 1, case.endswith("_tm")
        x = torch.randn(B, l_in, c_in).abs().cuda()
        w = (torch.randn(c_out, c_in, k) * 0.05).cuda()
        bias = (torch.randn(c_out) * 0.1).cuda()
    assert ops.wconv_bf16_supported(c_in, stride, pool)
    ref, _, _ = ops.wconv_fwd(x, w, bias, B, l_in, c_in, stride, do_abs, pool, 0.2, tm, False)
    out = ops.wconv_fwd_bf16(x, w, bias, B, l_in, c_in, stride, do_abs, pool, 0.2, tm, nsplit)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    print("wconv bf16 %s nsplit=%d: max deviation from the fp32 kernel %.2e of the output range" % (case, nsplit, err))
    assert err <= (3e-6 if nsplit >= 2 else 2e-2)         # 401 mixed-sign taps: the two fp32-class sums differ by a few ulp of the range


@pytest.mark.parametrize("nsplit", [3, 2, 1])
def test_wconv_bf16_planes_output_equals_split_of_fp32_output(ops, nsplit):
    """The last frozen CNN block hands its result to the next frozen GRU layer as bf16 planes written by the
    convolution's own epilogue: bit-equal to slu_split_bf16 of the fp32 (time-major) output, zero padding included."""
    torch.manual_seed(3)
    B, l_in, c_in, c_out, k = 9, 301, 60, 60, 5
    x = torch.randn(B, l_in, c_in).abs().cuda()
    w = (torch.randn(c_out, c_in, k) * 0.05).cuda()
    bias = (torch.randn(c_out) * 0.1).cuda()
    assert ops.wconv_bf16_planes_ok(c_out, 1) and not ops.wconv_bf16_planes_ok(16, 1) and not ops.wconv_bf16_planes_ok(c_out, 2)
    out = ops.wconv_fwd_bf16(x, w, bias, B, l_in, c_in, 1, False, 1, 0.2, True, nsplit)          # (l_out, B, c_out) fp32
    act = ops.wconv_fwd_bf16(x, w, bias, B, l_in, c_in, 1, False, 1, 0.2, True, nsplit, out_planes=True)
    torch.cuda.synchronize()
    T = out.shape[0]
    assert (act.T, act.B, act.C) == (T, B, c_out) and tuple(act.planes.shape) == (nsplit, T * B, 64)
    want = ops.split_bf16(out.view(T * B, c_out), nsplit)
    assert torch.equal(act.planes.view(torch.int16), want.view(torch.int16))
    assert act.planes[:, :, c_out:].float().abs().max().item() == 0.0


def test_wconv_bf16_row_table_input_equals_concatenated_input(ops):
    """A look-ahead super-batch read through a row-pointer table (the batches where they lie, addresses refreshed by
    slu_store_u64) == the same kernel on the concatenated copy, bit for bit."""
    torch.manual_seed(4)
    P, B, T = 5, 3, 16000
    xs = [(0.1 * torch.randn(B, T)).cuda() for _ in range(P)]
    w = (torch.randn(80, 1, 401) * 0.05).cuda()
    ref = ops.wconv_fwd_bf16(torch.cat(xs).unsqueeze(2).contiguous(), w, None, P * B, T, 1, 80, True, 2, 0.2, False, 3)
    words = torch.zeros(8, dtype=torch.int64, device="cuda")
    ops.store_u64(words, [x.data_ptr() for x in xs] + [77])
    assert words.tolist()[:P + 1] == [x.data_ptr() for x in xs] + [77]
    out = ops.wconv_fwd_bf16(ops.RowTable(words[:P], B, T), w, None, P * B, T, 1, 80, True, 2, 0.2, False, 3)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
