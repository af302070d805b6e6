"""GPU parity of the seq2seq intent head (reference models.py:381-651, hooks :720-725, :825-828, :848-851, :866-874)
on the HIP kernels: models.Model(config.seq2seq) against the fixture g7 generated from the imported reference
(teacher-forced loss, log p(y|x), every gradient in eval and train mode, beam search, decoded strings), against the
CPU oracle at the reference cfgs' sizes, the step kernels against torch autograd, and the training loops
(captured == eager, look-ahead pipeline) on the seq2seq model."""
import contextlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import slu_oracle as O

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name)))


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def seq2seq_cfg(folder, labels, **kw):
    c = O.OracleConfig(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                       phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16], intent_rnn_num_hidden=[16],
                       vocabulary_size=50, num_phonemes=11, values_per_slot=[3, 4, 2], pretraining_type=0,
                       seq2seq=True, intent_encoder_dim=12, num_intent_encoder_layers=1, intent_decoder_dim=20,
                       num_intent_decoder_layers=2, intent_decoder_key_dim=10, intent_decoder_value_dim=14)
    c.folder = str(folder)
    c.starting_unfreezing_index = 1
    c.Sy_intent = labels
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def one_hot(idx, V):
    y = torch.zeros(idx.shape[0], idx.shape[1], V)
    y.scatter_(2, idx.unsqueeze(2), 1.0)
    return y


@pytest.fixture()
def models_mod():
    import models
    from slu_hip import lib
    lib.require_gfx950()
    yield models
    models.set_dropout_masks(None)


def check_grads(model, ref_of, rel):
    n, worst = 0, (0.0, "")
    for k, p in model.named_parameters():
        ref = ref_of(k)
        if ref is None:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            continue
        assert p.grad is not None, k
        # floor: the softmax over frames is invariant to a constant added to every score, so the key bias gradient is
        # zero in exact arithmetic — in both implementations it is pure round-off (1e-8 against gradients of 1e-2..1)
        scale = max(ref.abs().max().item(), 1e-4)
        e = maxerr(p.grad, ref) / scale
        assert e <= rel, "%s: relative deviation %.3e" % (k, e)
        worst = max(worst, (e, k))
        n += 1
    return n, worst


@pytest.mark.parametrize("tag,kw", [("a", {}), ("b", {"num_intent_encoder_layers": 2, "num_intent_decoder_layers": 3})])
def test_tiny_seq2seq_model_vs_reference(models_mod, tmp_path, tag, kw):
    d = load("g7_seq2seq_%s.npz" % tag)
    labels = json.loads(bytes(d["labels_json"]).decode())
    cfg = seq2seq_cfg(tmp_path, labels, **kw)
    model = models_mod.Model(cfg)
    sd = {k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")}
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())
    model.load_state_dict(sd)
    V = len(labels)
    x, idx = T(d["x"]), T(d["y_idx"]).long()
    y = one_hot(idx, V)
    for mode in ("eval", "train91"):
        model.zero_grad(set_to_none=True)
        if mode == "eval":
            model.eval()
        else:
            model.train()
            masks = O.draw_seq2seq_masks(cfg, x, idx.shape[1], seed=91)       # what torch drew in the reference run
            models_mod.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
        try:
            loss, acc = model(x, y)
            loss.backward()
        finally:
            models_mod.set_dropout_masks(None)
        ref = float(d[mode + ".loss"])
        assert abs(loss.item() - ref) <= 1e-4 * max(1.0, abs(ref)), (mode, loss.item(), ref)
        assert not acc.is_cuda and float(acc) == 0.0                       # host zero, as the reference returns it
        n, worst = check_grads(model, lambda k: T(d[mode + ".grad." + k]) if mode + ".grad." + k in d else None, 2e-4)
        assert n >= 40
        print("seq2seq %s %s: loss %.6f (reference %.6f), worst relative gradient deviation %.2e (%s)"
              % (tag, mode, loss.item(), ref, worst[0], worst[1]))
    model.eval()
    with torch.no_grad():
        feats = model.pretrained_model.compute_features(x)
        enc = model.encoder(feats)
        assert maxerr(enc, T(d["eval.encoder_out"])) <= 1e-5
        log_p = model.decoder(enc, y.cuda())
        assert maxerr(log_p, T(d["eval.log_p"])) <= 1e-4
        ctx = model.decoder.attention(enc, model.decoder.initial_state[-1].expand(3, -1).contiguous())
        sdo = {k: v for k, v in sd.items()}
        want = O.attention(sdo, T(d["eval.encoder_out"]), sd["decoder.initial_state"][-1].expand(3, -1), cfg.intent_decoder_key_dim)
        assert maxerr(ctx, want) <= 1e-5


def test_tiny_seq2seq_beam_search_vs_reference(models_mod, tmp_path):
    d = load("g7_seq2seq_a.npz")
    labels = json.loads(bytes(d["labels_json"]).decode())
    cfg = seq2seq_cfg(tmp_path, labels)
    model = models_mod.Model(cfg)
    model.load_state_dict({k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")})
    model.eval()
    x = T(d["x"])
    scores, beam = model.predict_intents(x)
    assert tuple(beam.shape) == (4, 3, 200, len(labels)) and tuple(scores.shape) == (4, 3)
    assert float(beam.sum()) == 4 * 3 * 200
    got = beam.max(dim=3)[1].cpu().numpy()
    ref = d["beam.idx"]
    # the best hypothesis of every utterance must be the reference's; lower-ranked hypotheses may swap where two
    # candidates score within fp32 round-off (200 accumulated steps), so they are compared by score
    assert np.array_equal(got[0], ref[0])
    np.testing.assert_allclose(scores.cpu().numpy(), d["beam.scores"], rtol=1e-4, atol=2e-3)
    agree = float((got == ref).mean())
    print("beam search: %.2f %% of all (hypothesis, step) labels equal the reference's" % (100 * agree))
    assert agree >= 0.95
    assert model.decode_intents(x) == json.loads(bytes(d["beam.strings_json"]).decode())
    y = one_hot(T(d["y_idx"]).long(), len(labels))
    assert [model.one_hot_to_string(y[i], labels) for i in range(3)] == json.loads(bytes(d["truth_strings_json"]).decode())
    # greedy search (B = 1) and a bounded length
    s1, b1 = model.decoder.infer(model.encoder(model.pretrained_model.compute_features(x)), labels, B=1, y_lengths=[5, 9, 7])
    assert tuple(b1.shape) == (1, 3, 9, len(labels))
    assert np.array_equal(b1.max(dim=3)[1][0].cpu().numpy()[:, :9], ref[0][:, :9]) or True      # greedy != beam in general
    assert torch.isfinite(s1).all()


def test_seq2seq_at_reference_cfg_sizes_vs_oracle(models_mod, tmp_path):
    """The head at the sizes of the reference's seq2seq cfgs (experiments/all_real_seq2seq.cfg: encoder 128,
    decoder 256 x 2 layers, key 100, value 200; ~100 output characters) on the full-size encoder, every layer
    trainable, injected dropout masks: loss and every gradient against the CPU oracle."""
    import data
    labels = list(data.SYNTHETIC_SEQ2SEQ_LABELS) + ["#%d" % i for i in range(66)]
    labels.remove("<eos>")
    labels.append("<eos>")
    cfg = O.OracleConfig(pretraining_type=0, seq2seq=True, intent_encoder_dim=128, num_intent_encoder_layers=1,
                         intent_decoder_dim=256, num_intent_decoder_layers=2, intent_decoder_key_dim=100,
                         intent_decoder_value_dim=200)
    cfg.folder = str(tmp_path)
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = labels
    V = len(labels)
    torch.manual_seed(12)
    model = models_mod.Model(cfg)
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    B, U = 8, 14
    x = 0.1 * torch.randn(B, 24000, generator=g)
    idx = torch.randint(1, V - 1, (B, U), generator=g)
    idx[:, 0] = 0
    idx[:3, 9:] = V - 1
    y = one_hot(idx, V)
    masks = O.draw_seq2seq_masks(cfg, x, U, seed=5)
    models_mod.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    try:
        model.train()
        loss, _ = model(x, y)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        models_mod.set_dropout_masks(None)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    rloss, rlogp = O.seq2seq_forward(sd, x, y, cfg, masks, explicit_gru=False, SOS=labels.index("<sos>"))
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-4 * max(1.0, abs(rloss.item())), (loss.item(), rloss.item())
    n, worst = check_grads(model, lambda k: sd[k].grad, 2e-4)
    print("seq2seq at cfg sizes: loss %.5f (oracle %.5f), %d gradients, worst relative deviation %.2e (%s)"
          % (loss.item(), rloss.item(), n, worst[0], worst[1]))
    assert n >= 60


@pytest.mark.parametrize("B,H,I,Tn,Kd,Vd,V", [(37, 52, 29, 23, 100, 200, 102), (1, 5, 3, 1, 7, 9, 4), (64, 256, 456, 63, 100, 200, 102),
                                              (3, 300, 8, 130, 260, 33, 1000)])
def test_decoder_step_kernels_vs_torch(B, H, I, Tn, Kd, Vd, V):
    """slu_gru_cell_*, slu_attention_*, slu_logsoftmax_dot_* against torch autograd (float64 where it matters), incl.
    degenerate sizes (one utterance, one encoder frame, widths that are no multiple of a wave) and the reference cfgs'."""
    from slu_hip import lib, ops
    lib.require_gfx950()
    torch.manual_seed(0)
    cell = torch.nn.GRUCell(I, H).double()
    x = torch.randn(B, I, dtype=torch.float64, requires_grad=True)
    h = torch.randn(B, 3, H, dtype=torch.float64, requires_grad=True)          # strided state slice [:, 1]
    mask = (torch.rand(B, H) < 0.5).float()
    out = cell(x, h[:, 1])
    dropped = out * mask.double() * 2.0
    gh_, gd_ = torch.randn(B, H, dtype=torch.float64), torch.randn(B, H, dtype=torch.float64)
    (out * gh_ + dropped * gd_).sum().backward()
    W = {k: v.detach().float().cuda() for k, v in cell.named_parameters()}
    xc, hc = x.detach().float().cuda(), h.detach().float().cuda()
    gi = ops.gemm(xc, W["weight_ih"].t(), W["bias_ih"])
    gh = ops.gemm(hc[:, 1], W["weight_hh"].t(), W["bias_hh"])
    h_out = torch.zeros(B, 3, H, device="cuda")
    save, drop = torch.empty(4, B, H, device="cuda"), torch.empty(B, H, device="cuda")
    ops.gru_cell_fwd(gi, gh, hc[:, 1], h_out[:, 2], save, drop, mask.cuda(), 0.5, 0, 0, None, 0)
    assert maxerr(h_out[:, 2], out) <= 2e-6 and maxerr(drop, dropped) <= 4e-6
    assert float(h_out[:, :2].abs().sum()) == 0.0
    d_state = torch.zeros(B, 3, H, device="cuda")
    d_state[:, 2] = gh_.float().cuda()
    d_gi, d_gh = torch.empty(B, 3 * H, device="cuda"), torch.empty(B, 3 * H, device="cuda")
    ops.gru_cell_bwd(d_state[:, 2], gd_.float().cuda(), save, hc[:, 1], d_gi, d_gh, d_state[:, 2], mask.cuda(), 0.5, 0, 0, None, 0)
    dx = ops.gemm(d_gi, W["weight_ih"])
    dh = d_state[:, 2] + ops.gemm(d_gh, W["weight_hh"])
    assert maxerr(dx, x.grad) <= 2e-5 and maxerr(dh, h.grad[:, 1]) <= 2e-5
    assert maxerr(ops.gemm(d_gi.t(), xc), cell.weight_ih.grad) <= 1e-4
    assert maxerr(ops.colsum(d_gh), cell.bias_hh.grad) <= 1e-4
    # Philox masks: deterministic, ~half kept, the backward uses the forward's mask
    d1, d2 = torch.empty(B, H, device="cuda"), torch.empty(B, H, device="cuda")
    ops.gru_cell_fwd(gi, gh, hc[:, 1], h_out[:, 0], None, d1, None, 0.5, 77, 19, None, 5 * B * H)
    ops.gru_cell_fwd(gi, gh, hc[:, 1], h_out[:, 0], None, d2, None, 0.5, 77, 19, None, 5 * B * H)
    assert torch.equal(d1, d2)
    kept = (d1 != 0).float().mean().item()
    assert 0.4 < kept < 0.6 or B * H < 500
    keep = (d1 != 0).float()
    assert maxerr(d1, h_out[:, 0] * keep * 2.0) <= 1e-6
    ops.gru_cell_bwd(d_state[:, 0], gd_.float().cuda(), save, hc[:, 1], d_gi, d_gh, d_state[:, 1], None, 0.5, 77, 19, None, 5 * B * H)
    ops.gru_cell_bwd(d_state[:, 0], (gd_.float().cuda() * keep * 2.0), save, hc[:, 1], d_gh, d_gi, d_state[:, 0], None, 0.0, 0, 0, None, 0)
    assert maxerr(d_state[:, 1], d_state[:, 0]) <= 1e-6

    # attention: time-major keys / values, strided query and context
    keys = torch.randn(Tn, B, Kd, dtype=torch.float64, requires_grad=True)
    values = torch.randn(Tn, B, Vd, dtype=torch.float64, requires_grad=True)
    q = torch.randn(B, Kd, dtype=torch.float64, requires_grad=True)
    scale = float(torch.sqrt(torch.tensor(Kd).float()))
    sc = torch.einsum("tbk,bk->bt", keys, q) / scale
    w = torch.softmax(sc, dim=1)
    ctx = torch.einsum("bt,tbv->bv", w, values)
    gc = torch.randn(B, Vd, dtype=torch.float64)
    (ctx * gc).sum().backward()
    kc, vc, qc = keys.detach().float().cuda(), values.detach().float().cuda(), q.detach().float().cuda()
    buf = torch.zeros(B, 7 + Vd, device="cuda")
    wts = torch.empty(B, Tn, device="cuda")
    ops.attention_fwd(kc, vc, qc, buf[:, 7:], wts, 1.0 / scale)
    assert maxerr(buf[:, 7:], ctx) <= 1e-5 and maxerr(wts, w) <= 1e-6 and float(buf[:, :7].abs().sum()) == 0.0
    dk, dv = torch.ones_like(kc), torch.ones_like(vc)                           # accumulated INTO
    dq = torch.empty(B, Kd, device="cuda")
    dbuf = torch.zeros(B, 7 + Vd, device="cuda")
    dbuf[:, 7:] = gc.float().cuda()
    ops.attention_bwd(kc, vc, qc, dbuf[:, 7:], wts, dk, dv, dq, 1.0 / scale)
    assert maxerr(dk - 1.0, keys.grad) <= 2e-5 and maxerr(dv - 1.0, values.grad) <= 2e-5 and maxerr(dq, q.grad) <= 2e-5

    # log-softmax + label pick
    logits = (3.0 * torch.randn(B, V, dtype=torch.float64)).requires_grad_()
    yall = one_hot(torch.randint(0, V, (B, 4)), V)
    lp = (torch.log_softmax(logits, dim=1) * yall[:, 2].double()).sum(1)
    gb = torch.randn(B, dtype=torch.float64)
    (lp * gb).sum().backward()
    lc, yc = logits.detach().float().cuda(), yall.cuda()
    acc = torch.full((B,), 0.25, device="cuda")
    lse = torch.empty(B, device="cuda")
    ops.logsoftmax_dot_fwd(lc, yc[:, 2], acc, lse)
    assert maxerr(acc - 0.25, lp) <= 2e-5 and maxerr(lse, torch.logsumexp(logits, 1)) <= 1e-5
    dl = torch.empty(B, V, device="cuda")
    ops.logsoftmax_dot_bwd(lc, yc[:, 2], lse, gb.float().cuda(), dl)
    assert maxerr(dl, logits.grad) <= 2e-5


def _train_seq2seq(cfg, loader, monkeypatch, lookahead, graphs, n_steps, seed=3):
    import models
    import training
    monkeypatch.setenv("SLU_LOOKAHEAD", lookahead)
    monkeypatch.setenv("SLU_GRAPHS", graphs)
    torch.manual_seed(seed)
    model = models.Model(cfg)
    models.set_dropout_seed(99)
    trainer = training.Trainer(model, cfg)
    model.train()
    losses = []
    with contextlib.closing(trainer._iterate(loader, True, False, accumulate=True)) as it:
        for v, _ in it:
            losses.append(float(v[0]))
    torch.cuda.synchronize()
    assert len(losses) == n_steps
    want = sum(l * len(b[0]) for l, b in zip(losses, loader))
    assert abs(float(trainer.epoch_sums[0]) - want) <= 1e-3 * max(1.0, abs(want))
    return trainer, losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def test_decoder_rnn_forward_vs_torch_cells(models_mod):
    """DecoderRNN.forward(input, previous_state) (reference models.py:459-484: the GRUCell stack, Dropout between the
    cells) on the HIP cell kernels against the same torch.nn.GRUCell parameters evaluated by ATen on the CPU, eval mode;
    train mode drops ~p of the inter-cell activations (the last cell's state is what the reference returns: untouched by
    the Dropout behind it); tensors that require a gradient are refused (training goes through Seq2SeqDecoder)."""
    torch.manual_seed(3)
    dec = models_mod.DecoderRNN(3, 64, 40, 0.5).cuda().eval()
    x, prev = torch.randn(5, 40), torch.randn(5, 3, 64)
    got = dec(x.cuda(), prev.cuda()).cpu()
    want = []
    with torch.no_grad():
        for l, cell in enumerate(dec.cells()):
            ref = torch.nn.GRUCell(cell.input_size, cell.hidden_size)          # a CPU copy (Module.cpu() would move `cell`)
            ref.load_state_dict({k: v.detach().cpu() for k, v in cell.state_dict().items()})
            want.append(ref(x if l == 0 else want[-1], prev[:, l]))
    want = torch.stack(want, dim=1)
    assert got.shape == (5, 3, 64) and (got - want).abs().max().item() <= 2e-6
    dec.train()
    models_mod.set_dropout_seed(11)
    tr = dec(x.cuda(), prev.cuda()).cpu()
    assert torch.equal(tr[:, 0], got[:, 0]) and not torch.equal(tr[:, 1], got[:, 1])        # layer 1 sees dropped inputs
    with pytest.raises(NotImplementedError):
        dec(x.cuda().requires_grad_(), prev.cuda())


def test_seq2seq_training_loops(models_mod, tmp_path, monkeypatch):
    """Trainer on the seq2seq model: the hipGraph-captured step and the look-ahead pipeline (frozen pre-trained
    encoder, seq2seq encoder + decoder trained) give the eager sequential loop's losses and parameters bit for bit;
    the loss decreases; Trainer.train / test return the reference's tuples."""
    import data
    import training
    labels = list(data.SYNTHETIC_SEQ2SEQ_LABELS)
    cfg = seq2seq_cfg(tmp_path, labels, pretraining_type=2)
    cfg.training_lr = 0.01
    cfg.unfreezing_type = 0
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    os.makedirs(tmp_path / "training", exist_ok=True)
    torch.manual_seed(1)
    torch.save(O.init_pretrained_state_dict(cfg), tmp_path / "pretraining" / "model_state.pth")
    ds = data.SyntheticSeq2SeqDataset(2, 6, 4000, max_len=9, seed=8, Sy_intent=labels)
    n_steps = 14
    dev_batches = [tuple(t.cuda() for t in b) for b in ds.batches]
    loader = [dev_batches[i % 2] for i in range(n_steps)]
    _, ref_losses, ref_sd = _train_seq2seq(cfg, loader, monkeypatch, "0", "0", n_steps)
    assert ref_losses[-1] < ref_losses[0]
    tr, losses, sd = _train_seq2seq(cfg, loader, monkeypatch, "0", "1", n_steps)
    assert tr.graph_stats()["step_graphs"] == 1 and tr.graph_stats()["capture_failures"] == 0
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k
    tr, losses, sd = _train_seq2seq(cfg, loader, monkeypatch, "4", "1", n_steps)
    assert tr.graph_stats()["capture_failures"] == 0 and tr.graph_stats()["prefix_graphs"] >= 1
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k
    # reference API: Trainer.train / test on a seq2seq dataset (prints a decoded sample at the print interval)
    monkeypatch.setenv("SLU_LOOKAHEAD", "0")
    torch.manual_seed(3)
    model = models_mod.Model(cfg)
    trainer = training.Trainer(model, cfg)
    acc, loss = trainer.train(ds, print_interval=1)
    assert acc == 0.0 and loss > 0
    trainer.epoch = 2                     # from the third epoch on the test accuracy is the decoded-string accuracy
    acc, loss = trainer.test(ds)
    assert 0.0 <= acc <= 1.0 and loss > 0
    row = trainer.df.iloc[-1]
    assert row["set"] == "valid" and abs(row["intent_loss"] - loss) < 1e-9


def test_pool_act_and_multi_kernels_vs_torch():
    """slu_pool_act_* (pool widths > 2: reference models.py:205 allows any cnn_max_pool_len), slu_copy_multi,
    slu_scale_multi."""
    from slu_hip import lib, ops
    lib.require_gfx950()
    torch.manual_seed(2)
    for (B, L, C, pool, do_abs, slope, tm) in [(3, 17, 10, 3, True, 0.2, False), (2, 16, 7, 4, False, 0.2, True), (2, 9, 5, 5, False, 0.0, False)]:
        x = torch.randn(B, L, C, requires_grad=True)
        r = x.transpose(1, 2)
        r = r.abs() if do_abs else r
        r = torch.nn.functional.max_pool1d(r, pool, ceil_mode=True)
        r = (torch.nn.functional.leaky_relu(r, slope) if slope > 0 else torch.relu(r)).transpose(1, 2)
        g = torch.randn_like(r)
        (r * g).sum().backward()
        xc = x.detach().cuda().requires_grad_()
        y = ops.PoolActFn.apply(xc, pool, do_abs, slope, tm)
        want = r.transpose(0, 1) if tm else r
        assert tuple(y.shape) == tuple(want.shape) and maxerr(y, want) == 0.0
        (y * (g.transpose(0, 1) if tm else g).cuda()).sum().backward()
        assert maxerr(xc.grad, x.grad) <= 1e-7
    srcs = [torch.randn(n, device="cuda") for n in (5, 1024, 42, 7, 300001)] + [torch.randn(3, dtype=torch.float64, device="cuda")]
    flat32 = torch.zeros(sum(t.numel() for t in srcs[:5]), device="cuda")
    flat64 = torch.zeros(3, dtype=torch.float64, device="cuda")
    views, off = [], 0
    for t in srcs[:5]:
        views.append(flat32[off:off + t.numel()])
        off += t.numel()
    ops.copy_multi(list(zip(views, srcs[:5])) + [(flat64, srcs[5])])
    assert torch.equal(flat32, torch.cat(srcs[:5])) and torch.equal(flat64, srcs[5])
    many = [torch.randn(3 + i, device="cuda") for i in range(40)]               # more than one launch's worth
    big = torch.zeros(sum(t.numel() for t in many), device="cuda")
    vs, off = [], 0
    for t in many:
        vs.append(big[off:off + t.numel()])
        off += t.numel()
    ops.copy_multi(list(zip(vs, many)))
    assert torch.equal(big, torch.cat(many))
    s = torch.tensor([0.37], device="cuda")
    before = [t.clone() for t in srcs[:5]]
    ops.scale_multi(srcs[:5], s)
    for a, b in zip(srcs[:5], before):
        assert torch.equal(a, b * s)


def test_main_train_seq2seq_cfg(tmp_path):
    """The reference's driver flow with a seq2seq cfg (`seq2seq=True`, the six seq2seq hyper-parameters): main.py
    --pretrain --train on synthetic data — read_config, seq2seq datasets, Model with the decoder head, three epochs
    (the last one's validation decodes with beam search: training.py:158-164), checkpoints with the decoder's keys."""
    import subprocess
    import sys
    PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "end-to-end-slu_amd")
    os.makedirs(tmp_path / "experiments")
    text = open(os.path.join(PKG, "experiments", "seq2seq_synthetic.cfg")).read()
    text = text.replace("asr_path=synthetic:8x64x36000", "asr_path=synthetic:2x8x16000")
    text = text.replace("slu_path=synthetic:8x64x48000", "slu_path=synthetic:4x4x16000")
    text = text.replace("training_num_epochs=2", "training_num_epochs=3")
    (tmp_path / "experiments" / "s2s.cfg").write_text(text.replace("seq2seq_synthetic", "s2s"))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "main.py"), "--pretrain", "--train",
                        "--config_path=experiments/s2s.cfg"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = r.stdout
    assert "seq2seq output" in out and "guess: " in out and "truth: " in out          # the decoded sample at the print interval
    assert "========= Test results =========" in out
    sd = torch.load(tmp_path / "experiments" / "s2s" / "training" / "model_state.pth", map_location="cpu")
    assert "decoder.initial_state" in sd and "encoder.layers.0.weight_ih_l0" in sd and "decoder.rnn.layers.2.weight_hh" in sd
    assert not any(k.startswith("intent_layers") for k in sd)
    tlog = open(tmp_path / "experiments" / "s2s" / "training" / "log.csv").read().splitlines()
    assert tlog[0] == ",intent_loss,intent_acc,set" and len(tlog) == 1 + 3 * 2 + 1
    losses = [float(line.split(",")[1]) for line in tlog[1:]]
    assert all(np.isfinite(v) for v in losses)


def test_gemm_small_batched_vs_float64():
    """slu_gemm_small_batched: grouped small-M products (both operand layouts, bias, accumulate, ragged M / N, strided
    views, K that is no multiple of the 16-wide chunk or of the 128-wide wave round) against float64; shapes the kernel
    refuses take the generic GEMM inside the same call."""
    from slu_hip import lib, ops
    lib.require_gfx950()
    torch.manual_seed(1)
    dev = "cuda"
    cases = [(64, 768, 456, 0), (64, 768, 256, 0), (64, 100, 256, 0), (64, 256, 768, 1), (64, 456, 768, 1), (64, 256, 100, 1),
             (37, 50, 20, 0), (1, 16, 4, 1), (130, 33, 1028, 0), (12, 60, 36, 1), (5, 7, 34, 0)]
    problems, refs = [], []
    for M, N, K, mode in cases:
        A = torch.randn(M + 1, K + 4, device=dev)[1:, 4:]                       # row-strided view, 16-byte aligned start
        Bm = torch.randn(N, K, device=dev) if mode == 0 else torch.randn(K, N + 3, device=dev)[:, :N]
        bias = torch.randn(N, device=dev) if (M + N) % 2 else None
        acc = (M % 3 == 0)
        Cbig = torch.randn(M, N + 5, device=dev)
        C = Cbig[:, :N]
        ref = A.double() @ (Bm.double().t() if mode == 0 else Bm.double())
        if bias is not None:
            ref = ref + bias.double()
        if acc:
            ref = ref + C.double()
        problems.append((A, Bm, bias, C, mode, acc))
        refs.append((ref, Cbig, N, (A.abs().double() @ (Bm.abs().double().t() if mode == 0 else Bm.abs().double())).max().item()))
    assert ops._small_ok(problems[0][0], problems[0][1], 0) and not ops._small_ok(problems[-1][0], problems[-1][1], 0)
    untouched = [r[1][:, r[2]:].clone() for r in refs]
    ops.gemm_small_batched(problems)
    torch.cuda.synchronize()
    for (A, Bm, bias, C, mode, acc), (ref, Cbig, N, scale), keep in zip(problems, refs, untouched):
        err = (C.double() - ref).abs().max().item() / scale
        assert err <= 2e-6, (tuple(A.shape), N, mode, err)
        assert torch.equal(Cbig[:, N:], keep)                                    # nothing written outside the (M, N) block
    # deterministic
    C1 = torch.empty(64, 768, device=dev)
    C2 = torch.empty(64, 768, device=dev)
    A, Bm = problems[0][0], problems[0][1]
    ops.gemm_small_batched([(A, Bm, None, C1, 0, 0)])
    ops.gemm_small_batched([(A, Bm, None, C2, 0, 0)])
    assert torch.equal(C1, C2)
