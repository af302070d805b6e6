"""GPU parity tests at model level: models.Model / models.PretrainedModel / training.Trainer running
on the HIP kernels against (a) the golden fixtures generated from the imported reference and
(b) the CPU oracle on seeded inputs at BASELINE.json's full sizes.

North-star tolerance: logits max-abs deviation <= 1e-4 (fp32), predicted intents identical;
gradients within 1e-4 of the per-tensor max-abs (2e-4 for digests of full-size tensors).
"""
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


def _sy(vps):
    names = ["action", "object", "location"]
    return {names[s]: {"%s%d" % (names[s][0], v): v for v in range(n)} for s, n in enumerate(vps)}


def tiny_cfg(folder, **kw):
    c = O.OracleConfig(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                       phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16],
                       intent_rnn_num_hidden=[16], vocabulary_size=50, num_phonemes=11,
                       values_per_slot=[3, 4, 2], pretraining_type=0)
    c.folder = str(folder)
    c.starting_unfreezing_index = 1
    for k, v in kw.items():
        setattr(c, k, v)
    c.Sy_intent = _sy(c.values_per_slot)
    return c


def full_cfg(folder, **kw):
    c = O.OracleConfig(**kw)
    c.folder = str(folder)
    c.starting_unfreezing_index = 1
    c.training_lr = 0.001                    # experiments/no_unfreezing.cfg
    c.Sy_intent = _sy(c.values_per_slot)
    return c


def maxerr(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def check_grads(model, d, tag, rel=1e-4):
    n = 0
    for k, p in model.named_parameters():
        key = tag + "grad." + k
        if key in d:
            ref = T(d[key])
            assert p.grad is not None, k
            scale = max(ref.abs().max().item(), 1e-6)
            e = maxerr(p.grad, ref)
            assert e <= rel * scale, "%s %s: err %.3e scale %.3e" % (tag, k, e, scale)
            assert p.grad.dtype == ref.dtype, k
            n += 1
        else:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
    return n


def masks_to_cuda(masks):
    return {k: v.cuda() for k, v in masks.items()}


@pytest.fixture()
def models_mod():
    import models
    from slu_hip import lib
    lib.require_gfx950()
    yield models
    models.set_dropout_masks(None)


def test_tiny_model_eval_and_train_vs_reference(models_mod, tmp_path):
    d = load("g5_tiny_model.npz")
    cfg = tiny_cfg(tmp_path)
    model = models_mod.Model(cfg)
    model.load_state_dict({k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")})
    x, y = T(d["x"]), T(d["y"])
    model.eval()
    logits, pred = model.predict_intents(x)
    assert maxerr(logits, T(d["eval.logits"])) <= 1e-5
    assert np.array_equal(pred.cpu().numpy(), d["eval.pred"])
    # decode_intents (reference models.py:853-865: the nested scan of Sy_intent for the value whose index was predicted),
    # non-seq2seq: the strings of the REFERENCE's predicted intents (fixture g5), utterance by utterance
    want = [[value for idx, slot in enumerate(model.Sy_intent) for value in model.Sy_intent[slot]
             if int(row[idx]) == model.Sy_intent[slot][value]] for row in d["eval.pred"]]
    assert model.decode_intents(x) == want and all(len(w) == len(cfg.values_per_slot) for w in want)
    feats = model.pretrained_model.compute_features(x)
    assert tuple(feats.shape) == d["eval.features"].shape
    assert maxerr(feats, T(d["eval.features"])) <= 1e-5
    loss, acc = model(x, y)
    loss.backward()
    assert abs(loss.item() - float(d["eval.loss"])) <= 1e-5 and acc.item() == float(d["eval.acc"])
    assert check_grads(model, d, "eval.") >= 40
    # train mode with the dropout masks torch drew under seed 77 in the reference run
    model.zero_grad()
    model.train()
    models_mod.set_dropout_masks(masks_to_cuda(O.draw_dropout_masks(cfg, x, seed=77)))
    loss, acc = model(x, y)
    loss.backward()
    assert abs(loss.item() - float(d["train77.loss"])) <= 1e-5 and acc.item() == float(d["train77.acc"])
    assert check_grads(model, d, "train77.") >= 40


def test_tiny_asr_heads_vs_reference(models_mod, tmp_path):
    d = load("g5_tiny_asr.npz")
    for ptype in (2, 1):
        cfg = tiny_cfg(tmp_path, pretraining_type=ptype)
        pm = models_mod.PretrainedModel(cfg)
        pm.load_state_dict({k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")})
        pm.eval()
        x = T(d["x"])
        ph, wd = pm.compute_posteriors(x)
        assert maxerr(ph, T(d["posteriors.phoneme"])) <= 1e-5 and maxerr(wd, T(d["posteriors.word"])) <= 1e-5
        pl, wl, pa, wa = pm(x, T(d["y_phoneme"]), T(d["y_word"]))
        tag = "pt%d." % ptype
        assert abs(pl.item() - float(d[tag + "phoneme_loss"])) <= 1e-5
        assert abs(float(wl.sum()) - float(d[tag + "word_loss"].sum())) <= 1e-5
        assert pa.item() == float(d[tag + "phoneme_acc"]) and float(wa.sum()) == float(d[tag + "word_acc"].sum())
        (pl + wl.sum().to(pl.device) if ptype == 2 else pl).backward()
        assert check_grads(pm, d, tag) >= 20


def _full_model_from_seeds(models_mod, tmp_path, meta):
    cfg = full_cfg(tmp_path, pretraining_type=2)
    os.makedirs(os.path.join(cfg.folder, "pretraining"), exist_ok=True)
    os.makedirs(os.path.join(cfg.folder, "training"), exist_ok=True)
    torch.manual_seed(meta["pretrain_seed"])
    pre = O.init_pretrained_state_dict(cfg)
    torch.save(pre, os.path.join(cfg.folder, "pretraining", "model_state.pth"))
    torch.manual_seed(meta["model_seed"])
    model = models_mod.Model(cfg)
    return cfg, model


def _digest(t):
    f = t.detach().cpu().double().flatten()
    return np.concatenate([[f.norm().item(), f.sum().item()], f[:8].numpy()])


def _check_digests(model, d, prefix, trainable_only):
    n = 0
    for k, p in model.named_parameters():
        key = prefix + k
        if key in d:
            # gradients within 1e-4 of the tensor's scale.  The fixture holds the REFERENCE's own numbers per
            # tensor: L2 norm, sum, first eight values — the norm to 1e-4 relative, the eight values to 1e-4 of the
            # largest of them (a stricter scale than the tensor's max-abs, which the digest does not carry)
            ref = d[key]
            got = _digest(p.grad)
            assert abs(got[0] - ref[0]) <= 1e-4 * max(ref[0], 1e-6), (k, got[0], ref[0])
            np.testing.assert_allclose(got[2:], ref[2:], atol=1e-4 * max(np.abs(ref[2:]).max(), 1e-6), err_msg=k)
            n += 1
        elif trainable_only:
            assert p.grad is None, k
    return n


def test_baseline_config0_full_size_vs_reference(models_mod, tmp_path):
    """BASELINE.json configs[0]: no_unfreezing.cfg architecture, 16 synthetic 1 s waveforms, one
    forward+backward (+Adam) step; reference values from fixture g6."""
    import hashlib
    import training
    d = load("g6_full_model.npz")
    meta = json.loads(bytes(d["meta_json"]).decode())
    cfg, model = _full_model_from_seeds(models_mod, tmp_path, meta)
    sd = model.state_dict()
    assert {k: hashlib.sha256(v.cpu().contiguous().numpy().tobytes()).hexdigest() for k, v in sd.items()} == meta["model_sha256"]
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(16, 16000, generator=g)
    y = torch.stack([torch.randint(0, n, (16,), generator=g) for n in cfg.values_per_slot], dim=1)
    assert np.array_equal(y.numpy(), d["y"])
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
        loss, acc = model(x, y)
    err = maxerr(logits, T(d["eval.logits"]))
    print("configs[0] eval logits max-abs deviation: %.3e" % err)
    assert err <= 1e-4
    assert np.array_equal(pred.cpu().numpy(), d["eval.pred"])                  # intent-accuracy bit-identical
    assert abs(loss.item() - float(d["eval.loss"])) <= 1e-4 and acc.item() == float(d["eval.acc"])
    # train mode, frozen encoder, dropout seed 999
    model.train()
    model.zero_grad()
    models_mod.set_dropout_masks(masks_to_cuda(O.draw_dropout_masks(cfg, x, seed=999)))
    loss, acc = model(x, y)
    loss.backward()
    assert abs(loss.item() - float(d["train999.loss"])) <= 1e-4
    assert _check_digests(model, d, "train999.graddigest.", True) == 10
    # everything unfrozen, same masks
    for p in model.parameters():
        p.requires_grad = True
    model.zero_grad()
    loss, acc = model(x, y)
    loss.backward()
    assert abs(loss.item() - float(d["unfrozen999.loss"])) <= 1e-4
    assert _check_digests(model, d, "unfrozen999.graddigest.", False) >= 46
    # one Trainer.train step (Adam) with the masks of seed 2024
    model.freeze_all_layers()
    model.zero_grad(set_to_none=True)
    models_mod.set_dropout_masks(masks_to_cuda(O.draw_dropout_masks(cfg, x, seed=2024)))

    class DS:
        loader = [(x, y)]
    trainer = training.Trainer(model=model, config=cfg)
    tr_acc, tr_loss = trainer.train(DS())
    assert abs(tr_loss - float(d["trainer.loss"])) <= 1e-4 and tr_acc == float(d["trainer.acc"])
    for k, v in model.state_dict().items():
        key = "trainer.postadam_digest." + k
        if key in d:
            got, ref = _digest(v), d[key]
            assert abs(got[0] - ref[0]) <= 1e-4 * ref[0], k
            np.testing.assert_allclose(got[2:], ref[2:], atol=1e-4, err_msg=k)   # parameters after one lr = 1e-3 Adam step
    log = open(os.path.join(cfg.folder, "training", "log.csv")).read().splitlines()
    ref_log = meta["log_csv"].splitlines()
    assert log[0] == ref_log[0]
    a, b = log[1].split(","), ref_log[1].split(",")
    assert a[0] == b[0] and a[3] == b[3] and abs(float(a[1]) - float(b[1])) <= 1e-4 and float(a[2]) == float(b[2])


def assert_grads_vs_float64(named_gpu_grads, sd32, sd64, what, rel=1e-4, kink_grads=None):
    """gradients within 1e-4 of the per-tensor max-abs.  Two correct fp32 evaluations of a long recurrence /
    a heavily cancelling sum (the float64 Sinc parameters) can differ from EACH OTHER by more than that, so the arbiter is
    a float64 evaluation of the oracle (O.float64_evaluation: the same formulas without intermediate fp32 rounding): per
    tensor, the GPU may be off the float64 value by rel * max|grad| — or, where the fp32 ORACLE itself is further off than
    half of that, by twice the oracle's own deviation.
    kink_grads: None, or a callable -> (n, {parameter name: slack tensor}): n LeakyReLU inputs lie within fp32 round-off of
    the kink; such an input lands on either side in a correct fp32 evaluation, which changes the gradient of every
    parameter UPSTREAM of it by (1 - 0.2) * dL/dy_i * dx_i/dtheta.  slack = the sum over those inputs of |that|, element
    by element (float64): what any combination of branch choices can move.  Granted on top of the plain bound —
    evaluated only if the plain bound fails, and only for the parameters it names.
    -> (worst gpu deviation / scale, worst oracle deviation / scale, name of the worst)."""
    worst = (0.0, 0.0, "")
    kinks = None
    for k, g in named_gpu_grads:
        if sd64[k].grad is None:
            continue
        ref64 = sd64[k].grad
        scale = max(ref64.abs().max().item(), 1e-9)
        d_gpu = (g.detach().cpu().double() - ref64).abs()
        e_gpu = d_gpu.max().item()
        e_ref = (sd32[k].grad.double() - ref64).abs().max().item()
        bound = max(rel * scale, 2.0 * e_ref)
        if e_gpu > bound and kink_grads is not None:
            if kinks is None:
                kinks = kink_grads()
                print("%s: %d activation input(s) within fp32 round-off of a LeakyReLU kink" % (what, kinks[0]))
            n, slack_of = kinks
            if n > 0 and k in slack_of:
                excess = (d_gpu - slack_of[k]).max().item()
                assert excess <= bound, "%s %s: |gpu - f64| exceeds the kink slack by %.3e (bound %.3e, scale %.3e)" % (
                    what, k, excess, bound, scale)
                worst = max(worst, (max(excess, 0.0) / scale, e_ref / scale, k + " (beyond the kink slack)"))
                continue
        assert e_gpu <= bound, "%s %s: |gpu - f64| = %.3e, |oracle_fp32 - f64| = %.3e, scale %.3e" % (
            what, k, e_gpu, e_ref, scale)
        worst = max(worst, (e_gpu / scale, e_ref / scale, k))
    return worst


@pytest.mark.parametrize("B,seconds,train_math", [(64, 3, "fp32"), (5, 1, "fp32"), (64, 3, "split")])
def test_full_size_batch_vs_oracle(models_mod, tmp_path, monkeypatch, B, seconds, train_math):
    """BASELINE.json configs[2]/[3] shape: B=64 synthetic 3 s utterances through the whole SLU model,
    eval logits and fully-unfrozen train-mode gradients against the CPU oracle.  train_math = "split": the opt-in
    SLU_TRAIN_MATH=split arithmetic of the trainable layers' GEMMs (f16x2 forward products, bf16x3 products with a
    gradient operand: slu_gemm_bf16_a32 / slu_gemm_tn_bf16) must hold the SAME bounds as exact fp32."""
    monkeypatch.setenv("SLU_TRAIN_MATH", train_math)
    meta = {"pretrain_seed": 11, "model_seed": 12}
    cfg, model = _full_model_from_seeds(models_mod, tmp_path, meta)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(B, 16000 * seconds, generator=g)
    y = torch.stack([torch.randint(0, n, (B,), generator=g) for n in cfg.values_per_slot], dim=1)
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
    with torch.no_grad():
        _, _, ref_logits, ref_pred = O.slu_forward(sd, x, y, cfg, None, explicit_gru=False)
    err = maxerr(logits, ref_logits)
    print("B=%d %ds eval logits max-abs deviation: %.3e" % (B, seconds, err))
    assert err <= 1e-4
    assert torch.equal(pred.cpu(), ref_pred)
    # size-independent properties at full size: batch-permutation equivariance and zero-extension
    perm = torch.randperm(B, generator=g)
    with torch.no_grad():
        logits_p, _ = model.predict_intents(x[perm])
    assert maxerr(logits_p, logits[perm.to(logits.device)]) <= 1e-6
    # gradients, everything unfrozen, dropout masks by seed
    for p in model.parameters():
        p.requires_grad = True
    model.train()
    model.zero_grad()
    masks = O.draw_dropout_masks(cfg, x, seed=31)
    models_mod.set_dropout_masks(masks_to_cuda(masks))
    loss, acc = model(x, y)
    loss.backward()
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    rloss, racc, _, _ = O.slu_forward(sdg, x, y, cfg, masks, explicit_gru=False)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-4 and acc.item() == racc.item()
    if train_math == "fp32":
        # the default arithmetic: SURVEY 8(c)'s 1e-4, arbitrated by a float64 evaluation of the oracle
        sd64 = O.to_float64(sd)
        with O.float64_evaluation():
            l64, _, _, _ = O.slu_forward(sd64, x.double(), y, cfg, {k: v.double() for k, v in masks.items()}, explicit_gru=False)
        l64.backward()
        w = assert_grads_vs_float64(((k, p.grad) for k, p in model.named_parameters()), sdg, sd64, "B=%d" % B)
        print("B=%d exact fp32: worst gradient deviation from the float64 oracle %.3e of the tensor's max (fp32 oracle: %.3e) at %s"
              % ((B,) + w))
        return
    worst = 0.0
    for k, p in model.named_parameters():
        if sdg[k].grad is None:
            continue
        scale = max(sdg[k].grad.abs().max().item(), 1e-6)
        e = maxerr(p.grad, sdg[k].grad) / scale
        worst = max(worst, e)
        assert e <= 2e-4, (k, e)      # opt-in split-precision GEMMs of trainable layers: against the fp32 oracle
    print("B=%d SLU_TRAIN_MATH=%s worst relative gradient deviation: %.3e" % (B, train_math, worst))


def test_philox_dropout_training_step_runs_and_is_seeded(models_mod, tmp_path):
    cfg = tiny_cfg(tmp_path)
    model = models_mod.Model(cfg)
    x = 0.1 * torch.randn(4, 2000)
    y = torch.stack([torch.randint(0, n, (4,)) for n in cfg.values_per_slot], dim=1)
    model.train()
    models_mod.set_dropout_masks(None)
    models_mod.set_dropout_seed(5)
    l1, _ = model(x, y)
    l2, _ = model(x, y)                      # counter advanced -> different masks
    models_mod.set_dropout_seed(5)
    l3, _ = model(x, y)
    assert l1.item() == l3.item() and l1.item() != l2.item()
    l1.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_layerwise_iteration_matches_fused_path(models_mod, tmp_path):
    """The reference iterates `for layer in phoneme_layers: out = layer(out)` (models.py:354-359); the
    ModuleList mirror supports the same loop and agrees with the fused stage plan."""
    cfg = tiny_cfg(tmp_path)
    pm = models_mod.PretrainedModel(cfg).eval()
    x = (0.1 * torch.randn(3, 1500)).cuda()
    out = x.unsqueeze(1)
    for layer in pm.phoneme_layers:
        out = layer(out)
    for layer in pm.word_layers:
        out = layer(out)
    fused = pm.compute_features(x)
    assert maxerr(out, fused) <= 1e-6


VARIANTS = {
    # name: overrides of the tiny architecture (every cfg of the reference uses the same hyper-parameters;
    # these exercise the generality the reference's constructor offers)
    "conv_frontend_relu_pool22": dict(use_sincnet=False, cnn_act=["relu", "relu", "leaky_relu"], cnn_max_pool_len=[2, 2, 1]),
    "max_and_none_downsample": dict(phone_downsample_type=["max", "none"], phone_downsample_len=[3, 2],
                                    word_downsample_type=["avg", "max"], word_downsample_len=[1, 2],
                                    intent_downsample_type=["max"], intent_downsample_len=[2]),
    "unidirectional": dict(phone_rnn_bidirectional=False, word_rnn_bidirectional=False, intent_rnn_bidirectional=False),
    "wide_pool_fallback": dict(cnn_max_pool_len=[3, 1, 1]),
    "wide_pools_everywhere": dict(cnn_max_pool_len=[3, 4, 3], cnn_act=["leaky_relu", "relu", "leaky_relu"]),
    "two_intent_layers_three_phone": dict(intent_rnn_num_hidden=[16, 32], intent_rnn_drop=[0.5, 0.25],
                                          intent_downsample_type=["none", "avg"], intent_downsample_len=[1, 2],
                                          phone_rnn_num_hidden=[16, 32, 16], phone_downsample_len=[2, 1, 2],
                                          phone_downsample_type=["avg", "none", "avg"], phone_rnn_drop=[0.5, 0.0, 0.5]),
    # more layers per module than a step has dropout sites for (round 3 raised NotImplementedError past four)
    "five_phone_layers": dict(phone_rnn_num_hidden=[16, 16, 16, 16, 16], phone_downsample_len=[1, 2, 1, 1, 2],
                              phone_downsample_type=["avg"] * 5, phone_rnn_drop=[0.5] * 5),
    # the classifier reads 50-channel rows (not a multiple of four) behind a 32-wide encoder (round-3 advisor finding)
    "intent_unidirectional_h50": dict(intent_rnn_num_hidden=[50], intent_rnn_bidirectional=False),
}


@pytest.mark.parametrize("name", ["five_phone_layers", "intent_unidirectional_h50"])
def test_architecture_variants_train_with_philox_masks(models_mod, tmp_path, name):
    """The same geometries WITHOUT injected masks: the in-kernel Philox paths (dropout sites past the fourth layer of a
    module; the head's fused-dropout gate, which must look at the classifier's input width — 50 here — and not at the
    width of the last GRU layer's input) — two optimisation steps run, the losses are finite and different."""
    import training
    cfg = tiny_cfg(tmp_path, **VARIANTS[name])
    torch.manual_seed(3)
    model = models_mod.Model(cfg)
    models_mod.set_dropout_masks(None)
    models_mod.set_dropout_seed(5)
    model.train()
    g = torch.Generator().manual_seed(8)
    x = 0.1 * torch.randn(5, 2300, generator=g)
    y = torch.stack([torch.randint(0, n, (5,), generator=g) for n in cfg.values_per_slot], dim=1)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
    losses = []
    for _ in range(2):
        opt.zero_grad()
        loss, acc = model(x, y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l and abs(l) < 1e3 for l in losses) and losses[0] != losses[1]
    # (the two ASR heads are not part of the SLU forward: no gradient; every other trainable parameter has a finite one)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and "_linear." not in n]
    assert named and all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in named)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_architecture_variants_vs_oracle(models_mod, tmp_path, name):
    """Logits, loss and every gradient of reduced-size models built from non-default (but legal)
    architecture settings, in train mode with injected dropout masks, against the CPU oracle."""
    cfg = tiny_cfg(tmp_path, **VARIANTS[name])
    torch.manual_seed(3)
    model = models_mod.Model(cfg)
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    x = 0.1 * torch.randn(5, 2300, generator=g)
    y = torch.stack([torch.randint(0, n, (5,), generator=g) for n in cfg.values_per_slot], dim=1)
    masks = O.draw_dropout_masks(cfg, x, seed=21)
    model.train()
    models_mod.set_dropout_masks(masks_to_cuda(masks))
    loss, acc = model(x, y)
    loss.backward()
    rloss, racc, rlogits, rpred = O.slu_forward(sd, x, y, cfg, masks)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-5 and acc.item() == racc.item()
    for k, p in model.named_parameters():
        if sd[k].grad is None:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            continue
        scale = max(sd[k].grad.abs().max().item(), 1e-6)
        assert maxerr(p.grad, sd[k].grad) <= 1e-4 * scale, (name, k)
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
        _, _, elogits, epred = O.slu_forward({k: v.detach() for k, v in sd.items()}, x, y, cfg, None)
    assert maxerr(logits, elogits) <= 1e-5 and torch.equal(pred.cpu(), epred)


def test_ten_second_utterances_vs_oracle(models_mod, tmp_path):
    """BASELINE.json configs[4] shape (LibriSpeech-like 10 s utterances, T = 160 000 -> GRU lengths
    1000/500/250/125/63) in fp32: eval logits and unfrozen train-mode gradients vs the oracle."""
    meta = {"pretrain_seed": 21, "model_seed": 22}
    cfg, model = _full_model_from_seeds(models_mod, tmp_path, meta)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(77)
    x = 0.1 * torch.randn(6, 160000, generator=g)
    y = torch.stack([torch.randint(0, n, (6,), generator=g) for n in cfg.values_per_slot], dim=1)
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
        _, _, ref_logits, ref_pred = O.slu_forward(sd, x, y, cfg, None, explicit_gru=False)
    err = maxerr(logits, ref_logits)
    print("10 s eval logits max-abs deviation: %.3e" % err)
    assert err <= 1e-4 and torch.equal(pred.cpu(), ref_pred)
    for p in model.parameters():
        p.requires_grad = True
    model.train()
    masks = O.draw_dropout_masks(cfg, x, seed=5)
    models_mod.set_dropout_masks(masks_to_cuda(masks))
    loss, _ = model(x, y)
    loss.backward()
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    rloss, _, _, _ = O.slu_forward(sdg, x, y, cfg, masks, explicit_gru=False)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) <= 1e-4
    # 1000 dependent recurrence steps: arbitrated by the float64 evaluation (1e-4, or twice the fp32 oracle's own deviation)
    sd64 = O.to_float64(sd)
    with O.float64_evaluation():
        l64, _, _, _ = O.slu_forward(sd64, x.double(), y, cfg, {k: v.double() for k, v in masks.items()}, explicit_gru=False)
    l64.backward()
    w = assert_grads_vs_float64(((k, p.grad) for k, p in model.named_parameters()), sdg, sd64, "10 s")
    print("10 s: worst gradient deviation from the float64 oracle %.3e of the tensor's max (fp32 oracle: %.3e) at %s" % w)


def test_full_size_asr_pretraining_step_vs_oracle(models_mod, tmp_path):
    """BASELINE.json configs[2]: the full PretrainedModel (Sinc + 2 conv + 4 biGRU layers, phoneme head 42, word head
    VOCABULARY 10 000) forward + backward at B = 64 x 3 s with injected dropout masks against O.asr_forward
    (reference models.py:291-331): both losses and accuracies, every gradient (float64 Sinc parameters included)."""
    cfg = full_cfg(tmp_path, pretraining_type=2)
    torch.manual_seed(31)
    pm = models_mod.PretrainedModel(cfg)
    sd = {k: v.detach().cpu().clone().requires_grad_() for k, v in pm.state_dict().items()}
    g = torch.Generator().manual_seed(32)
    B, T = 64, 48000
    x = 0.1 * torch.randn(B, T, generator=g)
    Tp, Tw = -(-T // 640), -(-T // 2560)
    yp = torch.randint(0, cfg.num_phonemes, (B, Tp), generator=g)
    yw = torch.randint(0, cfg.vocabulary_size, (B, Tw), generator=g)
    yp[torch.rand(B, Tp, generator=g) < 0.1] = -1
    yw[torch.rand(B, Tw, generator=g) < 0.1] = -1
    masks = O.draw_dropout_masks(cfg, x, seed=33, include_intent=False)
    models_mod.set_dropout_masks(masks_to_cuda(masks))
    try:
        pm.train()
        pl, wl, pa, wa = pm(x, yp, yw)
        (pl + wl).backward()
        torch.cuda.synchronize()
    finally:
        models_mod.set_dropout_masks(None)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    rpl, rwl, rpa, rwa = O.asr_forward(sd, x, yp, yw, cfg, masks, explicit_gru=False)
    (rpl + rwl).backward()
    assert abs(pl.item() - rpl.item()) <= 1e-4 and abs(wl.item() - rwl.item()) <= 1e-4, (pl.item(), rpl.item(), wl.item(), rwl.item())
    assert abs(pa.item() - rpa.item()) <= 1e-6 and abs(wa.item() - rwa.item()) <= 1e-6
    # The arbiter is a float64 evaluation of the oracle: the GPU must be within 1e-4 of the float64 gradient, or no further
    # from it than twice the fp32 oracle is.  Round 3 found one conv1 output of 1.15 M within fp32 round-off of the
    # LeakyReLU kink in this draw (-1.3e-7 in the fp32 oracle, +2.2e-7 on the GPU): the two fp32 evaluations then take
    # different branches THERE, both legitimately, and the tiny float64 Sinc-parameter gradients move by ~1e-3 of their
    # max (round 5, measured against float64: GPU 8.2e-4 off, fp32 oracle 5e-6 off — the oracle happens to share float64's
    # branch).  Rounds 3-4 widened the bound to 3e-3 for parameters upstream of such an element; now the claim is PROVEN:
    # where the plain bound fails, float64 gradients are evaluated with the kink-adjacent inputs (|x| < 1e-6) forced to
    # either branch, and only their element-wise difference is granted on top of the plain bound.
    sd64 = O.to_float64({k: v.detach() for k, v in sd.items()})
    st64 = {}
    with O.float64_evaluation():
        p64, w64, _, _ = O.asr_forward(sd64, x.double(), yp, yw, cfg, {k: v.double() for k, v in masks.items()},
                                       explicit_gru=False, stages_out=st64)
    for c in (1, 2):
        st64["cnn%d" % c].retain_grad()
    (p64 + w64).backward(retain_graph=True)
    for k, p in pm.named_parameters():
        assert sd[k].grad is not None and p.grad is not None and p.grad.dtype == sd[k].grad.dtype, k

    # how far from 0 a conv output may be and still change sign between two correct fp32 evaluations: its worst-case
    # round-off, K * eps * max|term| = 400 * 6e-8 * 0.3 = 7e-6 for a 400-tap sum of these magnitudes, plus the propagated
    # round-off of its input (block 0's output, ~1e-6 per element, times sum|w| ~ 4)
    KINK_BAND = 1e-5

    def kink_grads():
        """For every conv-block pre-activation within KINK_BAND of the LeakyReLU kink (float64 evaluation): its branch choice moves
        the gradient of each upstream parameter by 0.8 * dL/dy_i * dx_i/dtheta — computed exactly, one small backward pass
        through the front end per such input."""
        idx = O.phoneme_layer_index(cfg)
        names = {0: ["phoneme_layers.%d.filt_b1" % idx["conv0"], "phoneme_layers.%d.filt_band" % idx["conv0"]],
                 1: ["phoneme_layers.%d.weight" % idx["conv1"], "phoneme_layers.%d.bias" % idx["conv1"]],
                 2: ["phoneme_layers.%d.weight" % idx["conv2"], "phoneme_layers.%d.bias" % idx["conv2"]]}
        slack, n = {}, 0
        for c in (1, 2):                                   # block 0's output is |x| >= 0 (Abs in front of the pool): no kink there
            pre, post = st64["conv%d" % c], st64["cnn%d" % c]
            near = (pre.detach().abs() < KINK_BAND).nonzero()
            upstream = [k for cc in range(c + 1) for k in names[cc]]
            for pos in near:
                n += 1
                pos = tuple(int(v) for v in pos)
                delta = post.grad[pos].item()
                gs = torch.autograd.grad(pre[pos], [sd64[k] for k in upstream], retain_graph=True, allow_unused=True)
                for k, g in zip(upstream, gs):
                    if g is not None:
                        slack[k] = slack.get(k, 0.0) + (0.8 * delta * g).abs()
        print("   kink slack, per parameter, as a fraction of the tensor's max |grad|: %s"
              % {k: "%.2e" % (float(v.max()) / max(float(sd64[k].grad.abs().max()), 1e-30)) for k, v in slack.items()})
        return n, slack

    worst = assert_grads_vs_float64(((k, p.grad) for k, p in pm.named_parameters()), sd, sd64, "ASR", kink_grads=kink_grads)
    n = sum(1 for _ in pm.named_parameters())
    print("full-size ASR step: losses %.5f / %.5f (oracle %.5f / %.5f), %d gradients, worst deviation from the float64 oracle "
          "%.2e of the tensor's max (fp32 oracle %.2e) at %s"
          % (pl.item(), wl.item(), rpl.item(), rwl.item(), n, worst[0], worst[1], worst[2]))
