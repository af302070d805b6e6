"""CPU: pins oracle/slu_oracle.py against the fixtures generated from the imported reference
(tests/golden/make_goldens.py).  Tolerances are the fp32 noise floor between two formulations."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import slu_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return dict(np.load(os.path.join(G, name)))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_g1_sinc_mel_init_and_filters():
    d = load("g1_sinc_filters.npz")
    b1, band = O.sinc_mel_init(80, 16000)
    assert np.array_equal(b1, d["b1_default"]) and np.array_equal(band, d["band_default"])
    f = O.sinc_filters(T(b1), T(band), 401, 16000)
    assert f.dtype == torch.float32
    np.testing.assert_allclose(f.numpy(), d["filters_default"], rtol=0, atol=1e-6)
    f = O.sinc_filters(T(d["b1_perturbed"]), T(d["band_perturbed"]), 401, 16000)
    np.testing.assert_allclose(f.numpy(), d["filters_perturbed"], rtol=0, atol=1e-6)
    b1s, bands = O.sinc_mel_init(8, 16000)
    assert np.array_equal(b1s, d["b1_small"])
    f = O.sinc_filters(T(b1s), T(bands), 41, 16000)
    np.testing.assert_allclose(f.numpy(), d["filters_small"], rtol=0, atol=1e-6)


def test_g1_sinc_filter_param_grads_are_float64():
    d = load("g1_sinc_filters.npz")
    b1 = T(d["b1_perturbed"]).requires_grad_()
    band = T(d["band_perturbed"]).requires_grad_()
    f = O.sinc_filters(b1, band, 401, 16000)
    (f * T(d["G"])).sum().backward()
    assert b1.grad.dtype == torch.float64
    for got, ref in ((b1.grad, d["grad_b1_perturbed"]), (band.grad, d["grad_band_perturbed"])):
        scale = np.abs(ref).max()
        np.testing.assert_allclose(got.numpy(), ref, rtol=0, atol=2e-4 * scale)


def test_g2_frontend_stages():
    d = load("g2_frontend.npz")
    cfg = O.OracleConfig()
    sd = {k: T(v) for k, v in d.items() if k.startswith("phoneme_layers")}
    x = T(d["x"])
    out = O.sinc_layer(x.unsqueeze(1), sd["phoneme_layers.0.filt_b1"], sd["phoneme_layers.0.filt_band"],
                       401, 16000, 80, 200)
    np.testing.assert_allclose(out.numpy(), d["after_sinc0"], atol=2e-6)
    out2 = O.sinc_layer(x.unsqueeze(1), sd["phoneme_layers.0.filt_b1"], sd["phoneme_layers.0.filt_band"],
                        401, 16000, 80, 200, faithful_loop=True)
    assert torch.equal(out, out2)
    # run the CNN part of encoder_stages by giving it only CNN weights (GRU part is skipped on KeyError)
    with pytest.raises(KeyError):
        O.encoder_stages(sd, x, cfg)
    st = {}
    h = x.unsqueeze(1)
    h = O.sinc_layer(h, sd["phoneme_layers.0.filt_b1"], sd["phoneme_layers.0.filt_band"], 401, 16000, 80, 200)
    h = O.activation(O.max_pool_ceil(torch.abs(h), 2), "leaky_relu")
    np.testing.assert_allclose(h.numpy(), d["after_dropout0"], atol=2e-6)
    h = torch.nn.functional.conv1d(h, sd["phoneme_layers.5.weight"], sd["phoneme_layers.5.bias"], padding=2)
    h = O.activation(h, "leaky_relu")
    np.testing.assert_allclose(h.numpy(), d["after_dropout1"], atol=2e-6)
    h = torch.nn.functional.conv1d(h, sd["phoneme_layers.9.weight"], sd["phoneme_layers.9.bias"], padding=2)
    h = O.activation(h, "leaky_relu")
    np.testing.assert_allclose(h.numpy(), d["after_dropout2"], atol=2e-6)


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(G) if f.startswith("g3_gru")))
@pytest.mark.parametrize("explicit", [True, False])
def test_g3_gru(name, explicit):
    d = load(name)
    bi = name.endswith("_bi.npz")
    p = {k: T(v).requires_grad_() for k, v in d.items() if k.startswith(("weight_", "bias_"))}
    x = T(d["x"]).requires_grad_()
    out = O.gru_layer(x, p, bidirectional=bi, explicit=explicit)
    np.testing.assert_allclose(out.detach().numpy(), d["out"], atol=2e-6)
    (out * T(d["g"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), d["dx"], atol=1e-5 * max(1.0, np.abs(d["dx"]).max()))
    for k, v in p.items():
        ref = d["grad_" + k]
        np.testing.assert_allclose(v.grad.numpy(), ref, atol=1e-5 * max(1.0, np.abs(ref).max()), err_msg=k)


def test_g4_downsample_and_dropout():
    d = load("g4_downsample.npz")
    for T_ in (25, 75, 6, 1):
        x = T(d["x_T%d" % T_])
        for method in ("none", "avg", "max"):
            for factor in (1, 2, 3):
                y = O.downsample(x, method, factor)
                ref = d["y_T%d_%s_%d" % (T_, method, factor)]
                assert y.shape == ref.shape
                np.testing.assert_allclose(y.numpy(), ref, atol=1e-7)
    with pytest.raises(ValueError):
        O.downsample(T(d["x_T6"]), "median", 2)
    for seed in (11, 12):
        torch.manual_seed(seed)
        mask = torch.empty(4, 9, 16).bernoulli_(0.5)
        y = O.dropout_with_mask(torch.ones(4, 9, 16), 0.5, mask)
        assert np.array_equal(y.numpy(), d["dropout_seed%d" % seed])


def tiny_cfg(**kw):
    c = O.OracleConfig(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                       phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16],
                       intent_rnn_num_hidden=[16], vocabulary_size=50, num_phonemes=11,
                       values_per_slot=[3, 4, 2], pretraining_type=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _model_case(d, tag, masks, cfg):
    sd = {k[3:]: T(v).requires_grad_() for k, v in d.items() if k.startswith("sd.")}
    x, y = T(d["x"]), T(d["y"])
    loss, acc, logits, pred = O.slu_forward(sd, x, y, cfg, masks)
    loss.backward()
    assert abs(loss.item() - d[tag + ".loss"]) < 2e-6 * max(1, abs(d[tag + ".loss"]))
    assert acc.item() == d[tag + ".acc"]
    for k, v in sd.items():
        key = tag + ".grad." + k
        if key in d:
            ref = d[key]
            np.testing.assert_allclose(v.grad.numpy(), ref, atol=1e-5 * max(1e-3, np.abs(ref).max()), err_msg=k)
    return logits, pred


def test_g5_tiny_model_eval_and_train():
    d = load("g5_tiny_model.npz")
    cfg = tiny_cfg()
    logits, pred = _model_case(d, "eval", None, cfg)
    np.testing.assert_allclose(logits.detach().numpy(), d["eval.logits"], atol=2e-6)
    assert np.array_equal(pred.numpy(), d["eval.pred"])
    sd = {k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")}
    st = O.encoder_stages(sd, T(d["x"]), cfg, prefix="pretrained_model.")
    np.testing.assert_allclose(st["features"].numpy(), d["eval.features"], atol=2e-6)
    masks = O.draw_dropout_masks(cfg, T(d["x"]), seed=77)
    _model_case(d, "train77", masks, cfg)


def test_g5_tiny_asr_heads():
    d = load("g5_tiny_asr.npz")
    for ptype in (2, 1):
        cfg = tiny_cfg(pretraining_type=ptype)
        sd = {k[3:]: T(v).requires_grad_() for k, v in d.items() if k.startswith("sd.")}
        pl, wl, pa, wa = O.asr_forward(sd, T(d["x"]), T(d["y_phoneme"]), T(d["y_word"]), cfg)
        tag = "pt%d." % ptype
        assert abs(pl.item() - d[tag + "phoneme_loss"]) < 5e-6
        assert abs(float(wl.sum()) - float(d[tag + "word_loss"].sum())) < 5e-6
        assert pa.item() == d[tag + "phoneme_acc"]
        assert float(wa.sum()) == float(d[tag + "word_acc"].sum())
        (pl + wl.sum() if ptype == 2 else pl).backward()
        for k, v in sd.items():
            key = tag + "grad." + k
            if key in d:
                ref = d[key]
                np.testing.assert_allclose(v.grad.numpy(), ref, atol=1e-5 * max(1e-3, np.abs(ref).max()), err_msg=k)


def _digest(t):
    f = t.detach().double().flatten()
    return np.concatenate([[f.norm().item(), f.sum().item()], f[:8].numpy()])


def test_g6_full_size_init_reproduction_and_forward():
    """BASELINE.json configs[0] on the CPU path: the oracle re-draws the reference's weights
    (same seed, same RNG order) bit-identically and reproduces logits / loss / grads."""
    import hashlib
    d = load("g6_full_model.npz")
    meta = json.loads(bytes(d["meta_json"]).decode())
    cfg = O.OracleConfig(pretraining_type=2)
    torch.manual_seed(meta["pretrain_seed"])
    pre = O.init_pretrained_state_dict(cfg)

    def sha(t):
        return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()

    assert {k: sha(v) for k, v in pre.items()} == meta["pretrained_sha256"]
    torch.manual_seed(meta["model_seed"])
    sd = O.init_model_state_dict(cfg, pretrained_sd=pre)
    assert {k: sha(v) for k, v in sd.items()} == meta["model_sha256"]
    assert {k: str(v.dtype) for k, v in sd.items()} == meta["dtypes"]
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(16, 16000, generator=g)
    y = torch.stack([torch.randint(0, n, (16,), generator=g) for n in cfg.values_per_slot], dim=1)
    assert np.array_equal(y.numpy(), d["y"])
    loss, acc, logits, pred = O.slu_forward(sd, x, y, cfg, None, explicit_gru=False)
    np.testing.assert_allclose(logits.numpy(), d["eval.logits"], atol=5e-6)
    assert np.array_equal(pred.numpy(), d["eval.pred"])
    assert abs(loss.item() - d["eval.loss"]) < 1e-5 and acc.item() == d["eval.acc"]
    # train-mode backward with everything unfrozen, dropout seed 999
    sdg = {k: v.clone().requires_grad_() for k, v in sd.items()}
    masks = O.draw_dropout_masks(cfg, x, seed=999)
    loss, acc, _, _ = O.slu_forward(sdg, x, y, cfg, masks, explicit_gru=False)
    assert abs(loss.item() - d["unfrozen999.loss"]) < 1e-5
    loss.backward()
    n = 0
    for k, v in sdg.items():
        key = "unfrozen999.graddigest." + k
        if key in d:
            ref = d[key]
            got = _digest(v.grad)
            assert abs(got[0] - ref[0]) <= 2e-4 * max(ref[0], 1e-6), k       # L2 norm
            np.testing.assert_allclose(got[2:], ref[2:], atol=2e-4 * max(np.abs(ref[2:]).max(), 1e-6), err_msg=k)
            n += 1
    assert n >= 40


# ---- seq2seq head (fixture g7: tiny Model with config.seq2seq, generated from the imported reference) ----------
def seq2seq_cfg(d, **kw):
    c = tiny_cfg(seq2seq=True, intent_encoder_dim=12, num_intent_encoder_layers=1, intent_decoder_dim=20,
                 num_intent_decoder_layers=2, intent_decoder_key_dim=10, intent_decoder_value_dim=14)
    c.Sy_intent = json.loads(bytes(d["labels_json"]).decode())
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def one_hot(idx, V):
    y = torch.zeros(idx.shape[0], idx.shape[1], V)
    y.scatter_(2, idx.unsqueeze(2), 1.0)
    return y


@pytest.mark.parametrize("tag,kw", [("a", {}), ("b", {"num_intent_encoder_layers": 2, "num_intent_decoder_layers": 3})])
def test_g7_seq2seq_loss_grads_and_init(tag, kw):
    d = load("g7_seq2seq_%s.npz" % tag)
    cfg = seq2seq_cfg(d, **kw)
    V = len(cfg.Sy_intent)
    x, idx = T(d["x"]), T(d["y_idx"]).long()
    y = one_hot(idx, V)
    for mode in ("eval", "train91"):
        sd = {k[3:]: T(v).requires_grad_() for k, v in d.items() if k.startswith("sd.")}
        masks = None if mode == "eval" else O.draw_seq2seq_masks(cfg, x, idx.shape[1], seed=91)
        loss, log_p = O.seq2seq_forward(sd, x, y, cfg, masks)
        loss.backward()
        assert abs(loss.item() - float(d[mode + ".loss"])) < 3e-6 * max(1.0, abs(float(d[mode + ".loss"]))), mode
        if mode == "eval":
            np.testing.assert_allclose(log_p.detach().numpy(), d["eval.log_p"], atol=2e-5)
        n = 0
        for k, v in sd.items():
            key = mode + ".grad." + k
            if key in d:
                np.testing.assert_allclose(v.grad.numpy(), d[key], atol=2e-5 * max(1e-3, np.abs(d[key]).max()), err_msg=k)
                n += 1
        assert n >= 40
    # the parameter set and its construction order (RNG consumption): the head's keys follow the encoder's
    keys = json.loads(bytes(d["sd_keys_json"]).decode())
    torch.manual_seed(70 + len(kw))                                 # the generator's seed for this variant
    O.init_pretrained_state_dict(cfg)                               # the encoder draws first (models.py:661)
    head = O.init_seq2seq_state_dict(cfg, V)
    want = [k for k in keys if not k.startswith("pretrained_model.")]
    assert sorted(head.keys()) == sorted(want)
    for k in want:
        assert np.array_equal(head[k].numpy(), d["sd." + k]), k     # same draws in the same order


def test_g7_seq2seq_beam_search_and_strings():
    d = load("g7_seq2seq_a.npz")
    cfg = seq2seq_cfg(d)
    V = len(cfg.Sy_intent)
    sd = {k[3:]: T(v) for k, v in d.items() if k.startswith("sd.")}
    with torch.no_grad():
        st = O.encoder_stages(sd, T(d["x"]), cfg, prefix="pretrained_model.")
        enc = O.seq2seq_encoder(sd, st["features"], cfg)
        np.testing.assert_allclose(enc.numpy(), d["eval.encoder_out"], atol=2e-6)
        scores, beam = O.seq2seq_infer(sd, enc, cfg, V, beam=4, max_len=200)
    np.testing.assert_allclose(scores.numpy(), d["beam.scores"], rtol=2e-5, atol=2e-4)
    assert np.array_equal(beam.max(dim=3)[1].numpy(), d["beam.idx"])
    strings = [O.one_hot_to_string(beam[0, i], cfg.Sy_intent) for i in range(beam.shape[1])]
    assert strings == json.loads(bytes(d["beam.strings_json"]).decode())
    truth = [O.one_hot_to_string(one_hot(T(d["y_idx"]).long(), V)[i], cfg.Sy_intent) for i in range(3)]
    assert truth == json.loads(bytes(d["truth_strings_json"]).decode())
