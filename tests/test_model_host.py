"""CPU: host-side logic of models.py that needs no GPU — state_dict layout and bit-identical
initialisation vs the reference (fixture g6), layer names, freezing schedule, loud failure
without a device."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import slu_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(tmp_path, **kw):
    c = O.OracleConfig(**kw)
    c.folder = str(tmp_path)
    c.Sy_intent = {"action": {"a%d" % i: i for i in range(6)}, "object": {"o%d" % i: i for i in range(14)},
                   "location": {"l%d" % i: i for i in range(4)}}
    c.starting_unfreezing_index = {0: 1 + 2 + 2 + 3, 1: 3, 2: 1, 3: 1}[c.pretraining_type]
    return c


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


@pytest.fixture()
def no_gpu(monkeypatch):
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)


def test_state_dict_is_bit_identical_to_the_reference_under_the_same_seeds(tmp_path, no_gpu):
    import models
    d = dict(np.load(os.path.join(G, "g6_full_model.npz")))
    meta = json.loads(bytes(d["meta_json"]).decode())
    cfg = _cfg(tmp_path, pretraining_type=2)
    os.makedirs(tmp_path / "pretraining")
    torch.manual_seed(meta["pretrain_seed"])
    pre = models.PretrainedModel(cfg)
    assert {k: sha(v) for k, v in pre.state_dict().items()} == meta["pretrained_sha256"]
    torch.save(pre.state_dict(), tmp_path / "pretraining" / "model_state.pth")
    torch.manual_seed(meta["model_seed"])
    model = models.Model(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == list(meta["model_sha256"].keys())           # same keys, same order
    assert {k: sha(v) for k, v in sd.items()} == meta["model_sha256"]
    assert {k: str(v.dtype) for k, v in sd.items()} == meta["dtypes"]
    assert {k: list(v.shape) for k, v in sd.items()} == meta["shapes"]
    # pretraining_type != 0 -> encoder frozen, heads + intent module trainable (models.py:672-673)
    frozen = {k for k, p in model.named_parameters() if not p.requires_grad}
    assert all(k.startswith("pretrained_model.phoneme_layers") or k.startswith("pretrained_model.word_layers") for k in frozen)
    assert any(k.startswith("pretrained_model.phoneme_linear") for k, p in model.named_parameters() if p.requires_grad)
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        model.print_frozen()
    assert buf.getvalue().splitlines() == meta["print_frozen"]


def test_layer_names_and_indices_match_the_reference(tmp_path, no_gpu):
    import models
    pm = models.PretrainedModel(_cfg(tmp_path, pretraining_type=0))
    names = [l.name for l in pm.phoneme_layers]
    assert names == ["sinc0", "abs0", "pool0", "act0", "dropout0", "conv1", "pool1", "act1", "dropout1",
                     "conv2", "pool2", "act2", "dropout2", "ncl2nlc",
                     "phone_rnn0", "phone_rnn_select0", "phone_dropout0", "phone_downsample0",
                     "phone_rnn1", "phone_rnn_select1", "phone_dropout1", "phone_downsample1"]
    assert [l.name for l in pm.word_layers] == [
        "word_rnn0", "word_rnn_select0", "word_dropout0", "word_downsample0",
        "word_rnn1", "word_rnn_select1", "word_dropout1", "word_downsample1"]
    assert pm.phoneme_layers[0].filt_b1.dtype == torch.float64
    m = models.Model(_cfg(tmp_path, pretraining_type=0))
    assert [l.name for l in m.intent_layers] == ["intent_rnn0", "intent_rnn_select0", "intent_dropout0",
                                                 "intent_downsample0", "final_classifier", "final_pool"]
    assert sum(p.numel() for p in m.parameters()) == 3960954


def _unfrozen(model):
    import models
    out = []
    for layer in list(model.pretrained_model.phoneme_layers) + list(model.pretrained_model.word_layers):
        if models.has_params(layer) and not models.is_frozen(layer):
            out.append(layer.name)
    return out


def test_gradual_unfreezing_schedule_matches_reference(tmp_path, no_gpu):
    """Fixture g9 (generated from the reference's Model.unfreeze_one_layer, models.py:754-795):
    unfrozen layer names after each call for unfreezing types 0/1/2 and pre-training types 1/2."""
    import models
    g9 = json.load(open(os.path.join(G, "g9_unfreeze.json")))
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    tiny = dict(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16], intent_rnn_num_hidden=[16],
                vocabulary_size=50, num_phonemes=11, values_per_slot=[3, 4, 2])
    for key, seq in g9.items():
        ptype, utype = int(key[5]), int(key[-1])
        cfg = _cfg(tmp_path, pretraining_type=ptype, unfreezing_type=utype, **tiny)
        cfg.Sy_intent = {"action": {"a%d" % i: i for i in range(3)}, "object": {"o%d" % i: i for i in range(4)},
                         "location": {"l%d" % i: i for i in range(2)}}
        cfg.starting_unfreezing_index = {1: 1 + len(cfg.word_rnn_num_hidden), 2: 1}[ptype]
        torch.save(models.PretrainedModel(cfg).state_dict(), tmp_path / "pretraining" / "model_state.pth")
        model = models.Model(cfg)
        assert _unfrozen(model) == []
        for want in seq:
            model.unfreeze_one_layer()
            assert _unfrozen(model) == want, (key, want)


def test_no_pretraining_leaves_everything_trainable_and_skips_checkpoint(tmp_path, no_gpu):
    import models
    model = models.Model(_cfg(tmp_path, pretraining_type=0))
    assert all(p.requires_grad for p in model.parameters())


def test_forward_without_gpu_fails_loudly(tmp_path, no_gpu):
    import models
    from slu_hip.lib import SluHipError
    model = models.Model(_cfg(tmp_path, pretraining_type=0))
    x = torch.zeros(2, 1600)
    with pytest.raises(SluHipError, match="no CPU fallback"):
        model(x, torch.zeros(2, 3, dtype=torch.long))
    with pytest.raises(SluHipError, match="no CPU fallback"):
        model.pretrained_model.compute_features(x)


def test_bad_downsample_method_exits_like_the_reference(capsys):
    import models
    with pytest.raises(SystemExit):
        models.Downsample(method="median", factor=2)
    assert "downsampling method must be one of" in capsys.readouterr().out


def test_decode_intents_mapping(tmp_path, no_gpu, monkeypatch):
    import models
    model = models.Model(_cfg(tmp_path, pretraining_type=0))
    monkeypatch.setattr(model, "predict_intents", lambda x: (None, torch.tensor([[1, 13, 0], [5, 0, 3]])))
    assert model.decode_intents(None) == [["a1", "o13", "l0"], ["a5", "o0", "l3"]]


@pytest.mark.parametrize("tag,kw", [("a", {}), ("b", {"num_intent_encoder_layers": 2, "num_intent_decoder_layers": 3})])
def test_seq2seq_state_dict_is_bit_identical_to_the_reference_under_the_same_seed(tmp_path, no_gpu, tag, kw):
    """Model(config.seq2seq): same state_dict keys IN ORDER, shapes and — under the fixture's seed — the same values
    as the reference's Model (g7), i.e. the same modules constructed in the same order (models.py:720-725); layer
    names of the seq2seq encoder / decoder as the reference sets them."""
    import models
    d = dict(np.load(os.path.join(G, "g7_seq2seq_%s.npz" % tag)))
    labels = json.loads(bytes(d["labels_json"]).decode())
    cfg = O.OracleConfig(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                         phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16], intent_rnn_num_hidden=[16],
                         vocabulary_size=50, num_phonemes=11, values_per_slot=[3, 4, 2], pretraining_type=0,
                         seq2seq=True, intent_encoder_dim=12, num_intent_encoder_layers=1, intent_decoder_dim=20,
                         num_intent_decoder_layers=2, intent_decoder_key_dim=10, intent_decoder_value_dim=14)
    for k, v in kw.items():
        setattr(cfg, k, v)
    cfg.folder, cfg.Sy_intent, cfg.starting_unfreezing_index = str(tmp_path), labels, 1
    torch.manual_seed(70 + len(kw))
    model = models.Model(cfg)
    sd = model.state_dict()
    assert list(sd.keys()) == json.loads(bytes(d["sd_keys_json"]).decode())
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), d["sd." + k]), k
    assert model.seq2seq and model.SOS == 0 and model.num_labels == len(labels) and model.intent_layers == []
    assert [l.name for l in model.encoder.layers][:3] == ["intent_encoder_rnn0", "intent_encoder_rnn_select0", "intent_encoder_dropout0"]
    assert [l.name for l in model.decoder.rnn.layers][:4] == ["gru0", "dropout0", "gru1", "dropout1"]
    assert float(model.decoder.attention.scale_factor) == float(torch.sqrt(torch.tensor(10).float()))
    with pytest.raises(Exception, match="no CPU fallback|GPU"):
        model(torch.zeros(2, 1800), torch.zeros(2, 3, len(labels)))
    assert model.one_hot_to_string(torch.eye(len(labels))[[0, 3, 4, len(labels) - 1]], labels) == labels[3] + labels[4]
