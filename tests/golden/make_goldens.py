#!/usr/bin/env python3
"""
Generates the golden fixtures in this directory by IMPORTING the unmodified reference
(lorenlugosch/end-to-end-SLU) from a path given on the command line (default /root/reference).

Runs only in the authoring container: the reference never travels to the GPU box, only the
small data fixtures written here do.  No reference source is copied: this script calls the
reference's public classes/functions (models.SincLayer, models.PretrainedModel, models.Model,
models.Downsample, data.read_config, training.Trainer) on seeded inputs and stores
inputs + outputs.

    python tests/golden/make_goldens.py [/root/reference]

Fixtures:
  g1_sinc_filters.npz   SincLayer filterbank for default and perturbed float64 parameters, + grads
  g2_frontend.npz       x(2,8000) through sinc0 / pool0+act0 / conv1+act1 / conv2+act2
  g3_gru_*.npz          nn.GRU cases (I,H,T,B): weights, x, output, grads for loss=(out*g).sum()
  g4_downsample.npz     Downsample none/avg/max on odd T; dropout mask seeds + checksums
  g5_tiny_model.npz     reduced-size Model: full state_dict, x, y, eval & train-mode loss/acc/logits
                        and every parameter gradient (train mode: dropout masks by seed)
  g5_tiny_asr.npz       reduced-size PretrainedModel.forward (ASR losses) + grads
  g7_seq2seq_{a,b}.npz  tiny seq2seq Model (config.seq2seq): teacher-forced loss, log p(y|x), every gradient (eval and
                        train mode by seed), beam-search scores / label sequences / decoded strings
  g6_full_model.npz     no_unfreezing.cfg architecture, seed 1234, x=randn(16,16000): state_dict
                        SHA-256 per tensor (weights are re-drawn, not stored), eval logits/loss/acc,
                        train-mode loss + grad digests, one Trainer.train step (post-Adam digests,
                        log.csv row)
  g9_unfreeze.json      gradual-unfreezing schedule (layer names unfrozen after each call)
  g8_config.json        for all 29 reference cfgs: cfg text (input) and vars(read_config(cfg))
                        or the exception the reference raises (output)
"""
import hashlib
import io
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
for m in ("torchaudio", "soundfile", "textgrid"):          # not installed; only dataset code uses them
    sys.modules.setdefault(m, types.ModuleType(m))

import models as ref_models                                  # noqa: E402
import data as ref_data                                      # noqa: E402
import training as ref_training                              # noqa: E402

torch.set_num_threads(8)


def sha(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def npd(t):
    return t.detach().cpu().numpy()


def digest(t):
    """Small, order-sensitive fingerprint of a float tensor: L2 norm, sum, first 8 values."""
    f = t.detach().double().flatten()
    return np.concatenate([[f.norm().item(), f.sum().item()], f[:8].numpy(),
                           np.zeros(max(0, 8 - f.numel()))]).astype(np.float64)


# ------------------------------------------------------------------------------------------
def g1():
    lay = ref_models.SincLayer(80, 401, 16000, stride=80, padding=200)
    x = torch.zeros(1, 1, 800)
    out = {}

    def filters_of(layer):
        # SincLayer has no filter accessor; recover the bank exactly by convolving unit impulses:
        # conv1d(delta_k) at an output position reads filters[:, j] — instead use stride 1 probe.
        probe = ref_models.SincLayer(80, 401, 16000, stride=1, padding=0)
        probe.filt_b1.data.copy_(layer.filt_b1.data)
        probe.filt_band.data.copy_(layer.filt_band.data)
        eye = torch.zeros(401, 1, 401)
        for k in range(401):
            eye[k, 0, k] = 1.0
        return probe(eye)[:, :, 0].t().contiguous(), probe   # (80,401)

    f0, _ = filters_of(lay)
    out["b1_default"] = npd(lay.filt_b1)
    out["band_default"] = npd(lay.filt_band)
    out["filters_default"] = npd(f0)
    g = torch.Generator().manual_seed(7)
    lay2 = ref_models.SincLayer(80, 401, 16000, stride=80, padding=200)
    lay2.filt_b1.data.mul_(1 + 0.3 * torch.randn(80, generator=g, dtype=torch.float64))
    lay2.filt_band.data.mul_(1 + 0.3 * torch.randn(80, generator=g, dtype=torch.float64))
    lay2.filt_b1.data[3] *= -1                                # exercise abs()
    lay2.filt_band.data[5] *= -1
    f1, probe = filters_of(lay2)
    out["b1_perturbed"] = npd(lay2.filt_b1)
    out["band_perturbed"] = npd(lay2.filt_band)
    out["filters_perturbed"] = npd(f1)
    # gradient of sum(filters * G) wrt the two float64 parameters (through the reference autograd)
    G = torch.randn(80, 401, generator=g)
    eye = torch.zeros(401, 1, 401)
    for k in range(401):
        eye[k, 0, k] = 1.0
    filt = probe(eye)[:, :, 0].t()
    (filt * G).sum().backward()
    out["G"] = npd(G)
    out["grad_b1_perturbed"] = npd(probe.filt_b1.grad)
    out["grad_band_perturbed"] = npd(probe.filt_band.grad)
    # small-shape bank (used by the tiny model): N_filt=8, Filt_dim=41
    probe3 = ref_models.SincLayer(8, 41, 16000, stride=1, padding=0)
    eye3 = torch.zeros(41, 1, 41)
    for k in range(41):
        eye3[k, 0, k] = 1.0
    out["b1_small"] = npd(probe3.filt_b1)
    out["band_small"] = npd(probe3.filt_band)
    out["filters_small"] = npd(probe3(eye3)[:, :, 0].t())
    np.savez_compressed(os.path.join(OUT, "g1_sinc_filters.npz"), **out)


# ------------------------------------------------------------------------------------------
class Cfg:
    pass


def base_cfg(**kw):
    c = Cfg()
    c.use_sincnet = True
    c.fs = 16000
    c.cnn_N_filt = [80, 60, 60]
    c.cnn_len_filt = [401, 5, 5]
    c.cnn_stride = [80, 1, 1]
    c.cnn_max_pool_len = [2, 1, 1]
    c.cnn_act = ["leaky_relu"] * 3
    c.cnn_drop = [0.0, 0.0, 0.0]
    c.phone_rnn_num_hidden = [128, 128]
    c.phone_downsample_len = [2, 2]
    c.phone_downsample_type = ["avg", "avg"]
    c.phone_rnn_drop = [0.5, 0.5]
    c.phone_rnn_bidirectional = True
    c.word_rnn_num_hidden = [128, 128]
    c.word_downsample_len = [2, 2]
    c.word_downsample_type = ["avg", "avg"]
    c.word_rnn_drop = [0.5, 0.5]
    c.word_rnn_bidirectional = True
    c.vocabulary_size = 10000
    c.intent_rnn_num_hidden = [128]
    c.intent_downsample_len = [1]
    c.intent_downsample_type = ["none"]
    c.intent_rnn_drop = [0.5]
    c.intent_rnn_bidirectional = True
    c.pretraining_type = 0
    c.unfreezing_type = 0
    c.starting_unfreezing_index = 1
    c.num_phonemes = 42
    c.values_per_slot = [6, 14, 4]
    c.Sy_intent = {"action": {str(i): i for i in range(6)},
                   "object": {str(i): i for i in range(14)},
                   "location": {str(i): i for i in range(4)}}
    c.seq2seq = False
    c.folder = "."
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def g2():
    torch.manual_seed(1234)
    cfg = base_cfg()
    pm = ref_models.PretrainedModel(cfg)
    pm.eval()
    x = 0.1 * torch.randn(2, 8000)
    out = {"x": npd(x)}
    for k in ("phoneme_layers.5.weight", "phoneme_layers.5.bias",
              "phoneme_layers.9.weight", "phoneme_layers.9.bias",
              "phoneme_layers.0.filt_b1", "phoneme_layers.0.filt_band"):
        out[k] = npd(pm.state_dict()[k])
    h = x.unsqueeze(1)
    for i, layer in enumerate(pm.phoneme_layers):
        h = layer(h)
        if layer.name in ("sinc0", "dropout0", "dropout1", "dropout2"):
            out["after_" + layer.name] = npd(h)
        if layer.name == "dropout2":
            break
    np.savez_compressed(os.path.join(OUT, "g2_frontend.npz"), **out)


def g3():
    for (I, H, T, B, bi) in [(60, 128, 100, 4, True), (256, 128, 50, 4, True), (8, 16, 7, 3, True),
                             (12, 32, 9, 5, False)]:
        torch.manual_seed(100 + I + H + T)
        gru = torch.nn.GRU(input_size=I, hidden_size=H, batch_first=True, bidirectional=bi)
        x = torch.randn(B, T, I, requires_grad=True)
        out, _ = gru(x)
        g = torch.randn_like(out)
        (out * g).sum().backward()
        d = {"x": npd(x), "out": npd(out), "g": npd(g), "dx": npd(x.grad)}
        for k, v in gru.named_parameters():
            d[k] = npd(v)
            d["grad_" + k] = npd(v.grad)
        np.savez_compressed(os.path.join(OUT, "g3_gru_I%d_H%d_T%d_B%d_%s.npz" % (I, H, T, B, "bi" if bi else "uni")), **d)


def g4():
    out = {}
    g = torch.Generator().manual_seed(4)
    for T in (25, 75, 6, 1):
        x = torch.randn(3, T, 10, generator=g)
        out["x_T%d" % T] = npd(x)
        for method in ("none", "avg", "max"):
            for factor in (1, 2, 3):
                y = ref_models.Downsample(method=method, factor=factor, axis=1)(x)
                out["y_T%d_%s_%d" % (T, method, factor)] = npd(y)
    # dropout: nn.Dropout(0.5) in train mode under a seed; store output of ones -> mask*2
    for seed in (11, 12):
        torch.manual_seed(seed)
        d = torch.nn.Dropout(0.5)
        y = d(torch.ones(4, 9, 16))
        out["dropout_seed%d" % seed] = npd(y)
    np.savez_compressed(os.path.join(OUT, "g4_downsample.npz"), **out)


def tiny_cfg(**kw):
    c = base_cfg(cnn_N_filt=[8, 6, 6], cnn_len_filt=[41, 5, 3], cnn_stride=[10, 1, 1],
                 cnn_max_pool_len=[2, 1, 1],
                 phone_rnn_num_hidden=[16, 16], word_rnn_num_hidden=[16, 16],
                 intent_rnn_num_hidden=[16], vocabulary_size=50, num_phonemes=11,
                 values_per_slot=[3, 4, 2])
    c.Sy_intent = {"action": {str(i): i for i in range(3)},
                   "object": {str(i): i for i in range(4)},
                   "location": {str(i): i for i in range(2)}}
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def run_model_case(model, x, y, train, seed):
    model.zero_grad()
    if train:
        model.train()
        torch.manual_seed(seed)
    else:
        model.eval()
    loss, acc = model(x, y)
    loss.backward()
    d = {"loss": npd(loss), "acc": npd(acc)}
    for k, p in model.named_parameters():
        if p.grad is not None:
            d["grad." + k] = npd(p.grad)
    return d


def g5():
    torch.manual_seed(55)
    cfg = tiny_cfg()
    model = ref_models.Model(cfg)
    x = 0.1 * torch.randn(3, 1000)
    y = torch.stack([torch.randint(0, n, (3,)) for n in cfg.values_per_slot], dim=1)
    out = {"x": npd(x), "y": npd(y)}
    for k, v in model.state_dict().items():
        out["sd." + k] = npd(v)
    model.eval()
    logits, pred = model.predict_intents(x)
    out["eval.logits"] = npd(logits)
    out["eval.pred"] = npd(pred)
    feats = model.pretrained_model.compute_features(x)
    out["eval.features"] = npd(feats)
    for k, v in run_model_case(model, x, y, train=False, seed=0).items():
        out["eval." + k] = v
    for k, v in run_model_case(model, x, y, train=True, seed=77).items():
        out["train77." + k] = v
    np.savez_compressed(os.path.join(OUT, "g5_tiny_model.npz"), **out)

    # ASR heads (PretrainedModel.forward), pretraining_type 2 and 1
    torch.manual_seed(56)
    cfg = tiny_cfg(pretraining_type=2)
    pm = ref_models.PretrainedModel(cfg)
    x = 0.1 * torch.randn(3, 1000)
    st = None
    pm.eval()
    ph, wd = pm.compute_posteriors(x)
    Tp, Tw = ph.shape[1], wd.shape[1]
    yp = torch.randint(0, cfg.num_phonemes, (3, Tp))
    yw = torch.randint(0, cfg.vocabulary_size, (3, Tw))
    yp[0, -2:] = -1
    yw[1, -1] = -1
    out = {"x": npd(x), "y_phoneme": npd(yp), "y_word": npd(yw),
           "posteriors.phoneme": npd(ph), "posteriors.word": npd(wd)}
    for k, v in pm.state_dict().items():
        out["sd." + k] = npd(v)
    for ptype in (2, 1):
        pm.pretraining_type = ptype
        pm.zero_grad()
        pl, wl, pa, wa = pm(x, yp, yw)
        loss = pl + wl if ptype == 2 else pl
        loss.backward()
        tag = "pt%d." % ptype
        out[tag + "phoneme_loss"], out[tag + "word_loss"] = npd(pl), npd(wl)
        out[tag + "phoneme_acc"], out[tag + "word_acc"] = npd(pa), npd(wa)
        for k, p in pm.named_parameters():
            if p.grad is not None:
                out[tag + "grad." + k] = npd(p.grad)
    np.savez_compressed(os.path.join(OUT, "g5_tiny_asr.npz"), **out)


def seq2seq_cfg(**kw):
    labels = ["<sos>"] + list("abcdefgh {}:'") + ["<eos>"]
    c = tiny_cfg(seq2seq=True, intent_encoder_dim=12, num_intent_encoder_layers=1, intent_decoder_dim=20,
                 num_intent_decoder_layers=2, intent_decoder_key_dim=10, intent_decoder_value_dim=14)
    c.Sy_intent = labels
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def g7():
    """seq2seq head (reference models.py:381-651, :720-725, :825-828, :848-851, :866-874): tiny Model with
    config.seq2seq = True.  Teacher-forced loss / per-utterance log-probabilities and every gradient in eval mode and
    in train mode (dropout masks by torch seed), the beam search of predict_intents (width 4, 200 steps) and the
    decoded strings; a two-encoder-layer / three-decoder-layer variant for the loss."""
    for tag, kw in (("a", {}), ("b", {"num_intent_encoder_layers": 2, "num_intent_decoder_layers": 3})):
        torch.manual_seed(70 + len(kw))
        cfg = seq2seq_cfg(**kw)
        model = ref_models.Model(cfg)
        V = len(cfg.Sy_intent)
        B, U = 3, 7
        x = 0.1 * torch.randn(B, 1800)
        idx = torch.randint(1, V - 1, (B, U))
        idx[:, 0] = 0
        idx[0, 5:] = V - 1
        idx[2, 6:] = V - 1
        y = ref_data.one_hot(idx, V)
        out = {"x": npd(x), "y_idx": npd(idx), "labels_json": np.frombuffer(json.dumps(cfg.Sy_intent).encode(), dtype=np.uint8)}
        for k, v in model.state_dict().items():
            out["sd." + k] = npd(v)
        out["sd_keys_json"] = np.frombuffer(json.dumps(list(model.state_dict().keys())).encode(), dtype=np.uint8)
        for mode, seed in (("eval", 0), ("train91", 91)):
            model.zero_grad()
            if mode == "eval":
                model.eval()
            else:
                model.train()
                torch.manual_seed(seed)
            loss, acc = model(x, y)
            loss.backward()
            out[mode + ".loss"], out[mode + ".acc"] = npd(loss), npd(acc)
            for k, p_ in model.named_parameters():
                if p_.grad is not None:
                    out[mode + ".grad." + k] = npd(p_.grad)
        model.eval()
        with torch.no_grad():
            enc = model.encoder(model.pretrained_model.compute_features(x))
            out["eval.encoder_out"] = npd(enc)
            out["eval.log_p"] = npd(model.decoder(enc, y))
            if tag == "a":
                scores, beam = model.predict_intents(x)
                out["beam.scores"] = npd(scores)
                out["beam.idx"] = npd(beam.max(dim=3)[1]).astype(np.int16)      # (4, B, 200) label indices
                out["beam.strings_json"] = np.frombuffer(json.dumps(model.decode_intents(x)).encode(), dtype=np.uint8)
                out["truth_strings_json"] = np.frombuffer(
                    json.dumps([model.one_hot_to_string(y[i], cfg.Sy_intent) for i in range(B)]).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "g7_seq2seq_%s.npz" % tag), **out)


CFG_NO_UNFREEZING = "no_unfreezing.cfg"


def g6():
    """BASELINE.json configs[0]: experiments/no_unfreezing.cfg on the CPU reference path,
    16 synthetic 1 s waveforms, one forward+backward (+Adam) step."""
    work = tempfile.mkdtemp()
    cwd = os.getcwd()
    try:
        os.chdir(work)
        os.mkdir("experiments")
        shutil.copy(os.path.join(REF, "experiments", CFG_NO_UNFREEZING), "experiments/")
        cfg = ref_data.read_config("experiments/" + CFG_NO_UNFREEZING)
        cfg.values_per_slot = [6, 14, 4]
        cfg.Sy_intent = base_cfg().Sy_intent
        cfg.num_phonemes = 42
        # synthetic pre-training checkpoint (the published .pth files are absent): seed 4321
        torch.manual_seed(4321)
        pre = ref_models.PretrainedModel(cfg)
        torch.save(pre.state_dict(), os.path.join(cfg.folder, "pretraining", "model_state.pth"))
        pre_sha = {k: sha(v) for k, v in pre.state_dict().items()}
        torch.manual_seed(cfg.seed)
        model = ref_models.Model(cfg)
        out = {}
        meta = {"pretrain_seed": 4321, "model_seed": cfg.seed,
                "pretrained_sha256": pre_sha,
                "model_sha256": {k: sha(v) for k, v in model.state_dict().items()},
                "dtypes": {k: str(v.dtype) for k, v in model.state_dict().items()},
                "shapes": {k: list(v.shape) for k, v in model.state_dict().items()}}
        g = torch.Generator().manual_seed(1234)
        x = 0.1 * torch.randn(16, 16000, generator=g)
        y = torch.stack([torch.randint(0, n, (16,), generator=g) for n in cfg.values_per_slot], dim=1)
        out["x_sha256"] = np.frombuffer(bytes.fromhex(sha(x)), dtype=np.uint8)
        out["y"] = npd(y)
        model.eval()
        logits, pred = model.predict_intents(x)
        out["eval.logits"] = npd(logits)
        out["eval.pred"] = npd(pred)
        out["eval.features_digest"] = digest(model.pretrained_model.compute_features(x))
        loss, acc = model(x, y)
        out["eval.loss"], out["eval.acc"] = npd(loss), npd(acc)
        # train-mode forward/backward, dropout seeded
        model.train()
        model.zero_grad()
        torch.manual_seed(999)
        loss, acc = model(x, y)
        loss.backward()
        out["train999.loss"], out["train999.acc"] = npd(loss), npd(acc)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["train999.graddigest." + k] = digest(p.grad)
        # everything unfrozen: same batch, same dropout seed
        for p in model.parameters():
            p.requires_grad = True
        model.zero_grad()
        torch.manual_seed(999)
        loss, acc = model(x, y)
        loss.backward()
        out["unfrozen999.loss"] = npd(loss)
        for k, p in model.named_parameters():
            if p.grad is not None:
                out["unfrozen999.graddigest." + k] = digest(p.grad)
        model.freeze_all_layers()
        model.zero_grad()
        # one Trainer.train step over a one-batch "dataset"
        class DS:
            loader = [(x, y)]
        trainer = ref_training.Trainer(model=model, config=cfg)
        buf = io.StringIO()
        stdout = sys.stdout
        sys.stdout = buf
        torch.manual_seed(2024)
        try:
            tr_acc, tr_loss = trainer.train(DS())
        finally:
            sys.stdout = stdout
        meta["print_frozen"] = [l for l in buf.getvalue().splitlines() if ": " in l and "intent" not in l]
        out["trainer.acc"], out["trainer.loss"] = np.float64(tr_acc), np.float64(tr_loss)
        for k, v in model.state_dict().items():
            if k.startswith("intent_layers"):
                out["trainer.postadam_digest." + k] = digest(v)
        with open(os.path.join(cfg.folder, "training", "log.csv")) as f:
            meta["log_csv"] = f.read()
        out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(os.path.join(OUT, "g6_full_model.npz"), **out)
    finally:
        os.chdir(cwd)
        shutil.rmtree(work)


def g8():
    work = tempfile.mkdtemp()
    cwd = os.getcwd()
    res = {}
    try:
        os.chdir(work)
        os.mkdir("experiments")
        for name in sorted(os.listdir(os.path.join(REF, "experiments"))):
            if not name.endswith(".cfg"):
                continue
            with open(os.path.join(REF, "experiments", name)) as f:
                text = f.read()
            shutil.copy(os.path.join(REF, "experiments", name), "experiments/")
            entry = {"text": text}
            buf = io.StringIO()
            stdout = sys.stdout
            sys.stdout = buf
            try:
                cfg = ref_data.read_config("experiments/" + name)
                entry["expected"] = {k: v for k, v in vars(cfg).items()}
            except Exception as e:                              # noqa: BLE001
                entry["error_type"] = type(e).__name__
                entry["error"] = str(e)
            finally:
                sys.stdout = stdout
            entry["stdout"] = buf.getvalue()
            res[name] = entry
    finally:
        os.chdir(cwd)
        shutil.rmtree(work)
    with open(os.path.join(OUT, "g8_config.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def g9():
    """Freezing schedule: names of unfrozen parametrised encoder layers after each
    Model.unfreeze_one_layer() call, for unfreezing_type 0/1/2 (pretraining_type 2) and the
    starting index of pretraining_type 1."""
    work = tempfile.mkdtemp()
    res = {}
    try:
        os.mkdir(os.path.join(work, "pretraining"))
        for ptype, utype in [(2, 0), (2, 1), (2, 2), (1, 1), (1, 2)]:
            cfg = tiny_cfg(pretraining_type=ptype, unfreezing_type=utype, folder=work)
            cfg.starting_unfreezing_index = {1: 1 + len(cfg.word_rnn_num_hidden), 2: 1}[ptype]
            torch.save(ref_models.PretrainedModel(cfg).state_dict(), os.path.join(work, "pretraining", "model_state.pth"))
            model = ref_models.Model(cfg)
            seq = []
            for _ in range(9):
                model.unfreeze_one_layer()
                names = []
                for layer in list(model.pretrained_model.phoneme_layers) + list(model.pretrained_model.word_layers):
                    if ref_models.has_params(layer) and not ref_models.is_frozen(layer):
                        names.append(layer.name)
                seq.append(names)
            res["ptype%d_utype%d" % (ptype, utype)] = seq
    finally:
        shutil.rmtree(work)
    with open(os.path.join(OUT, "g9_unfreeze.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def g10():
    """Real-data SLU input pipeline (reference data.py:132-391) on the tiny FSC-shaped tree of
    tests/slu_data_fixture.py: split/subset/wording logic and label dictionaries of get_SLU_datasets,
    SLUDataset.__len__/__getitem__ and CollateWavsSLU.  torchaudio/sox is not installed: for this fixture
    only, the sox effects chain the reference builds in __getitem__ (data.py:273-292, no effects with
    augment=False: a plain decode) is replaced by a PCM16 -> float32 / 32768 wav read, so the fixture
    pins path/label/padding logic and the sox sample convention, not sox's decoder."""
    sys.path.insert(0, os.path.dirname(OUT))
    import slu_data_fixture as fx
    from scipy.io import wavfile

    class _Chain:
        def set_input_file(self, path):
            self.path = path

        def append_effect_to_chain(self, *a):
            raise AssertionError("augmentation is disabled in the reference")

        def sox_build_flow_effects(self):
            fs, pcm = wavfile.read(self.path)
            return torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(0), fs

    sys.modules["torchaudio"].sox_effects = types.SimpleNamespace(SoxEffectsChain=_Chain)
    res = {}
    for name, (over, np_seed, tree_kw) in fx.VARIANTS.items():
        root = tempfile.mkdtemp()
        try:
            fx.make_fsc_tree(root, seed=11, **tree_kw)
            cfg = tiny_cfg(folder=root)
            cfg.slu_path = root
            cfg.seq2seq = False
            cfg.training_batch_size = 4
            cfg.real_speaker_subset_percentage = 1.0
            cfg.synthetic_speaker_subset_percentage = 1.0
            cfg.real_dataset_subset_percentage = 1.0
            cfg.synthetic_dataset_subset_percentage = 1.0
            cfg.train_wording_path = None
            cfg.test_wording_path = None
            cfg.dataset_upsample_factor = 1
            for k, v in over.items():
                setattr(cfg, k, os.path.join(root, v) if k.endswith("_path") else v)
            np.random.seed(np_seed)
            buf = io.StringIO()
            stdout = sys.stdout
            sys.stdout = buf
            try:
                tr, va, te = ref_data.get_SLU_datasets(cfg)
            finally:
                sys.stdout = stdout
            entry = {"stdout": buf.getvalue(), "values_per_slot": cfg.values_per_slot, "Sy_intent": cfg.Sy_intent,
                     "len": [len(tr), len(va), len(te)]}
            for tag, ds in (("train", tr), ("valid", va), ("test", te)):
                entry[tag + "_paths"] = [str(v) for v in ds.df.path.tolist()]
                entry[tag + "_index"] = [int(v) for v in ds.df.index.tolist()]
            # items: waveform digest + labels for a few indices (incl. one beyond len(df) when upsampled)
            if name not in ("subsets", "real_subset"):   # items by index need the gap-free index of the other variants
                idxs = [0, 1, len(tr.df) - 1] + ([len(tr.df) + 2] if len(tr) > len(tr.df) else [])
                items = []
                for i in idxs:
                    x, y = tr[i]
                    items.append({"idx": i, "n": int(len(x)), "sum": float(np.float64(x).sum()),
                                  "first": [float(v) for v in x[:4]], "y": [int(v) for v in y], "dtype": str(x.dtype)})
                entry["items"] = items
            res[name] = entry
        finally:
            shutil.rmtree(root)
    # seq2seq variant (reference data.py:143-146, 186-187, 201-208, 318-326): the output alphabet is built from a Python
    # set, i.e. its ORDER changes from process to process; the fixture pins the set, the <sos>/<eos> positions and the
    # label sequences as strings
    root = tempfile.mkdtemp()
    try:
        fx.make_fsc_tree(root, seed=11, seq2seq=True)
        cfg = tiny_cfg(folder=root)
        cfg.slu_path = root
        cfg.seq2seq = True
        cfg.training_batch_size = 4
        for k in ("real_speaker_subset_percentage", "synthetic_speaker_subset_percentage",
                  "real_dataset_subset_percentage", "synthetic_dataset_subset_percentage"):
            setattr(cfg, k, 1.0)
        cfg.train_wording_path = cfg.test_wording_path = None
        cfg.dataset_upsample_factor = 1
        np.random.seed(0)
        stdout, sys.stdout = sys.stdout, io.StringIO()
        try:
            tr, va, te = ref_data.get_SLU_datasets(cfg)
        finally:
            sys.stdout = stdout
        Sy = cfg.Sy_intent
        entry = {"alphabet_sorted": sorted(Sy[1:-1]), "first": Sy[0], "last": Sy[-1], "len": [len(tr), len(va), len(te)],
                 "train_paths": [str(v) for v in tr.df.path.tolist()], "items": []}
        for i in (0, 5, len(tr.df) - 1):
            x, y = tr[i]
            entry["items"].append({"idx": i, "n": int(len(x)), "labels": [Sy[k] for k in y]})
        res["seq2seq"] = entry
    finally:
        shutil.rmtree(root)
    labels = ["<sos>"] + list("abc{}' :") + ["<eos>"]
    coll = ref_data.CollateWavsSLU(labels, True)
    rs = np.random.RandomState(6)
    batch = [(rs.randn(n).astype(np.float32), [0] + [int(rs.randint(1, 9)) for _ in range(u)] + [9])
             for n, u in ((7, 3), (12, 6), (3, 1))]
    x, y = coll(batch)
    res["collate_seq2seq"] = {"labels": labels, "lens": [7, 12, 3], "ulens": [3, 6, 1], "seed": 6, "x": npd(x).tolist(),
                              "y_idx": npd(y.max(dim=2)[1]).tolist(), "y_shape": list(y.shape), "y_sum": float(y.sum()),
                              "y_dtype": str(y.dtype)}
    # collate: ragged float32 waveforms -> zero-padded batch
    coll = ref_data.CollateWavsSLU({"action": {}, "object": {}, "location": {}}, False)
    rs = np.random.RandomState(5)
    batch = [(rs.randn(n).astype(np.float32), [int(rs.randint(6)), int(rs.randint(14)), int(rs.randint(4))])
             for n in (7, 12, 3, 12, 9)]
    x, y = coll(batch)
    res["collate"] = {"lens": [7, 12, 3, 12, 9], "seed": 5, "x": npd(x).tolist(), "y": npd(y).tolist(),
                      "x_dtype": str(x.dtype), "y_dtype": str(y.dtype)}
    with open(os.path.join(OUT, "g10_slu_data.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def g11():
    """ASR pre-training input pipeline (reference data.py:393-545) on the LibriSpeech-shaped tree of
    tests/slu_data_fixture.py: vocabulary construction, ASRDataset.__getitem__ (label tracks, random
    snippet with torch's RNG, strided labels) and CollateWavsASR.  `soundfile` and `textgrid` are not
    installed: for this fixture only they are replaced by a PCM16 -> float64 / 32768 wav read and by a
    small regex TextGrid reader written here (independent of the package's parser), so the fixture pins the
    reference's label / cropping / padding logic, not those libraries."""
    import re
    sys.path.insert(0, os.path.dirname(OUT))
    import slu_data_fixture as fx
    from scipy.io import wavfile

    def sf_read(path):
        fs, pcm = wavfile.read(path)
        return pcm.astype(np.float64) / 32768.0, fs

    class _Interval:
        def __init__(self, a, b, m):
            self.minTime, self.maxTime, self.mark = a, b, m

    class _TextGrid:
        def read(self, path):
            text = open(path).read()
            self.tiers = {}
            for block in re.split(r"item \[\d+\]:", text)[1:]:
                name = re.search(r'name = "(.*?)"', block).group(1)
                ivs = re.findall(r'intervals \[\d+\]:\s*xmin = (\S+)\s*xmax = (\S+)\s*text = "(.*?)"', block)
                self.tiers.setdefault(name, []).append([_Interval(float(a), float(b), m) for a, b, m in ivs])

        def getList(self, name):
            return self.tiers[name]

    sys.modules["soundfile"].read = sf_read
    sys.modules["textgrid"].TextGrid = _TextGrid
    ref_data.sf.read = sf_read
    ref_data.textgrid.TextGrid = _TextGrid
    root = tempfile.mkdtemp()
    res = {}
    try:
        base = fx.make_asr_tree(root, seed=5)
        cfg = tiny_cfg(folder=os.path.join(root, "exp"))
        os.makedirs(os.path.join(cfg.folder, "pretraining"))
        cfg.asr_path = base
        cfg.vocabulary_size = 5
        cfg.pretraining_batch_size = 3
        cfg.pretraining_length_mean = 1.0
        cfg.pretraining_length_var = 0.4
        cfg.phone_downsample_factor = 40
        cfg.word_downsample_factor = 160
        rel = lambda p: os.path.relpath(p, base)

        def call():
            buf = io.StringIO()
            stdout = sys.stdout
            sys.stdout = buf
            try:
                out = ref_data.get_ASR_datasets(cfg)
            finally:
                sys.stdout = stdout
            return out, buf.getvalue()

        (tr, va, te), out1 = call()
        res["stdout_first"] = out1
        res["num_phonemes"] = cfg.num_phonemes
        res["Sy_phoneme_sorted"] = sorted(tr.Sy_phoneme)        # first-seen order depends on glob order
        res["Sy_word_set"] = sorted(tr.Sy_word)
        res["len"] = [len(tr), len(va), len(te)]
        for tag, ds in (("train", tr), ("valid", va), ("test", te)):
            res[tag + "_wavs"] = sorted(rel(p) for p in ds.wav_paths)
        # fix the vocabularies (independent of directory listing order) and re-read them
        with open(os.path.join(cfg.folder, "pretraining", "phonemes.txt"), "w") as f:
            f.write("\n".join(sorted(tr.Sy_phoneme)) + "\n")
        with open(os.path.join(cfg.folder, "pretraining", "words.txt"), "w") as f:
            f.write("\n".join(sorted(tr.Sy_word)) + "\n")
        (tr, va, te), out2 = call()
        res["stdout_second"] = out2
        res["Sy_phoneme"] = tr.Sy_phoneme
        res["Sy_word"] = tr.Sy_word
        items = {}
        for i in range(len(tr)):
            torch.manual_seed(100 + len(rel(tr.wav_paths[i])) + i * 0)
            key = rel(tr.wav_paths[i])
            torch.manual_seed(int(hashlib.sha256(key.encode()).hexdigest()[:6], 16))
            x, yp, yw = tr[i]
            items[key] = {"n": int(len(x)), "dtype": str(np.asarray(x).dtype), "sum": float(np.sum(x)),
                          "first": [float(v) for v in x[:3]], "y_phoneme": [int(v) for v in yp], "y_word": [int(v) for v in yw]}
        res["items"] = items
    finally:
        shutil.rmtree(root)
    rs = np.random.RandomState(8)
    batch = [(rs.randn(n), [int(v) for v in rs.randint(-1, 9, size=-(-n // 4))], [int(v) for v in rs.randint(-1, 5, size=-(-n // 16))])
             for n in (33, 50, 17)]
    x, yp, yw = ref_data.CollateWavsASR()(batch)
    res["collate"] = {"lens": [33, 50, 17], "seed": 8, "x": npd(x).tolist(), "yp": npd(yp).tolist(), "yw": npd(yw).tolist(),
                      "dtypes": [str(x.dtype), str(yp.dtype), str(yw.dtype)]}
    with open(os.path.join(OUT, "g11_asr_data.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    only = os.environ.get("GOLDEN_ONLY")
    fns = {"g1": g1, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g7": g7, "g8": g8, "g9": g9, "g10": g10, "g11": g11}
    for k, fn in fns.items():
        if only is None or k in only.split(","):
            fn()
    for fn in sorted(os.listdir(OUT)):
        print("%9d  %s" % (os.path.getsize(os.path.join(OUT, fn)), fn))
