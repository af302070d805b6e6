"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (torch-CPU fp32 ops, differentiable, so torch autograd yields the gradient
oracle too) of the reference's SincNet-conv + stacked-biGRU speech-encoder hot path and the
SLU/ASR loss heads.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module.  The shipped code under `end-to-end-slu_amd/` never does: it calls
the HIP kernels through the C-ABI (`include/slu_hip.h`) and fails loudly when they are missing.

Pinning: the reference (lorenlugosch/end-to-end-SLU) ships no tests, golden vectors or
known-answer fixtures for this path and delegates its arithmetic to an
unpinned third-party dependency, PyTorch (`F.conv1d`, `nn.GRU`, pooling, `cross_entropy`).
This restatement is therefore pinned against outputs of the reference ITSELF, imported
unmodified in the authoring container (torch 2.10.0 CPU): `tests/golden/make_goldens.py`
generated `tests/golden/*.npz|json`, and `tests/test_oracle_vs_golden.py` checks every function
here against them.

Every function cites the reference file:line it follows (paths relative to the reference root).
The layout conventions are the reference's: waveforms (B, T); CNN activations NCL (B, C, L);
RNN activations NLC (B, T, C).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# Sinc front end
# --------------------------------------------------------------------------------------------


def sinc_mel_init(N_filt, fs):
    """models.py:56-68 — mel-spaced initial values of the two float64 SincLayer parameters.

    Returns (filt_b1, filt_band) as float64 numpy arrays of shape (N_filt,).
    """
    low_freq_mel = 80
    high_freq_mel = 2595 * np.log10(1 + (fs / 2) / 700)
    mel_points = np.linspace(low_freq_mel, high_freq_mel, N_filt)
    f_cos = 700 * (10 ** (mel_points / 2595) - 1)
    b1 = np.roll(f_cos, 1)
    b2 = np.roll(f_cos, -1)
    b1[0] = 30
    b2[-1] = (fs / 2) - 100
    freq_scale = fs * 1.0
    return b1 / freq_scale, (b2 - b1) / freq_scale


# Working precision of the restatement.  float32 = the reference's own arithmetic (every cast point below is the
# reference's; this is the mode every golden fixture pins).  float64_evaluation() switches the same formulas to float64
# WITHOUT the intermediate float32 roundings: the rounding-free value of the function the reference computes, which the
# gradient-parity tests use as the arbiter between two fp32 evaluations (|gpu - f64| against |oracle_fp32 - f64|).
WORK_DTYPE = torch.float32


class float64_evaluation:
    """with O.float64_evaluation(): ... — evaluate the oracle in float64 (pass float64 weights and inputs: to_float64)."""

    def __enter__(self):
        global WORK_DTYPE
        self._old, WORK_DTYPE = WORK_DTYPE, torch.float64
        return self

    def __exit__(self, *exc):
        global WORK_DTYPE
        WORK_DTYPE = self._old
        return False


def to_float64(sd, requires_grad=True):
    """A float64 copy of a state_dict (floating tensors only are converted), as fresh leaves."""
    return {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()).requires_grad_(requires_grad and v.is_floating_point())
            for k, v in sd.items()}


def sinc_filters(filt_b1, filt_band, Filt_dim, fs):
    """models.py:79-106 (+ flip/sinc models.py:7-24), vectorised over the 80 filters.

    filt_b1, filt_band: float64 tensors (N_filt,) (may require grad).
    Returns the (N_filt, Filt_dim) float32 filterbank: band-pass = difference of two
    windowed-sinc low-passes, normalised by its own max, times a Hamming window.
    The operation order and the float64 -> float32 cast points are the reference's.
    """
    N = Filt_dim
    freq_scale = fs * 1.0
    half = int((N - 1) / 2)
    t_right = torch.linspace(1, (N - 1) / 2, steps=half, dtype=WORK_DTYPE) / fs   # :82 (float32)
    min_freq = 50.0
    min_band = 50.0
    beg = torch.abs(filt_b1) + min_freq / freq_scale                      # :88 (float64)
    end = beg + (torch.abs(filt_band) + min_band / freq_scale)            # :89 (float64)
    n = torch.linspace(0, N, steps=N, dtype=WORK_DTYPE)                   # :91
    window = (0.54 - 0.46 * torch.cos(2 * math.pi * n / N)).to(WORK_DTYPE)   # :94-95 (.float())

    def low_pass(f32_freq):                                               # :99-100 with sinc() :17-24
        band = (f32_freq * freq_scale).unsqueeze(1)                       # (N_filt,1) float32
        arg = 2 * math.pi * band * t_right.unsqueeze(0)                   # :18
        y_right = torch.sin(arg) / arg
        y_left = torch.flip(y_right, dims=[1])                            # :19 (flip() :7-14)
        ones = torch.ones(y_right.shape[0], 1, dtype=WORK_DTYPE)
        y = torch.cat([y_left, ones, y_right], dim=1)                     # :22
        return 2 * f32_freq.unsqueeze(1) * y

    band_pass = low_pass(end.to(WORK_DTYPE)) - low_pass(beg.to(WORK_DTYPE))   # :99-101 (.float())
    band_pass = band_pass / band_pass.max(dim=1, keepdim=True)[0]         # :103
    return band_pass * window.unsqueeze(0)                                # :106


def sinc_layer(x_ncl, filt_b1, filt_band, Filt_dim, fs, stride, padding, faithful_loop=False):
    """models.py:77-110 SincLayer.forward.  x_ncl: (B,1,T) -> (B,N_filt,L).

    The reference calls conv1d inside its 80-iteration filter loop (models.py:98-108) and keeps
    the last result; the value equals ONE convolution with the finished filterbank.  With
    faithful_loop=True the 80 redundant convolutions are executed too (used only to time the
    reference-faithful CPU baseline).
    """
    filters = sinc_filters(filt_b1, filt_band, Filt_dim, fs)
    w = filters.view(filters.shape[0], 1, Filt_dim)
    if faithful_loop:
        for _ in range(filters.shape[0] - 1):
            F.conv1d(x_ncl, w, stride=stride, padding=padding)
    return F.conv1d(x_ncl, w, stride=stride, padding=padding)             # :108


def max_pool_ceil(x_ncl, k):
    """models.py:205 — MaxPool1d(k, ceil_mode=True) along L."""
    if k == 1:
        return x_ncl
    return F.max_pool1d(x_ncl, kernel_size=k, ceil_mode=True)


def activation(x, name):
    """models.py:210-213 — LeakyReLU(0.2) when cfg says "leaky_relu", else ReLU."""
    if name == "leaky_relu":
        return F.leaky_relu(x, 0.2)
    return F.relu(x)


def dropout_with_mask(x, p, mask):
    """torch.nn.Dropout(p) in train mode with an explicit keep-mask (models.py:218,246,276,700).

    mask: None (eval mode or p == 0: identity) or a {0,1} tensor of x's shape, as produced by
    `torch.empty_like(x).bernoulli_(1 - p)`; output = x * mask / (1 - p).
    """
    if mask is None or p == 0.0:
        return x
    return x * mask * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------------------------
# GRU (the algorithm torch.nn.GRU implements; models.py:232,262,686)
# --------------------------------------------------------------------------------------------


def gru_direction_explicit(x_btI, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of a 1-layer GRU, h0 = 0, gate order [r; z; n] (models.py:232 via nn.GRU).

        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
        z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
        n = tanh   (W_in x + b_in + r * (W_hn h + b_hn))
        h' = (1 - z) * n + z * h

    x_btI: (B,T,I).  Returns (B,T,H).  Padded frames are NOT masked (the reference does not
    pack sequences).
    """
    B, T, _ = x_btI.shape
    H = w_hh.shape[1]
    gx = x_btI @ w_ih.t() + b_ih                                         # (B,T,3H)
    h = x_btI.new_zeros(B, H)
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gx[:, t, :H] + gh[:, :H])
        z = torch.sigmoid(gx[:, t, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[:, t, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        outs[t] = h
    return torch.stack(outs, dim=1)


def gru_layer(x_btI, p, bidirectional=True, explicit=True):
    """models.py:232 + RNNSelect models.py:138-149: GRU output sequence, directions concatenated.

    p: dict with weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 (+ *_reverse).
    explicit=False evaluates the same recurrence with ATen's fused CPU GRU (what the reference
    itself runs); used for CPU-baseline timing, and checked equal to the explicit loop in tests.
    """
    if not explicit:
        flat = [p["weight_ih_l0"], p["weight_hh_l0"], p["bias_ih_l0"], p["bias_hh_l0"]]
        if bidirectional:
            flat += [p["weight_ih_l0_reverse"], p["weight_hh_l0_reverse"],
                     p["bias_ih_l0_reverse"], p["bias_hh_l0_reverse"]]
        H = p["weight_hh_l0"].shape[1]
        h0 = x_btI.new_zeros(2 if bidirectional else 1, x_btI.shape[0], H)
        out, _ = torch._VF.gru(x_btI, h0, flat, True, 1, 0.0, False, bidirectional, True)
        return out
    fwd = gru_direction_explicit(x_btI, p["weight_ih_l0"], p["weight_hh_l0"],
                                 p["bias_ih_l0"], p["bias_hh_l0"], reverse=False)
    if not bidirectional:
        return fwd
    bwd = gru_direction_explicit(x_btI, p["weight_ih_l0_reverse"], p["weight_hh_l0_reverse"],
                                 p["bias_ih_l0_reverse"], p["bias_hh_l0_reverse"], reverse=True)
    return torch.cat([fwd, bwd], dim=2)


def downsample(x_btc, method, factor):
    """models.py:26-46 Downsample(axis=1).  "none": strided slice; "avg"/"max": pool1d with
    ceil_mode=True (a partial last window averages over the elements that are present)."""
    if method == "none":
        return x_btc[:, ::factor]
    if method == "avg":
        return F.avg_pool1d(x_btc.transpose(1, 2), kernel_size=factor, ceil_mode=True).transpose(1, 2)
    if method == "max":
        return F.max_pool1d(x_btc.transpose(1, 2), kernel_size=factor, ceil_mode=True).transpose(1, 2)
    raise ValueError("downsampling method must be one of: none, avg, max")   # models.py:36-38


# --------------------------------------------------------------------------------------------
# Whole-path assembly.  `sd` is a reference-format state_dict: tensors only.
# `cfg` is any object with the attributes data.read_config produces (data.py:25-128).
# --------------------------------------------------------------------------------------------


def _gru_params(sd, prefix):
    keys = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    out = {}
    for k in keys:
        out[k] = sd[prefix + k]
        if prefix + k + "_reverse" in sd:
            out[k + "_reverse"] = sd[prefix + k + "_reverse"]
    return out


def phoneme_layer_index(cfg):
    """Index of each parametrised module inside PretrainedModel.phoneme_layers (models.py:180-255):
    per conv block [conv|sinc, (abs,) pool, act, dropout], then ncl2nlc, then per rnn layer
    [gru, select, dropout, downsample]."""
    idx = {}
    i = 0
    for c in range(len(cfg.cnn_N_filt)):
        idx["conv%d" % c] = i
        i += 1
        if c == 0:
            i += 1            # abs0 (models.py:194-196)
        i += 3                # pool, act, dropout
    i += 1                    # ncl2nlc
    for r in range(len(cfg.phone_rnn_num_hidden)):
        idx["phone_rnn%d" % r] = i
        i += 4
    return idx


def encoder_stages(sd, x_bt, cfg, masks=None, prefix="", explicit_gru=True, faithful_sinc=False,
                   upto="features"):
    """PretrainedModel.compute_features (models.py:349-361) with every intermediate returned.

    masks: None (eval) or dict name -> keep-mask for "phone_dropout{i}", "word_dropout{i}".
    Returns an ordered dict of named stage outputs; ["features"] is the encoder output (B,T',C),
    ["phoneme_features"] the output of the phoneme module (input of phoneme_linear).
    """
    masks = masks or {}
    st = {}
    pidx = phoneme_layer_index(cfg)
    out = x_bt.unsqueeze(1)                                               # :354
    for c in range(len(cfg.cnn_N_filt)):
        li = pidx["conv%d" % c]
        if c == 0 and cfg.use_sincnet:
            out = sinc_layer(out, sd[prefix + "phoneme_layers.%d.filt_b1" % li],
                             sd[prefix + "phoneme_layers.%d.filt_band" % li],
                             cfg.cnn_len_filt[0], cfg.fs, cfg.cnn_stride[0],
                             cfg.cnn_len_filt[0] // 2, faithful_loop=faithful_sinc)   # :186
        else:
            out = F.conv1d(out, sd[prefix + "phoneme_layers.%d.weight" % li],
                           sd[prefix + "phoneme_layers.%d.bias" % li],
                           stride=cfg.cnn_stride[c], padding=cfg.cnn_len_filt[c] // 2)  # :190,200
        st["conv%d" % c] = out
        if c == 0:
            out = torch.abs(out)                                          # :194 (Abs, both front ends)
        out = max_pool_ceil(out, cfg.cnn_max_pool_len[c])                 # :205
        out = activation(out, cfg.cnn_act[c])                             # :210-213
        out = dropout_with_mask(out, cfg.cnn_drop[c], masks.get("dropout%d" % c))   # :218
        st["cnn%d" % c] = out
    out = out.transpose(1, 2)                                             # NCL2NLC :125-136
    for r in range(len(cfg.phone_rnn_num_hidden)):
        li = pidx["phone_rnn%d" % r]
        out = gru_layer(out, _gru_params(sd, prefix + "phoneme_layers.%d." % li),
                        cfg.phone_rnn_bidirectional, explicit_gru)         # :232
        st["phone_rnn%d" % r] = out
        out = dropout_with_mask(out, cfg.phone_rnn_drop[r], masks.get("phone_dropout%d" % r))  # :246
        out = downsample(out, cfg.phone_downsample_type[r], cfg.phone_downsample_len[r])      # :251
        st["phone_down%d" % r] = out
    st["phoneme_features"] = out
    if upto == "phoneme_features":
        return st
    for r in range(len(cfg.word_rnn_num_hidden)):
        out = gru_layer(out, _gru_params(sd, prefix + "word_layers.%d." % (4 * r)),
                        cfg.word_rnn_bidirectional, explicit_gru)          # :262
        st["word_rnn%d" % r] = out
        out = dropout_with_mask(out, cfg.word_rnn_drop[r], masks.get("word_dropout%d" % r))    # :276
        out = downsample(out, cfg.word_downsample_type[r], cfg.word_downsample_len[r])        # :281
        st["word_down%d" % r] = out
    st["features"] = out
    return st


def intent_logits(sd, feats_btc, cfg, masks=None, explicit_gru=True):
    """Model.forward's intent stack (models.py:683-715, 807-809): per layer GRU -> Dropout ->
    Downsample, then Linear ("final_classifier") and FinalPool = max over time (models.py:112-123)."""
    masks = masks or {}
    out = feats_btc
    n = len(cfg.intent_rnn_num_hidden)
    for r in range(n):
        out = gru_layer(out, _gru_params(sd, "intent_layers.%d." % (4 * r)),
                        cfg.intent_rnn_bidirectional, explicit_gru)
        out = dropout_with_mask(out, cfg.intent_rnn_drop[r], masks.get("intent_dropout%d" % r))
        out = downsample(out, cfg.intent_downsample_type[r], cfg.intent_downsample_len[r])
    out = out @ sd["intent_layers.%d.weight" % (4 * n)].t() + sd["intent_layers.%d.bias" % (4 * n)]
    return out.max(dim=1)[0]


def slu_loss_acc(logits, y_intent, values_per_slot):
    """models.py:811-821: sum over slots of mean cross-entropy; accuracy = all slots correct."""
    loss = 0.0
    start = 0
    pred = []
    for slot, nv in enumerate(values_per_slot):
        sub = logits[:, start:start + nv]
        loss = loss + F.cross_entropy(sub, y_intent[:, slot])
        pred.append(sub.max(1)[1])
        start += nv
    pred = torch.stack(pred, dim=1)
    acc = (pred == y_intent).prod(1).float().mean()
    return loss, acc, pred


def slu_forward(sd, x_bt, y_intent, cfg, masks=None, explicit_gru=True, faithful_sinc=False):
    """Model.forward (models.py:797-823), non-seq2seq branch.  Returns (loss, acc, logits, pred)."""
    st = encoder_stages(sd, x_bt, cfg, masks, prefix="pretrained_model.",
                        explicit_gru=explicit_gru, faithful_sinc=faithful_sinc)
    logits = intent_logits(sd, st["features"], cfg, masks, explicit_gru)
    loss, acc, pred = slu_loss_acc(logits, y_intent, cfg.values_per_slot)
    return loss, acc, logits, pred


def asr_forward(sd, x_bt, y_phoneme, y_word, cfg, masks=None, explicit_gru=True, stages_out=None):
    """PretrainedModel.forward (models.py:291-331): phoneme/word cross-entropy with
    ignore_index=-1 and frame accuracies over the non-ignored frames.
    stages_out: optional dict that receives the encoder's named stage outputs (tests that inspect intermediates)."""
    upto = "phoneme_features" if cfg.pretraining_type == 1 else "features"
    st = encoder_stages(sd, x_bt, cfg, masks, prefix="", explicit_gru=explicit_gru, upto=upto)
    if stages_out is not None:
        stages_out.update(st)
    ph = st["phoneme_features"] @ sd["phoneme_linear.weight"].t() + sd["phoneme_linear.bias"]
    ph = ph.reshape(ph.shape[0] * ph.shape[1], -1)
    yp = y_phoneme.reshape(-1)
    phoneme_loss = F.cross_entropy(ph, yp, ignore_index=-1)                # :312
    valid = yp != -1
    phoneme_acc = (ph.max(1)[1][valid] == yp[valid]).float().mean()        # :314
    if cfg.pretraining_type == 1:                                          # :317-319
        return phoneme_loss, torch.tensor([0.]), phoneme_acc, torch.tensor([0.])
    wd = st["features"] @ sd["word_linear.weight"].t() + sd["word_linear.bias"]
    wd = wd.reshape(wd.shape[0] * wd.shape[1], -1)
    yw = y_word.reshape(-1)
    word_loss = F.cross_entropy(wd, yw, ignore_index=-1)                   # :327
    validw = yw != -1
    word_acc = (wd.max(1)[1][validw] == yw[validw]).float().mean()         # :329
    return phoneme_loss, word_loss, phoneme_acc, word_acc


# --------------------------------------------------------------------------------------------
# Reference-format random initialisation (what PretrainedModel.__init__/Model.__init__ draw from
# the torch RNG, in the reference's order — models.py:174-289, 657-728).  Used by tests/bench to
# build full-size weights on the GPU box, where the reference itself is not available.
# --------------------------------------------------------------------------------------------


def _conv_init(cout, cin, k):
    m = torch.nn.Conv1d(cin, cout, k)
    return m.weight.detach().clone(), m.bias.detach().clone()


def _gru_init(I, H, bidirectional):
    m = torch.nn.GRU(input_size=I, hidden_size=H, batch_first=True, bidirectional=bidirectional)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def _linear_init(I, O):
    m = torch.nn.Linear(I, O)
    return m.weight.detach().clone(), m.bias.detach().clone()


def init_pretrained_state_dict(cfg):
    """A state_dict with the keys/shapes/dtypes and the RNG-consumption order of
    PretrainedModel(config) (models.py:174-289): conv1.., phone GRUs, phoneme_linear,
    word GRUs, word_linear.  SincLayer draws nothing (deterministic mel init, float64)."""
    sd = {}
    pidx = phoneme_layer_index(cfg)
    for c in range(len(cfg.cnn_N_filt)):
        li = pidx["conv%d" % c]
        if c == 0 and cfg.use_sincnet:
            b1, band = sinc_mel_init(cfg.cnn_N_filt[0], cfg.fs)
            sd["phoneme_layers.%d.filt_b1" % li] = torch.from_numpy(b1)
            sd["phoneme_layers.%d.filt_band" % li] = torch.from_numpy(band)
        else:
            cin = 1 if c == 0 else cfg.cnn_N_filt[c - 1]
            w, b = _conv_init(cfg.cnn_N_filt[c], cin, cfg.cnn_len_filt[c])
            sd["phoneme_layers.%d.weight" % li] = w
            sd["phoneme_layers.%d.bias" % li] = b
    out_dim = cfg.cnn_N_filt[-1]
    for r, H in enumerate(cfg.phone_rnn_num_hidden):
        li = pidx["phone_rnn%d" % r]
        for k, v in _gru_init(out_dim, H, cfg.phone_rnn_bidirectional).items():
            sd["phoneme_layers.%d.%s" % (li, k)] = v
        out_dim = H * (2 if cfg.phone_rnn_bidirectional else 1)
    w, b = _linear_init(out_dim, cfg.num_phonemes)
    sd["phoneme_linear.weight"], sd["phoneme_linear.bias"] = w, b
    word = {}
    for r, H in enumerate(cfg.word_rnn_num_hidden):
        for k, v in _gru_init(out_dim, H, cfg.word_rnn_bidirectional).items():
            word["word_layers.%d.%s" % (4 * r, k)] = v
        out_dim = H * (2 if cfg.word_rnn_bidirectional else 1)
    sd.update(word)
    w, b = _linear_init(out_dim, cfg.vocabulary_size)
    sd["word_linear.weight"], sd["word_linear.bias"] = w, b
    return sd


def init_model_state_dict(cfg, pretrained_sd=None):
    """State dict of Model(config) (models.py:657-728): "pretrained_model."-prefixed encoder
    (freshly drawn, or `pretrained_sd` when a pre-training checkpoint is loaded, :662-667 — the
    fresh draw still consumes the RNG first, as in the reference) + the intent module."""
    fresh = init_pretrained_state_dict(cfg)
    enc = pretrained_sd if pretrained_sd is not None else fresh
    sd = {"pretrained_model." + k: v for k, v in enc.items()}
    out_dim = cfg.word_rnn_num_hidden[-1] * (2 if cfg.word_rnn_bidirectional else 1)
    n = len(cfg.intent_rnn_num_hidden)
    for r, H in enumerate(cfg.intent_rnn_num_hidden):
        for k, v in _gru_init(out_dim, H, cfg.intent_rnn_bidirectional).items():
            sd["intent_layers.%d.%s" % (4 * r, k)] = v
        out_dim = H * (2 if cfg.intent_rnn_bidirectional else 1)
    w, b = _linear_init(out_dim, sum(cfg.values_per_slot))
    sd["intent_layers.%d.weight" % (4 * n)], sd["intent_layers.%d.bias" % (4 * n)] = w, b
    return sd


class OracleConfig:
    """Attribute bag with the hyper-parameters shared by all 29 reference cfgs
    (experiments/no_unfreezing.cfg:5-39), overridable by keyword."""

    def __init__(self, **kw):
        self.use_sincnet = True
        self.fs = 16000
        self.cnn_N_filt = [80, 60, 60]
        self.cnn_len_filt = [401, 5, 5]
        self.cnn_stride = [80, 1, 1]
        self.cnn_max_pool_len = [2, 1, 1]
        self.cnn_act = ["leaky_relu"] * 3
        self.cnn_drop = [0.0, 0.0, 0.0]
        self.phone_rnn_num_hidden = [128, 128]
        self.phone_downsample_len = [2, 2]
        self.phone_downsample_type = ["avg", "avg"]
        self.phone_rnn_drop = [0.5, 0.5]
        self.phone_rnn_bidirectional = True
        self.word_rnn_num_hidden = [128, 128]
        self.word_downsample_len = [2, 2]
        self.word_downsample_type = ["avg", "avg"]
        self.word_rnn_drop = [0.5, 0.5]
        self.word_rnn_bidirectional = True
        self.vocabulary_size = 10000
        self.intent_rnn_num_hidden = [128]
        self.intent_downsample_len = [1]
        self.intent_downsample_type = ["none"]
        self.intent_rnn_drop = [0.5]
        self.intent_rnn_bidirectional = True
        self.pretraining_type = 2
        self.unfreezing_type = 0
        self.num_phonemes = 42
        self.values_per_slot = [6, 14, 4]
        self.seq2seq = False
        for k, v in kw.items():
            setattr(self, k, v)


def draw_dropout_masks(cfg, x_bt, seed, include_intent=True):
    """Keep-masks drawn exactly as torch-CPU nn.Dropout draws them in one train-mode forward of
    Model (order phone0, phone1, word0, word1, intent): each is
    `torch.empty(shape).bernoulli_(1-p)` under `torch.manual_seed(seed)`.  Shapes follow from T.
    """
    g = torch.Generator().manual_seed(seed)
    B, T = x_bt.shape
    L = T
    for c in range(len(cfg.cnn_N_filt)):
        k = cfg.cnn_len_filt[c]
        L = (L + 2 * (k // 2) - k) // cfg.cnn_stride[c] + 1
        L = -(-L // cfg.cnn_max_pool_len[c])
    masks = {}

    def one(name, T_, C, p):
        # ATen's CPU GRU with batch_first returns a (B,T,C) VIEW of a time-major (T,B,C) buffer;
        # nn.Dropout draws its mask with empty_like (same strides) in MEMORY order, so the mask
        # element for (b,t,c) is the (t,b,c)-th draw.  Pinned by tests/golden/g5, g6.
        if p > 0:
            masks[name] = torch.empty(T_, B, C).bernoulli_(1 - p, generator=g).transpose(0, 1)

    for r, H in enumerate(cfg.phone_rnn_num_hidden):
        C = H * (2 if cfg.phone_rnn_bidirectional else 1)
        one("phone_dropout%d" % r, L, C, cfg.phone_rnn_drop[r])
        L = -(-L // cfg.phone_downsample_len[r])
    for r, H in enumerate(cfg.word_rnn_num_hidden):
        C = H * (2 if cfg.word_rnn_bidirectional else 1)
        one("word_dropout%d" % r, L, C, cfg.word_rnn_drop[r])
        L = -(-L // cfg.word_downsample_len[r])
    if include_intent:
        for r, H in enumerate(cfg.intent_rnn_num_hidden):
            C = H * (2 if cfg.intent_rnn_bidirectional else 1)
            one("intent_dropout%d" % r, L, C, cfg.intent_rnn_drop[r])
            L = -(-L // cfg.intent_downsample_len[r]) if cfg.intent_downsample_type[r] != "none" \
                else len(range(0, L, cfg.intent_downsample_len[r]))
    return masks


# --------------------------------------------------------------------------------------------
# seq2seq intent head (models.py:381-651, hooks :720-725, :825-828, :848-851)
# State-dict keys as the reference's Model registers them: "encoder.layers.{3i}.*" (nn.GRU),
# "decoder.embed.*", "decoder.attention.{key,query,value}_linear.*", "decoder.rnn.layers.{2l}.*"
# (nn.GRUCell: weight_ih, weight_hh, bias_ih, bias_hh), "decoder.initial_state", "decoder.linear.*".
# --------------------------------------------------------------------------------------------


def seq2seq_encoder(sd, feats_btc, cfg, masks=None, explicit_gru=True):
    """Seq2SeqEncoder.forward (models.py:381-416): per layer biGRU -> RNNSelect -> Dropout(0.5)."""
    masks = masks or {}
    out = feats_btc
    for i in range(cfg.num_intent_encoder_layers):
        out = gru_layer(out, _gru_params(sd, "encoder.layers.%d." % (3 * i)), True, explicit_gru)    # :388
        out = dropout_with_mask(out, 0.5, masks.get("intent_encoder_dropout%d" % i))                 # :401
    return out


def attention(sd, enc_btc, dec_state_bd, key_dim):
    """Attention.forward (models.py:427-438): dot-product attention of one query over the encoder states."""
    p = "decoder.attention."
    keys = enc_btc @ sd[p + "key_linear.weight"].t() + sd[p + "key_linear.bias"]
    values = enc_btc @ sd[p + "value_linear.weight"].t() + sd[p + "value_linear.bias"]
    query = dec_state_bd @ sd[p + "query_linear.weight"].t() + sd[p + "query_linear.bias"]
    scale = torch.sqrt(torch.tensor(key_dim).float())                                                # :421
    scores = torch.matmul(keys, query.unsqueeze(2)) / scale                                          # (B,T,1)
    w = torch.softmax(scores, dim=1).transpose(1, 2)
    return torch.matmul(w, values).squeeze(1)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell (models.py:450-452): one GRU step, gate order [r; z; n]."""
    H = h.shape[1]
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def decoder_rnn(sd, inp, prev_state_bld, num_layers, step_masks=None):
    """DecoderRNN.forward (models.py:462-485): stacked GRUCells, Dropout(0.5) between them (the dropout after the
    LAST cell draws a mask too, its output is discarded).  step_masks: None or list of (B, D) keep-masks per layer."""
    state = []
    out = inp
    for l in range(num_layers):
        p = "decoder.rnn.layers.%d." % (2 * l)
        out = gru_cell(out, prev_state_bld[:, l], sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], sd[p + "bias_hh"])
        state.append(out)
        out = dropout_with_mask(out, 0.5, None if step_masks is None else step_masks[l])
    return torch.stack(state, dim=1)


def seq2seq_decoder_forward(sd, enc_btc, y_buv, cfg, masks=None, SOS=0):
    """Seq2SeqDecoder.forward (models.py:504-557): teacher-forced log p(y|x) per utterance."""
    B, U, V = y_buv.shape
    L = cfg.num_intent_decoder_layers
    state = torch.stack([sd["decoder.initial_state"]] * B)                                           # :521
    log_p = 0
    y_prev = torch.zeros(B, V)
    y_prev[:, SOS] = 1.0                                                                             # :525-526
    for u in range(U):
        ctx = attention(sd, enc_btc, state[:, -1], cfg.intent_decoder_key_dim)                       # :530
        emb = y_prev @ sd["decoder.embed.weight"].t() + sd["decoder.embed.bias"]                     # :531
        sm = None if not masks else [masks["decoder_dropout_u%d_l%d" % (u, l)] for l in range(L)]
        state = decoder_rnn(sd, torch.cat([emb, ctx], dim=1), state, L, sm)                          # :532-533
        out = torch.log_softmax(state[:, -1] @ sd["decoder.linear.weight"].t() + sd["decoder.linear.bias"], dim=1)
        log_p = log_p + (out * y_buv[:, u, :]).sum(dim=1)                                            # :537-540
        y_prev = y_buv[:, u, :]                                                                      # :543
    return log_p


def seq2seq_forward(sd, x_bt, y_buv, cfg, masks=None, explicit_gru=True, SOS=0):
    """Model.forward, seq2seq branch (models.py:825-828): -> (loss = -mean log p(y|x), log_p (B))."""
    st = encoder_stages(sd, x_bt, cfg, masks, prefix="pretrained_model.", explicit_gru=explicit_gru)
    enc = seq2seq_encoder(sd, st["features"], cfg, masks, explicit_gru)
    log_p = seq2seq_decoder_forward(sd, enc, y_buv, cfg, masks, SOS)
    return -log_p.mean(), log_p


def sort_beam(ext, scores, ptrs):
    """sort_beam (models.py:487-502): per utterance, order the candidate extensions by score, descending.
    ext (W, B, V), scores (W, B), ptrs (W, B) with W candidates."""
    order = scores.sort(dim=0, descending=True)[1]
    cols = torch.arange(scores.shape[1])
    return ext[order, cols], scores[order, cols], ptrs[order, cols]


def seq2seq_infer(sd, enc_btc, cfg, num_labels, beam=4, max_len=200):
    """Seq2SeqDecoder.infer (models.py:559-651) in eval mode: beam search of width `beam`; the first input is the
    all-zero vector (:596), only hypothesis 0 is expanded at the first step (:625).
    Returns (beam_scores (beam, B), beam (beam, B, max_len, V) one-hot)."""
    Bsz = enc_btc.shape[0]
    L = cfg.num_intent_decoder_layers
    init = torch.stack([sd["decoder.initial_state"]] * Bsz)
    hyp = torch.zeros(beam, Bsz, max_len, num_labels)
    hyp_scores = torch.zeros(beam, Bsz)
    states = torch.zeros(beam, *init.shape)
    for u in range(max_len):
        exts, ext_scores, ptrs = [], [], []
        for b in range(beam):
            if u == 0:
                state, score, y_prev = init, hyp_scores[b], torch.zeros(Bsz, num_labels)
            else:
                state, score, y_prev = states[b], hyp_scores[b], hyp[b][:, u - 1, :]
            ctx = attention(sd, enc_btc, state[:, -1], cfg.intent_decoder_key_dim)
            emb = y_prev @ sd["decoder.embed.weight"].t() + sd["decoder.embed.bias"]
            state = decoder_rnn(sd, torch.cat([emb, ctx], dim=1), state, L, None)
            states[b] = state
            out = torch.log_softmax(state[:, -1] @ sd["decoder.linear.weight"].t() + sd["decoder.linear.bias"], dim=1)
            top_s, top_i = out.topk(beam)                                                            # :612
            for e in range(beam):
                onehot = torch.zeros(Bsz, num_labels)
                onehot[torch.arange(Bsz), top_i[:, e]] = 1.0
                exts.append(onehot)
                ext_scores.append(top_s[:, e] + score)
                ptrs.append(torch.full((Bsz,), b, dtype=torch.long))
            if u == 0:
                break
        exts, ext_scores, ptrs = sort_beam(torch.stack(exts), torch.stack(ext_scores), torch.stack(ptrs))
        old_hyp, old_states = hyp.clone(), states.clone()
        cols = torch.arange(Bsz)
        for b in range(beam):                                                                        # :640-647
            hyp[b] = old_hyp[ptrs[b], cols]
            hyp[b, :, u, :] = exts[b]
            hyp_scores[b] = ext_scores[b]
            states[b] = old_states[ptrs[b], cols]
    return hyp_scores, hyp


def one_hot_to_string(onehot_uv, S):
    """Model.one_hot_to_string (models.py:731-737): arg-max labels joined, then lstrip("<sos>") / rstrip("<eos>")
    (character-SET strips, as the reference applies them)."""
    return "".join([S[c] for c in onehot_uv.max(dim=1)[1]]).lstrip("<sos>").rstrip("<eos>")


def init_seq2seq_state_dict(cfg, num_labels):
    """Parameters of the seq2seq head in the reference's construction order (models.py:720-725: Seq2SeqEncoder GRUs;
    Seq2SeqDecoder: embed, attention key / query / value, GRUCells, initial_state (randn), linear)."""
    sd = {}
    out_dim = cfg.word_rnn_num_hidden[-1] * (2 if cfg.word_rnn_bidirectional else 1)
    for i in range(cfg.num_intent_encoder_layers):
        for k, v in _gru_init(out_dim, cfg.intent_encoder_dim, True).items():
            sd["encoder.layers.%d.%s" % (3 * i, k)] = v
        out_dim = 2 * cfg.intent_encoder_dim
    Dd, Kd, Vd = cfg.intent_decoder_dim, cfg.intent_decoder_key_dim, cfg.intent_decoder_value_dim
    sd["decoder.embed.weight"], sd["decoder.embed.bias"] = _linear_init(num_labels, Dd)
    for name, (i, o) in (("key", (2 * cfg.intent_encoder_dim, Kd)), ("query", (Dd, Kd)), ("value", (2 * cfg.intent_encoder_dim, Vd))):
        w, b = _linear_init(i, o)
        sd["decoder.attention.%s_linear.weight" % name], sd["decoder.attention.%s_linear.bias" % name] = w, b
    for l in range(cfg.num_intent_decoder_layers):
        m = torch.nn.GRUCell(input_size=(Dd + Vd) if l == 0 else Dd, hidden_size=Dd)
        for k, v in m.state_dict().items():
            sd["decoder.rnn.layers.%d.%s" % (2 * l, k)] = v.detach().clone()
    sd["decoder.initial_state"] = torch.randn(cfg.num_intent_decoder_layers, Dd)
    sd["decoder.linear.weight"], sd["decoder.linear.bias"] = _linear_init(Dd, num_labels)
    return sd


def draw_seq2seq_masks(cfg, x_bt, U, seed):
    """Keep-masks of one train-mode seq2seq forward in torch-CPU's draw order: encoder stack (draw_dropout_masks
    without the intent layers), the Seq2SeqEncoder dropouts (B,T,C) in memory order (T,B,C), then per decoding
    step and decoder layer a (B, decoder_dim) mask.  Same generator stream as torch.manual_seed(seed)."""
    g = torch.Generator().manual_seed(seed)
    B, T = x_bt.shape
    L = T
    for c in range(len(cfg.cnn_N_filt)):
        k = cfg.cnn_len_filt[c]
        L = (L + 2 * (k // 2) - k) // cfg.cnn_stride[c] + 1
        L = -(-L // cfg.cnn_max_pool_len[c])
    masks = {}
    for prefix, hidden, bi, drops, lens in (("phone", cfg.phone_rnn_num_hidden, cfg.phone_rnn_bidirectional, cfg.phone_rnn_drop, cfg.phone_downsample_len),
                                            ("word", cfg.word_rnn_num_hidden, cfg.word_rnn_bidirectional, cfg.word_rnn_drop, cfg.word_downsample_len)):
        for r, H in enumerate(hidden):
            C = H * (2 if bi else 1)
            if drops[r] > 0:
                masks["%s_dropout%d" % (prefix, r)] = torch.empty(L, B, C).bernoulli_(1 - drops[r], generator=g).transpose(0, 1)
            L = -(-L // lens[r])
    for i in range(cfg.num_intent_encoder_layers):
        masks["intent_encoder_dropout%d" % i] = torch.empty(L, B, 2 * cfg.intent_encoder_dim).bernoulli_(0.5, generator=g).transpose(0, 1)
    for u in range(U):
        for l in range(cfg.num_intent_decoder_layers):
            masks["decoder_dropout_u%d_l%d" % (u, l)] = torch.empty(B, cfg.intent_decoder_dim).bernoulli_(0.5, generator=g)
    return masks
