"""MI355X-native mirror of the reference's `models.py` API surface for the speech-encoder hot path.

Same public names, constructor arguments, method signatures, `state_dict` keys/shapes/dtypes,
per-layer `.name` attributes and freezing semantics as the reference
(lorenlugosch/end-to-end-SLU `models.py`: SincLayer :49-110, Downsample :26-46, PretrainedModel
:170-361, freeze helpers :363-379, Model :653-874), so `experiments/*.cfg`, `training.Trainer`
and user scripts run unchanged — but every hot operator executes in hand-written HIP kernels for
gfx950 (libslu_hip.so, C ABI in include/slu_hip.h) instead of ATen:

  reference module chain (one ATen op each)             this file (one fused HIP stage each)
  sinc0, abs0, pool0, act0, dropout0                 -> ops.SincBlockFn   (filter build + MFMA conv
                                                         + |.| + max-pool + LeakyReLU epilogue)
  convN, poolN, actN, dropoutN                       -> ops.ConvBlockFn   (same kernel, dense taps)
  ncl2nlc                                            -> free: kernels write channels-last /
                                                         time-major directly
  *_rnnN, *_rnn_selectN, *_dropoutN, *_downsampleN   -> ops.GRULayerFn    (MFMA input projection,
                                                         persistent recurrence, dropout+pool)

The `phoneme_layers` / `word_layers` / `intent_layers` ModuleLists keep the reference's indices
(so checkpoints interchange) and still work layer-by-layer, but `compute_features`, `forward`
and `predict_intents` run the fused stage plan.  There is no CPU path: without a gfx950 device
the forward methods raise.  The seq2seq head (models.py:381-651: Seq2SeqEncoder, Attention, DecoderRNN,
Seq2SeqDecoder with beam search) is provided on the same kernels (ops.Seq2SeqDecoderFn).
"""
import contextlib
import os
import sys

import numpy as np
import torch

from slu_hip import ops as _ops
from slu_hip import lib as _lib


# ------------------------------------------------------------------------------------------------
# dropout control (train-mode masks): Philox in-kernel by default, injected masks for parity tests
# ------------------------------------------------------------------------------------------------
class _DropoutState:
    masks = None        # dict: layer name ("phone_dropout0", ...) -> float {0,1} mask, logical (B,T,C)
    seed = None         # None -> torch.initial_seed()
    step = 0            # index of the last top-level forward; Philox offset = step * 16 + dropout site
    current = 0         # step the stages running right now belong to
    current_dev = None  # or: 1-element int64 CUDA tensor holding step*16 (hipGraph-captured stages)
    sub_batch = 0       # > 0: the batch is several steps' batches concatenated (consecutive steps)


def set_dropout_masks(masks):
    """Parity hook: use these keep-masks (as `torch.nn.Dropout` would have drawn them, logical
    shape (B,T,C)) instead of the in-kernel Philox stream.  Pass None to restore Philox."""
    _DropoutState.masks = masks


def set_dropout_seed(seed):
    """Seed of the in-kernel Philox stream (per-rank streams under data parallelism)."""
    _DropoutState.seed = seed
    _DropoutState.step = _DropoutState.current = 0


def next_rng_step():
    """Reserve the dropout stream index of one top-level forward.  Every dropout site of that
    forward draws from Philox(seed, offset = step*16 + site), whichever HIP stream or order its
    stages run in — which is what lets the trainer run the frozen part of the encoder ahead of
    time on side streams and still produce exactly the sequential run's masks."""
    _DropoutState.step += 1
    return _DropoutState.step


def _dropout_args(name, site, p, training, cnn=False):
    """-> (p_eff, mask_time_major_or_None, seed, offset)"""
    if not training or p == 0.0:
        return 0.0, None, 0, 0
    if _DropoutState.masks is not None:
        m = _DropoutState.masks[name]
        if cnn:                                    # reference shape (B,C,L) -> channels-last (B,L,C)
            return p, m.transpose(1, 2).contiguous(), 0, 0
        return p, m.transpose(0, 1), 0, 0          # (T,B,C) view; kernel takes its strides
    seed = _DropoutState.seed if _DropoutState.seed is not None else torch.initial_seed()
    if _DropoutState.current_dev is not None:
        return p, None, seed & 0xFFFFFFFFFFFFFFFF, (site, _DropoutState.current_dev, _DropoutState.sub_batch)
    if _DropoutState.sub_batch:
        return p, None, seed & 0xFFFFFFFFFFFFFFFF, (_DropoutState.current * 16 + site, None, _DropoutState.sub_batch)
    return p, None, seed & 0xFFFFFFFFFFFFFFFF, _DropoutState.current * 16 + site


# 16 dropout sites per step (Philox offset = step * 16 + site).  The seq2seq head replaces the intent stack, so its encoder
# layers take the intent sites and the decoder's per-step dropouts share site 11 (element index = step, layer, b, j) — which
# is also the site a fourth encoder layer would take, so deeper stacks continue in a disjoint region of the 64-bit offset
# (below).
_SITE_BASE = {"phone": 0, "word": 4, "intent": 8, "cnn": 12, "intent_encoder": 8}
_DECODER_SITE = 11
_SITE_OVERFLOW_SHIFT = 40       # layers past a module's own sites: offset += (extra block) << 40 (steps stay below 2^36)


def _site(module, idx):
    """Dropout-site number of layer `idx` of a module: the Philox offset of a step is step*16 + site.  A module owns four
    consecutive sites (the seq2seq encoder three: site 11 is its decoder's); the reference builds as many layers as
    its cfg lists name (models.py:227-286, 683-705), so deeper layers get offsets of their own in a region no step count
    reaches: site = base + idx % n + ((idx // n) << 40).  Every layer of the shipped cfgs keeps the offset it always had."""
    if idx < 0:
        raise ValueError("negative layer index")
    n = 3 if module == "intent_encoder" else 4
    return _SITE_BASE[module] + idx % n + ((idx // n) << _SITE_OVERFLOW_SHIFT)


class _FrozenMath:
    """State of the frozen stages' arithmetic in the guarded mode (SLU_FROZEN_MATH=auto; slu_hip/guard.py)."""
    scope = None        # the RangeGuard watching the evaluation that is running now: only then auto = f16x2


@contextlib.contextmanager
def frozen_math_scope(guard):
    """Run the enclosed frozen-stage evaluation under `guard` (a slu_hip.guard.RangeGuard; None = unguarded)."""
    prev = _FrozenMath.scope
    _FrozenMath.scope = guard
    try:
        yield guard
    finally:
        _FrozenMath.scope = prev


# Round 5: the DEFAULT arithmetic of frozen stages is as wide as the reference's fp32 ATen kernels (models.py:108, 200,
# 232): bf16x3 = every fp32 operand as three bf16 terms (3 x 8 significand bits = fp32's 24, fp32's exponent range, so
# the split is EXACT), six bf16 MFMA products per fp32 product (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; what is
# dropped is below 2^-24 |a b| per product, the size of one fp32 rounding), fp32 accumulation.  The 22-bit f16x2 scheme
# under its range guard ("auto") is faster and opt-in: narrower than the reference's arithmetic, it is not the default.
DEFAULT_FROZEN_MATH = "bf16x3"


def frozen_math_mode():
    mode = os.environ.get("SLU_FROZEN_MATH", DEFAULT_FROZEN_MATH)
    if mode not in _FROZEN_MATH:
        raise ValueError("SLU_FROZEN_MATH=%r: expected one of %s" % (mode, sorted(_FROZEN_MATH)))
    return mode


def contraction_nsplit(frozen):
    """Arithmetic of the forward contractions of a stage (convolution, GRU input projection and recurrence):
      0  exact fp32 MFMA — trainable layers (default);
      2  "f16x2": fp32 values as two fp16 terms (22-bit significand), three fp16 MFMA products, two fp32
         accumulators (fp32-class: the deviation from float64 equals an fp32 fmaf chain's, csrc/slu_bf16.h) — but fp16's
         exponent range: operands below 65504.  FROZEN layers in the default mode (SLU_FROZEN_MATH=auto) use it ONLY
         inside a guarded evaluation (frozen_math_scope: the range words of slu_hip/guard.py are checked before the
         result is used, a violation re-runs the evaluation on bf16x3), or unguarded with SLU_FROZEN_MATH=f16x2;
      3  "bf16x3": three bf16 terms, six bf16 MFMA products (fp32-class, fp32's range) — FROZEN layers in the default
         mode wherever no guard is active, or always with SLU_FROZEN_MATH=bf16x3; SLU_FROZEN_MATH=fp32 switches the split
         schemes off (0);
      1  plain bf16 operands, fp32 accumulation and gate math — every layer when SLU_DTYPE=bf16
         (BASELINE configs[4]: weights and activations enter every forward contraction and the data-gradient
         contractions as bf16 — ops.bf16_mode; weight gradients, master weights and Adam stay fp32)."""
    if os.environ.get("SLU_DTYPE", "f32") == "bf16":
        return 1
    if frozen:
        mode = frozen_math_mode()
        if mode == "auto":
            return 2 if _FrozenMath.scope is not None else 3
        return _FROZEN_MATH[mode]
    return 0


_FROZEN_MATH = {"auto": None, "f16x2": 2, "bf16x3": 3, "fp32": 0}


def guarded_frozen_nsplit(model=None):
    """The split scheme of frozen stages INSIDE a guarded evaluation (what the look-ahead pipeline and the eager entry
    points run): the default mode gives f16x2 unless `model` (a PretrainedModel / Model) is pinned to bf16x3 or its
    weights are out of range; explicit modes as set.  For reports (bench.py) and tests."""
    if frozen_math_mode() != "auto" or os.environ.get("SLU_DTYPE", "f32") == "bf16":
        return contraction_nsplit(True)
    pm = getattr(model, "pretrained_model", model)
    return 2 if (pm is None or pm.f16x2_allowed()) else 3


def _require_device(t):
    if not t.is_cuda:
        raise _lib.SluHipError("the HIP kernels are the only compute path of this package: move the "
                               "model to a gfx950 (MI355X) device; there is no CPU fallback")


# ------------------------------------------------------------------------------------------------
# thin layer modules (API/state_dict compatibility; each also runs stand-alone)
# ------------------------------------------------------------------------------------------------
class Downsample(torch.nn.Module):
    """Time-axis downsampling (reference models.py:26-46): "none" = x[:, ::factor], "avg"/"max" =
    ceil-mode pooling.  Stand-alone it takes (B,T,C) like the reference."""

    def __init__(self, method="none", factor=1, axis=1):
        super().__init__()
        self.factor = factor
        self.method = method
        self.axis = axis
        if self.method not in ("none", "avg", "max"):
            print("Error: downsampling method must be one of the following: \"none\", \"avg\", \"max\"")
            sys.exit()

    def forward(self, x):
        _require_device(x)
        if self.axis != 1 or x.dim() != 3:
            raise NotImplementedError("Downsample: only axis=1 of a (B,T,C) tensor is supported")
        xt = x.transpose(0, 1).contiguous()
        return _PoolOnlyFn.apply(xt, self.method, self.factor).transpose(0, 1)


class _PoolOnlyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xt, method, factor):
        ctx.cfg = (method, factor)
        ctx.save_for_backward(xt)
        return _ops.dropout_pool_fwd(xt, None, 0.0, 0, 0, method, factor)

    @staticmethod
    def backward(ctx, dy):
        method, factor = ctx.cfg
        (xt,) = ctx.saved_tensors
        return _ops.dropout_pool_bwd(dy, xt, None, 0.0, 0, 0, method, factor), None, None


class _DropOnlyFn(torch.autograd.Function):
    """Dropout alone on a contiguous 3-D activation (the CNN blocks' nn.Dropout): slu_dropout_pool with
    factor 1.  The kernel's (T,B,C) indexing is applied to the tensor as it lies in memory."""

    @staticmethod
    def forward(ctx, x, p, mask, seed, offset):
        offset, offset_dev, sub_batch = offset if isinstance(offset, tuple) else (offset, None, 0)
        if sub_batch:
            raise NotImplementedError("CNN dropout inside a look-ahead super-batch")
        ctx.cfg = (p, seed, offset)
        ctx.offset_dev = offset_dev
        ctx.save_for_backward(x, mask)
        return _ops.dropout_pool_fwd(x, mask, p, seed, offset, "none", 1, offset_dev, 0)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset = ctx.cfg
        x, mask = ctx.saved_tensors
        return _ops.dropout_pool_bwd(dy, x, mask, p, seed, offset, "none", 1, ctx.offset_dev), None, None, None, None


class SincLayer(torch.nn.Module):
    """SincNet band-pass filterbank layer (reference models.py:49-110).  Two float64 parameters
    per filter; mel-spaced deterministic initialisation (no RNG draw)."""

    def __init__(self, N_filt, Filt_dim, fs, stride=1, padding=0, is_cuda=False):
        super().__init__()
        mel_hi = 2595 * np.log10(1 + (fs / 2) / 700)
        hz = 700 * (10 ** (np.linspace(80, mel_hi, N_filt) / 2595) - 1)
        lo, hi = np.roll(hz, 1), np.roll(hz, -1)
        lo[0] = 30
        hi[-1] = (fs / 2) - 100
        self.freq_scale = fs * 1.0
        self.filt_b1 = torch.nn.Parameter(torch.from_numpy(lo / self.freq_scale))
        self.filt_band = torch.nn.Parameter(torch.from_numpy((hi - lo) / self.freq_scale))
        self.N_filt, self.Filt_dim, self.fs = N_filt, Filt_dim, fs
        self.stride, self.padding, self.is_cuda = stride, padding, is_cuda
        if padding != Filt_dim // 2:
            raise NotImplementedError("SincLayer: only padding == Filt_dim // 2 is supported")

    def filters(self):
        """(N_filt, Filt_dim) float32 filterbank built on the device."""
        _require_device(self.filt_b1)
        return _ops.sinc_filters(self.filt_b1.detach(), self.filt_band.detach(), self.Filt_dim, self.fs)

    def forward(self, x):
        """x (B,1,T) -> (B,N_filt,L), the plain strided convolution (no abs/pool)."""
        _require_device(x)
        out = _ops.SincBlockFn.apply(x.reshape(x.shape[0], -1), self.filt_b1, self.filt_band,
                                     self.Filt_dim, self.fs, self.stride, 1, 1.0, False, False)
        return out.transpose(1, 2)


class Conv1d(torch.nn.Module):
    """Dense Conv1d(Cin, Cout, k, stride, padding=k//2) with torch's parameter names and default
    initialisation (reference models.py:190,200 use torch.nn.Conv1d)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        proto = torch.nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)
        self.weight, self.bias = proto.weight, proto.bias        # same RNG draw as the reference
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        if padding != kernel_size // 2:
            raise NotImplementedError("Conv1d: only padding == kernel_size // 2 is supported")

    def forward(self, x):
        """x (B,Cin,L) -> (B,Cout,L')"""
        _require_device(x)
        out = _ops.ConvBlockFn.apply(x.transpose(1, 2).contiguous(), self.weight, self.bias,
                                     self.stride, False, 1, 1.0, False)
        return out.transpose(1, 2)

    def extra_repr(self):
        return "%d, %d, kernel_size=(%d,), stride=(%d,), padding=(%d,)" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding)


class GRU(torch.nn.Module):
    """Single-layer (bi)GRU with torch.nn.GRU's parameter names/shapes/initialisation
    (reference models.py:232,262,686).  Stand-alone it mimics nn.GRU(batch_first=True):
    returns (output (B,T,D*H), None)."""

    def __init__(self, input_size, hidden_size, batch_first=True, bidirectional=False):
        super().__init__()
        if not batch_first:
            raise NotImplementedError("GRU: batch_first=True only (as the reference uses it)")
        proto = torch.nn.GRU(input_size=input_size, hidden_size=hidden_size, batch_first=True,
                             bidirectional=bidirectional)
        for k, v in proto.named_parameters():
            self.register_parameter(k, v)
        self.input_size, self.hidden_size, self.bidirectional = input_size, hidden_size, bidirectional
        self.batch_first = True

    def _stacked_ih(self):
        """(W_ih, b_ih) of both directions as ONE (D*3H, I) / (D*3H) storage that the four parameters
        are views of, so that the input projection (and its gradients) is a single GEMM with no
        concatenation.  The link is (re)established lazily — e.g. after .cuda() gave every parameter
        its own storage — by stacking once and re-pointing the parameters' .data at the slices;
        optimizers and state_dict keep working on the same Parameter objects."""
        if not self.bidirectional:
            return self.weight_ih_l0, self.bias_ih_l0
        w_f, w_r, b_f, b_r = self.weight_ih_l0, self.weight_ih_l0_reverse, self.bias_ih_l0, self.bias_ih_l0_reverse
        n = w_f.shape[0]
        stk = getattr(self, "_ih_storage", None)
        linked = (stk is not None and stk[0].device == w_f.device
                  and w_f.data_ptr() == stk[0].data_ptr() and w_r.data_ptr() == stk[0][n:].data_ptr()
                  and b_f.data_ptr() == stk[1].data_ptr() and b_r.data_ptr() == stk[1][n:].data_ptr())
        if not linked:
            with torch.no_grad():
                W = torch.cat([w_f.data, w_r.data])
                b = torch.cat([b_f.data, b_r.data])
                w_f.data, w_r.data = W[:n], W[n:]
                b_f.data, b_r.data = b[:n], b[n:]
            self._ih_storage = (W, b)
        return self._ih_storage

    def split_frozen(self):
        """nsplit (1 / 2 / 3) when this layer runs FROZEN on the split-precision kernels, else 0."""
        if any(q.requires_grad for q in self.parameters()):
            return 0
        nsplit = contraction_nsplit(True)
        return nsplit if _ops.split_path_supported(self.hidden_size, 2 if self.bidirectional else 1) else 0

    def run_time_major(self, xt, p=0.0, mask=None, seed=0, offset=0, method="none", factor=1, out_planes=False):
        w_ih, b_ih = self._stacked_ih()
        frozen = not any(q.requires_grad for q in self.parameters())
        nsplit = contraction_nsplit(frozen)
        packed = None
        if nsplit and frozen and _ops.split_path_supported(self.hidden_size, 2 if self.bidirectional else 1):
            # bf16 planes of the frozen W_ih in MFMA fragment order, rebuilt when the weight changes (the cache
            # lives with the weight's owner: a global table keyed by address would outlive the tensor)
            key = (nsplit, w_ih.data_ptr(), w_ih._version)
            if getattr(self, "_packed_ih", (None, None))[0] != key:
                self._packed_ih = (key, _ops.gemm_bf16_pack(w_ih.detach(), nsplit))
            packed = self._packed_ih[1]
            if nsplit == 2 and _FrozenMath.scope is not None and not isinstance(xt, _ops.SplitAct):
                # guarded f16x2 fed by an fp32 tensor no convolution launch has watched (a last CNN block with dropout,
                # pool 2 or a channel count without plane output hands fp32 over): raise the guard's last word to this
                # input's largest |value| before split_bf16 turns it into fp16 terms
                _ops.absmax_into(xt, _FrozenMath.scope.word(1 << 20))
            if isinstance(xt, _ops.SplitAct) or not xt.requires_grad:
                # nothing to differentiate: the whole layer outside autograd, activations may stay in bf16 planes
                rev = (self.weight_hh_l0_reverse, self.bias_hh_l0_reverse) if self.bidirectional else (None, None)
                with torch.no_grad():
                    return _ops.gru_layer_frozen(xt, w_ih, b_ih, packed, self.weight_hh_l0, self.bias_hh_l0, rev[0],
                                                 rev[1], p, mask, seed, offset, method, factor, nsplit, out_planes)
        if self.bidirectional:
            ih = (self.weight_ih_l0, self.weight_ih_l0_reverse, self.bias_ih_l0, self.bias_ih_l0_reverse)
            rev = (self.weight_hh_l0_reverse, self.bias_hh_l0_reverse)
        else:
            ih = (self.weight_ih_l0, None, self.bias_ih_l0, None)
            rev = (None, None)
        return _ops.GRULayerFn.apply(xt, w_ih.detach(), b_ih.detach(), *ih, self.weight_hh_l0, self.bias_hh_l0,
                                     rev[0], rev[1], p, mask, seed, offset, method, factor, nsplit, packed)

    def forward(self, x):
        _require_device(x)
        return self.run_time_major(x.transpose(0, 1)).transpose(0, 1), None

    def extra_repr(self):
        return "%d, %d, batch_first=True, bidirectional=%s" % (self.input_size, self.hidden_size, self.bidirectional)


class FinalPool(torch.nn.Module):
    """max over time of (B,T,C) (reference models.py:112-123)."""

    def forward(self, input):
        return input.max(dim=1)[0]


class NCL2NLC(torch.nn.Module):
    """(B,C,L) -> (B,L,C) view (reference models.py:125-136)."""

    def forward(self, input):
        return input.transpose(1, 2)


class RNNSelect(torch.nn.Module):
    """keeps the output sequence of an RNN's (output, h_n) tuple (reference models.py:138-149)."""

    def forward(self, input):
        return input[0]


class Abs(torch.nn.Module):
    """reference models.py:163-168"""

    def forward(self, input):
        return torch.abs(input)


def _named(layer, name):
    layer.name = name
    return layer


# ------------------------------------------------------------------------------------------------
# seq2seq intent head (reference models.py:381-651): same class / attribute / state_dict names; every contraction,
# the GRUCell gates, the attention and the log-softmax pick run on the HIP kernels (ops.Seq2SeqDecoderFn and
# ops.decoder_step); the beam bookkeeping (top-k, re-ordering of hypotheses) is host-side torch as in the reference.
# ------------------------------------------------------------------------------------------------
class Seq2SeqEncoder(torch.nn.Module):
    """Stack of bidirectional GRU layers, each followed by RNNSelect and Dropout(0.5) (reference models.py:381-416)."""

    def __init__(self, input_dim, num_layers, encoder_dim):
        super().__init__()
        layers, self._stages = [], []
        out_dim = input_dim
        for idx in range(num_layers):
            gru = _named(GRU(input_size=out_dim, hidden_size=encoder_dim, batch_first=True, bidirectional=True),
                         "intent_encoder_rnn%d" % idx)
            layers.append(gru)
            out_dim = 2 * encoder_dim
            layers.append(_named(RNNSelect(), "intent_encoder_rnn_select%d" % idx))
            layers.append(_named(torch.nn.Dropout(p=0.5), "intent_encoder_dropout%d" % idx))
            self._stages.append(_RnnStage(gru, "intent_encoder_dropout%d" % idx, 0.5, "none", 1))
        self.layers = torch.nn.ModuleList(layers)

    def run_time_major(self, h, training):
        for st in self._stages:
            h = st.run(h, training)
        return h

    def forward(self, x):
        """(B, T, C) -> (B, T, 2 * encoder_dim)"""
        _require_device(x)
        _DropoutState.current = next_rng_step()
        return self.run_time_major(x.transpose(0, 1), self.training).transpose(0, 1)


class Attention(torch.nn.Module):
    """Dot-product attention of one decoder state over the encoder states (reference models.py:418-438)."""

    def __init__(self, encoder_dim, decoder_dim, key_dim, value_dim):
        super().__init__()
        self.scale_factor = torch.sqrt(torch.tensor(key_dim).float())
        self.key_linear = torch.nn.Linear(encoder_dim, key_dim)
        self.query_linear = torch.nn.Linear(decoder_dim, key_dim)
        self.value_linear = torch.nn.Linear(encoder_dim, value_dim)
        self.softmax = torch.nn.Softmax(dim=1)

    def forward(self, encoder_states, decoder_state):
        """encoder_states (B, T, encoder_dim), decoder_state (B, decoder_dim) -> (B, value_dim); inference helper
        (no gradient path: training goes through Seq2SeqDecoder.forward's fused Function)."""
        _require_device(encoder_states)
        with torch.no_grad():
            B, T, C = encoder_states.shape
            enc = encoder_states.transpose(0, 1).contiguous().view(T * B, C)
            keys = _ops.gemm(enc, self.key_linear.weight.t(), self.key_linear.bias).view(T, B, -1)
            values = _ops.gemm(enc, self.value_linear.weight.t(), self.value_linear.bias).view(T, B, -1)
            q = _ops.gemm(decoder_state.contiguous(), self.query_linear.weight.t(), self.query_linear.bias)
            ctx = torch.empty(B, values.shape[2], dtype=torch.float32, device=enc.device)
            w = torch.empty(B, T, dtype=torch.float32, device=enc.device)
            _ops.attention_fwd(keys, values, q, ctx, w, 1.0 / float(self.scale_factor))
        return ctx


class DecoderRNN(torch.nn.Module):
    """Stacked GRUCells with Dropout between them (reference models.py:440-485).  The cells are parameter holders
    with torch.nn.GRUCell's names / shapes / initialisation; the arithmetic is ops.decoder_step's."""

    def __init__(self, num_decoder_layers, num_decoder_hidden, input_size, dropout):
        super().__init__()
        layers = []
        self.num_layers = num_decoder_layers
        for index in range(num_decoder_layers):
            cell = torch.nn.GRUCell(input_size=input_size if index == 0 else num_decoder_hidden,
                                    hidden_size=num_decoder_hidden)
            layers.append(_named(cell, "gru%d" % index))
            layers.append(_named(torch.nn.Dropout(p=dropout), "dropout%d" % index))
        self.layers = torch.nn.ModuleList(layers)
        self.dropout = dropout

    def cells(self):
        return [l for l in self.layers if isinstance(l, torch.nn.GRUCell)]

    def forward(self, input, previous_state):
        """input (B, input_size), previous_state (B, num_layers, hidden) -> the new state (B, num_layers, hidden)
        (reference models.py:459-484: each GRUCell takes the previous cell's output through the Dropout between them; the
        Dropout behind the last cell does not reach the returned state).  On the HIP cell kernels (ops.decoder_step's:
        slu_gemm_small_batched + slu_gru_cell_fwd).  A stand-alone call is a forward evaluation: NO gradient flows through
        it — the decoder is trained through Seq2SeqDecoder.forward, whose autograd Function owns the backward kernels —
        so tensors that require a gradient are refused instead of being silently detached."""
        _require_device(input)
        if torch.is_grad_enabled() and (input.requires_grad or previous_state.requires_grad):
            raise NotImplementedError("DecoderRNN.forward is a forward evaluation on the HIP kernels: call it under "
                                      "torch.no_grad() or with detached tensors (Seq2SeqDecoder.forward trains the decoder)")
        cells = self.cells()
        Lc, Dd = len(cells), cells[0].hidden_size
        with torch.no_grad():
            x_in = input.detach().float().contiguous()
            prev = previous_state.detach().float().contiguous()
            B = x_in.shape[0]
            dev = x_in.device
            state = torch.empty(B, Lc, Dd, dtype=torch.float32, device=dev)
            seed = _DropoutState.seed if _DropoutState.seed is not None else torch.initial_seed()
            offset = _DropoutState.current * 16 + _DECODER_SITE
            for l, cell in enumerate(cells):
                gi = torch.empty(B, 3 * Dd, dtype=torch.float32, device=dev)
                gh = torch.empty(B, 3 * Dd, dtype=torch.float32, device=dev)
                _ops.gemm_small_batched([(x_in, cell.weight_ih.detach(), cell.bias_ih.detach(), gi, 0, 0),
                                         (prev[:, l], cell.weight_hh.detach(), cell.bias_hh.detach(), gh, 0, 0)])
                p = self.dropout if (self.training and l < Lc - 1) else 0.0
                drop = torch.empty(B, Dd, dtype=torch.float32, device=dev) if p > 0.0 else None
                _ops.gru_cell_fwd(gi, gh, prev[:, l], state[:, l], None, drop, None, p, seed & 0xFFFFFFFFFFFFFFFF, offset,
                                  None, l * B * Dd)
                x_in = drop if drop is not None else state[:, l]
        return state


def sort_beam(beam_extensions, beam_extension_scores, beam_pointers):
    """Order the candidate extensions of every utterance by score, descending (reference models.py:487-502).
    Lists of W tensors (B, V) / (B) / (B) -> stacked (W, B, V), (W, B), (W, B).
    Ties: the reference calls torch.sort without `stable`, i.e. leaves the order of EQUAL scores to the sort implementation;
    here the sort is stable — equal scores keep their candidate order (source hypothesis major, then extension rank: the order
    in which the reference appends them, models.py:622-632) — so the result is deterministic on every device."""
    ext, sc, ptr = torch.stack(beam_extensions), torch.stack(beam_extension_scores), torch.stack(beam_pointers)
    sc = sc.view(len(beam_pointers), -1)
    order = sc.sort(dim=0, descending=True, stable=True)[1]
    cols = torch.arange(sc.shape[1], device=sc.device)
    return ext[order, cols], sc[order, cols], ptr[order, cols]


class Seq2SeqDecoder(torch.nn.Module):
    """Attention-based decoder for seq2seq SLU (reference models.py:504-651)."""

    def __init__(self, num_labels, num_layers, encoder_dim, decoder_dim, key_dim, value_dim, SOS=0):
        super().__init__()
        embedding_dim = decoder_dim
        self.embed = torch.nn.Linear(num_labels, embedding_dim)
        self.attention = Attention(encoder_dim * 2, decoder_dim, key_dim, value_dim)
        self.rnn = DecoderRNN(num_layers, decoder_dim, embedding_dim + value_dim, dropout=0.5)
        self.initial_state = torch.nn.Parameter(torch.randn(num_layers, decoder_dim))
        self.linear = torch.nn.Linear(decoder_dim, num_labels)
        self.log_softmax = torch.nn.LogSoftmax(dim=1)
        self.SOS = SOS

    def _params(self):
        """(names, tensors) in ops.Seq2SeqDecoderFn's argument order."""
        a = self.attention
        items = [("embed.weight", self.embed.weight), ("embed.bias", self.embed.bias),
                 ("key.weight", a.key_linear.weight), ("key.bias", a.key_linear.bias),
                 ("query.weight", a.query_linear.weight), ("query.bias", a.query_linear.bias),
                 ("value.weight", a.value_linear.weight), ("value.bias", a.value_linear.bias)]
        for l, cell in enumerate(self.rnn.cells()):
            items += [("w_ih%d" % l, cell.weight_ih), ("w_hh%d" % l, cell.weight_hh),
                      ("b_ih%d" % l, cell.bias_ih), ("b_hh%d" % l, cell.bias_hh)]
        items += [("initial_state", self.initial_state), ("linear.weight", self.linear.weight),
                  ("linear.bias", self.linear.bias)]
        return tuple(n for n, _ in items), [t for _, t in items]

    def teacher_forced(self, enc_tm, y):
        """enc_tm time-major (T, B, 2 * encoder_dim), y (B, U, num_labels) -> (loss_acc (2) = [-mean log p, 0],
        log_p (B)) on the current dropout step (models._DropoutState)."""
        p = self.rnn.dropout if self.training else 0.0
        masks, seed, offset, offset_dev = None, 0, 0, None
        if p > 0.0:
            if _DropoutState.masks is not None:
                masks = _DropoutState.masks
            else:
                seed = (_DropoutState.seed if _DropoutState.seed is not None else torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
                if _DropoutState.current_dev is not None:
                    offset, offset_dev = _DECODER_SITE, _DropoutState.current_dev
                else:
                    offset = _DropoutState.current * 16 + _DECODER_SITE
        names, tensors = self._params()
        return _ops.Seq2SeqDecoderFn.apply(enc_tm, y, (names, self.SOS, p, masks, seed, offset, offset_dev), *tensors)

    def forward(self, encoder_outputs, y, y_lengths=None):
        """encoder_outputs (B, T, 2 * encoder_dim), y (B, U, num_labels) one-hot, padded with <eos> -> log p(y|x) (B)
        (reference models.py:519-557; y_lengths is unused there too)."""
        _require_device(encoder_outputs)
        _DropoutState.current = next_rng_step()
        _, log_p = self.teacher_forced(encoder_outputs.transpose(0, 1), y)
        return log_p

    def infer(self, encoder_outputs, Sy, B=4, debug=False, y_lengths=None):
        """Beam search of width B for argmax_y log p(y|x) (reference models.py:559-651; B = 1 is greedy search).
        -> (beam_scores (B, batch), beam (B, batch, U, |Sy|) one-hot), U = 200 or max(y_lengths).
        All B hypotheses of all utterances advance in ONE batched decoder step on the HIP kernels (rows w * batch +
        b); the first input is the all-zero vector and only hypothesis 0 is expanded at the first step, as in the
        reference.  Candidate selection is host-side torch (top-k per hypothesis, then the B best of the B * B, by a STABLE
        descending sort over the reference's candidate order — source hypothesis major — so exact ties resolve as
        sort_beam's).  Against the reference's own run the BEST hypothesis of every utterance is identical; hypotheses
        further down the beam can differ where two candidates' scores are closer than the fp32 round-off between the two
        evaluations (1e-6 on log-probabilities of ~-10: the search continues for U = 200 steps past <eos>, where many
        continuations are nearly equally (im)probable) — that is a property of comparing two fp32 implementations, not of
        the tie rule (tests/test_hip_seq2seq.py::test_tiny_seq2seq_beam_search_vs_reference: best hypothesis identical,
        >= 95 % of all hypothesis labels, scores to 1e-4 relative)."""
        _require_device(encoder_outputs)
        dev = encoder_outputs.device
        W, bsz, V = B, encoder_outputs.shape[0], len(Sy)
        U = 200 if y_lengths is None else max(y_lengths)
        names, tensors = self._params()
        P = {n: t.detach() for n, t in zip(names, tensors)}
        Kd, Vd = P["key.weight"].shape[0], P["value.weight"].shape[0]
        P["inv_scale"] = 1.0 / float(self.attention.scale_factor)
        Lc, Dd = P["initial_state"].shape
        E = P["embed.weight"].shape[0]
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        with torch.no_grad():
            enc = encoder_outputs.detach().float().transpose(0, 1).contiguous()          # (T, bsz, C)
            T = enc.shape[0]
            enc2 = enc.view(T * bsz, -1)
            # keys / values once, replicated per hypothesis: row w * bsz + b of the step batch reads utterance b
            keys = _ops.gemm(enc2, P["key.weight"].t(), P["key.bias"]).view(T, 1, bsz, Kd).expand(T, W, bsz, Kd).reshape(T, W * bsz, Kd)
            values = _ops.gemm(enc2, P["value.weight"].t(), P["value.bias"]).view(T, 1, bsz, Vd).expand(T, W, bsz, Vd).reshape(T, W * bsz, Vd)
            R = W * bsz
            state, state_next = f(R, Lc, Dd), f(R, Lc, Dd)
            _ops.broadcast_rows(P["initial_state"].contiguous().view(-1), state.view(R, Lc * Dd))
            y_prev = torch.zeros(R, V, dtype=torch.float32, device=dev)
            q, inp0, att_w, logits = f(R, Kd), f(R, E + Vd), f(R, T), f(R, V)
            gi, gh, lse, sink = f(R, 3 * Dd), f(Lc, R, 3 * Dd), f(R), f(R)
            drop = [f(R, Dd) for _ in range(Lc - 1)]
            hyp = torch.zeros(W, bsz, U, dtype=torch.int64, device=dev)
            scores = torch.zeros(W, bsz, dtype=torch.float32, device=dev)
            cols = torch.arange(bsz, device=dev)
            zeros_y = torch.zeros(R, V, dtype=torch.float32, device=dev)
            for u in range(U):
                _ops.decoder_step(P, keys, values, state, state_next, y_prev, q, inp0, att_w, gi, gh, None, drop, logits, u,
                                  (0.0, None, 0, 0, None, R * Dd))
                _ops.logsoftmax_dot_fwd(logits, zeros_y, sink, lse)                     # lse only (y = 0)
                top_s, top_i = logits.topk(W, dim=1)                                    # log_softmax keeps the order
                cand = (top_s - lse.unsqueeze(1)).view(W, bsz, W) + scores.unsqueeze(2)  # [src hypothesis, b, extension]
                if u == 0:
                    cand[1:] = float("-inf")
                flat = cand.permute(1, 0, 2).reshape(bsz, W * W)                        # candidate order: src major
                best, pick = flat.sort(dim=1, descending=True, stable=True)
                best, pick = best[:, :W].t().contiguous(), pick[:, :W].t()              # (W, bsz)
                src, ext = pick // W, pick % W
                label = top_i.view(W, bsz, W)[src, cols.unsqueeze(0), ext]              # (W, bsz)
                hyp = hyp[src, cols.unsqueeze(0)]
                hyp[:, :, u] = label
                scores = best
                state = state_next.view(W, bsz, Lc, Dd)[src, cols.unsqueeze(0)].reshape(R, Lc, Dd)
                y_prev = torch.zeros(R, V, dtype=torch.float32, device=dev)
                y_prev.scatter_(1, label.reshape(R, 1), 1.0)
                if debug:
                    print("step %d | best score of utterance 0: %1.2f" % (u, scores[0, 0].item()))
            beam = torch.zeros(W, bsz, U, V, dtype=torch.float32, device=dev)
            beam.scatter_(3, hyp.unsqueeze(3), 1.0)
        return scores, beam


# ------------------------------------------------------------------------------------------------
# fused stage plan
# ------------------------------------------------------------------------------------------------
class _ConvStage:
    """[conv|sinc, (abs), pool, act, dropout] of one CNN block (reference models.py:182-220)."""

    def __init__(self, conv, is_sinc, do_abs, pool, act, drop, idx=0):
        self.conv, self.is_sinc, self.do_abs, self.pool, self.drop = conv, is_sinc, do_abs, pool, drop
        self.idx = idx
        self.drop_name, self.site = "dropout%d" % idx, (_site("cnn", idx) if drop > 0.0 else -1)
        self.slope = 0.2 if act == "leaky_relu" else 0.0
        self.time_major = False       # set on the last CNN stage: its output feeds the RNN stack

    def parameters(self):
        return list(self.conv.parameters())

    def _frozen_cache(self, nsplit):
        """What a FROZEN block recomputed per call in round 2: the Sinc filterbank and the filters packed in MFMA
        fragment order.  Kept per (weight version, arithmetic, HIP stream): a look-ahead slot's graph reads its own
        copy, built by that stream's first (eager) call, so no stream ever waits for another one's pack."""
        key = (nsplit, torch.cuda.current_stream().cuda_stream)
        version = tuple((q.data_ptr(), q._version) for q in self.parameters())
        caches = self.__dict__.setdefault("_caches", {})
        c = caches.get(key)
        if c is None or c["version"] != version:
            # one entry per (arithmetic, stream); new weights REPLACE that entry only.  (Round 3 cleared the whole table
            # at eight entries, freeing packs that other streams' captured graphs still read.  A graph captured with the
            # replaced entry belongs to the old weight version and is dropped by its owner: PrefixSlot.signature.)
            c = caches[key] = {"version": version, "filters": None, "pack": {}}
        return c

    def run(self, h, training, out_planes=False):
        """h: (B,T) for the first block, else channels-last (B,L,C).
        out_planes: the consumer is a frozen split-precision GRU layer — hand over bf16 planes (SplitAct) when this
        block runs on the split-precision kernel itself, time-major and without dropout (else plain fp32)."""
        time_major = self.time_major
        fused_pool = self.pool in (1, 2)
        tm = time_major and fused_pool and self.drop == 0.0
        pool = self.pool if fused_pool else 1
        slope = self.slope if fused_pool else 1.0
        # FROZEN block with nothing to differentiate: the convolution on the split-precision kernels, outside
        # autograd (same decision in the pipelined and the sequential loop: it depends on requires_grad only)
        # (a TRAINABLE block in bf16 mode takes bf16 operands inside ops.SincBlockFn / ConvBlockFn: forward on the bf16
        # kernel with the route bits, fp32 backward)
        nsplit = contraction_nsplit(True) if not any(q.requires_grad for q in self.parameters()) else 0
        c_in = 1 if (self.is_sinc or h.dim() == 2) else h.shape[2]
        k_t = self.conv.Filt_dim if self.is_sinc else self.conv.kernel_size
        if (nsplit and fused_pool and not h.requires_grad
                and _ops.wconv_bf16_supported(c_in, self.conv.stride, pool, k_t, nsplit)):
            with torch.no_grad():
                cache = self._frozen_cache(nsplit)
                if self.is_sinc:
                    if cache.get("filters") is None and not torch.cuda.is_current_stream_capturing():
                        cache["filters"] = self.conv.filters()
                    filt = cache.get("filters")
                    w = (filt if filt is not None else self.conv.filters()).view(self.conv.N_filt, 1, self.conv.Filt_dim)
                    bias, do_abs = None, self.do_abs
                else:
                    w, bias, do_abs = self.conv.weight.detach(), self.conv.bias.detach(), self.do_abs
                if isinstance(h, _ops.RowTable):       # a look-ahead super-batch read where its batches lie
                    x3, (B, l_in) = h, h.shape
                else:
                    x3 = (h if h.dim() == 3 else h.unsqueeze(2)).contiguous()      # (int16 PCM samples stay int16)
                    B, l_in = x3.shape[0], x3.shape[1]
                planes = (out_planes and tm and not (self.drop > 0.0 and training)
                          and _ops.wconv_bf16_planes_ok(w.shape[0], pool))
                scope = _FrozenMath.scope
                h = _ops.wconv_fwd_bf16(x3, w, bias, B, l_in, c_in, self.conv.stride, do_abs, pool, slope,
                                        tm, nsplit, planes, pack_cache=cache["pack"],
                                        absmax=scope.word(self.idx) if (scope is not None and nsplit == 2) else None)
                if planes:
                    return h
            if self.drop > 0.0 and training:
                p, mask, seed, offset = _dropout_args(self.drop_name, self.site, self.drop, training, cnn=True)
                h = _DropOnlyFn.apply(h.contiguous(), p, mask, seed, offset)
            if time_major and not tm:
                h = h.transpose(0, 1).contiguous()
            return h
        if isinstance(h, _ops.RowTable):
            raise _lib.SluHipError("a row-pointer table can only be read by the split-precision first block")
        if h.dtype == torch.int16:             # PCM16 batch, first block on the exact fp32 kernels: sample / 32768
            h = _ops.pcm16_to_f32(h)
        if self.is_sinc:
            h = _ops.SincBlockFn.apply(h, self.conv.filt_b1, self.conv.filt_band, self.conv.Filt_dim,
                                       self.conv.fs, self.conv.stride, pool, slope, tm,
                                       self.do_abs and fused_pool)
        else:
            if h.dim() == 2:
                h = h.unsqueeze(2)
            h = _ops.ConvBlockFn.apply(h, self.conv.weight, self.conv.bias, self.conv.stride,
                                       self.do_abs and fused_pool, pool, slope, tm)
        if not fused_pool:          # pool widths the convolution's epilogue does not fuse (> 2): slu_pool_act_*
            h = _ops.PoolActFn.apply(h, self.pool, self.do_abs, self.slope, False)
        if self.drop > 0.0 and training:
            # nn.Dropout of the CNN block (models.py:217-220) on the same step-indexed Philox stream as
            # the RNN sites (or the injected mask, given in the reference's (B,C,L) shape)
            p, mask, seed, offset = _dropout_args(self.drop_name, self.site, self.drop, training, cnn=True)
            h = _DropOnlyFn.apply(h.contiguous(), p, mask, seed, offset)
        if time_major and not tm:
            h = h.transpose(0, 1).contiguous()
        return h


class _RnnStage:
    """[gru, select, dropout, downsample] (reference models.py:230-253, 260-283, 684-707)."""

    def __init__(self, gru, drop_name, p, method, factor):
        self.gru, self.drop_name, self.p, self.method, self.factor = gru, drop_name, p, method, factor
        module, idx = drop_name.split("_dropout")
        self.site = _site(module, int(idx)) if p > 0.0 else -1

    def parameters(self):
        return list(self.gru.parameters())

    def run(self, xt, training, out_planes=False):
        """out_planes: the consumer is another frozen split-precision GRU layer: hand over bf16 planes."""
        p, mask, seed, offset = _dropout_args(self.drop_name, self.site, self.p, training)
        return self.gru.run_time_major(xt, p, mask, seed, offset, self.method, self.factor, out_planes)


def _build_rnn_stack(layers, stages, prefix, in_dim, hidden, bidirectional, drops, ds_types, ds_lens):
    """Appends the reference's 4 modules per RNN layer to `layers` and one fused stage to `stages`."""
    out_dim = in_dim
    for idx, H in enumerate(hidden):
        gru = _named(GRU(input_size=out_dim, hidden_size=H, batch_first=True, bidirectional=bidirectional),
                     "%s_rnn%d" % (prefix, idx))
        layers.append(gru)
        out_dim = H * (2 if bidirectional else 1)
        layers.append(_named(RNNSelect(), "%s_rnn_select%d" % (prefix, idx)))
        layers.append(_named(torch.nn.Dropout(p=drops[idx]), "%s_dropout%d" % (prefix, idx)))
        layers.append(_named(Downsample(method=ds_types[idx], factor=ds_lens[idx], axis=1),
                             "%s_downsample%d" % (prefix, idx)))
        stages.append(_RnnStage(gru, "%s_dropout%d" % (prefix, idx), drops[idx], ds_types[idx], ds_lens[idx]))
    return out_dim


class PretrainedModel(torch.nn.Module):
    """Encoder pre-trained to recognise phonemes and words (reference models.py:170-361)."""

    def __init__(self, config):
        super().__init__()
        self.is_cuda = torch.cuda.is_available()
        phoneme_layers, self._cnn_stages, self._phone_stages = [], [], []
        n_conv = len(config.cnn_N_filt)
        for idx in range(n_conv):
            k, stride = config.cnn_len_filt[idx], config.cnn_stride[idx]
            is_sinc = idx == 0 and config.use_sincnet
            if is_sinc:
                conv = _named(SincLayer(config.cnn_N_filt[0], k, config.fs, stride=stride, padding=k // 2,
                                        is_cuda=self.is_cuda), "sinc0")
            else:
                cin = 1 if idx == 0 else config.cnn_N_filt[idx - 1]
                conv = _named(Conv1d(cin, config.cnn_N_filt[idx], k, stride=stride, padding=k // 2),
                              "conv%d" % idx)
            phoneme_layers.append(conv)
            if idx == 0:
                phoneme_layers.append(_named(Abs(), "abs0"))
            phoneme_layers.append(_named(torch.nn.MaxPool1d(config.cnn_max_pool_len[idx], ceil_mode=True),
                                         "pool%d" % idx))
            act = torch.nn.LeakyReLU(0.2) if config.cnn_act[idx] == "leaky_relu" else torch.nn.ReLU()
            phoneme_layers.append(_named(act, "act%d" % idx))
            phoneme_layers.append(_named(torch.nn.Dropout(p=config.cnn_drop[idx]), "dropout%d" % idx))
            self._cnn_stages.append(_ConvStage(conv, is_sinc, idx == 0, config.cnn_max_pool_len[idx],
                                               config.cnn_act[idx], config.cnn_drop[idx], idx))
        phoneme_layers.append(_named(NCL2NLC(), "ncl2nlc"))
        out_dim = _build_rnn_stack(phoneme_layers, self._phone_stages, "phone", config.cnn_N_filt[-1],
                                   config.phone_rnn_num_hidden, config.phone_rnn_bidirectional,
                                   config.phone_rnn_drop, config.phone_downsample_type,
                                   config.phone_downsample_len)
        self.phoneme_layers = torch.nn.ModuleList(phoneme_layers)
        self.phoneme_linear = torch.nn.Linear(out_dim, config.num_phonemes)

        word_layers, self._word_stages = [], []
        out_dim = _build_rnn_stack(word_layers, self._word_stages, "word", out_dim,
                                   config.word_rnn_num_hidden, config.word_rnn_bidirectional,
                                   config.word_rnn_drop, config.word_downsample_type,
                                   config.word_downsample_len)
        self.word_layers = torch.nn.ModuleList(word_layers)
        self.word_linear = torch.nn.Linear(out_dim, config.vocabulary_size)
        self.pretraining_type = config.pretraining_type
        if self.is_cuda:
            self.cuda()

    # -- fused execution ------------------------------------------------------------------------
    def _to_device(self, *tensors):
        self.is_cuda = next(self.parameters()).is_cuda
        if not self.is_cuda:
            raise _lib.SluHipError("model parameters are not on a GPU: the HIP kernels are the only "
                                   "compute path (no CPU fallback)")
        dev = next(self.parameters()).device
        return [t.to(dev, non_blocking=True) if t is not None else None for t in tensors]

    def _stages(self):
        return self._cnn_stages + self._phone_stages + self._word_stages

    # -- default arithmetic of frozen stages: guarded f16x2 (slu_hip/guard.py) -----------------------------------
    def _frozen_split_tensors(self):
        """The fp32 tensors the split-precision kernels of the FROZEN stages split: filters and GRU matrices."""
        out = []
        for st in self._stages():
            if any(q.requires_grad for q in st.parameters()):
                continue
            if isinstance(st, _ConvStage):
                out.append(st.conv.filters() if st.is_sinc else st.conv.weight)
            else:
                g = st.gru
                out += [q for n, q in g.named_parameters() if n.startswith("weight")]
        return out

    def f16x2_allowed(self):
        """May a guarded evaluation of this model's frozen stages use f16x2?  Default mode, not pinned to bf16x3 by an
        earlier range violation, frozen weights inside fp16's comfortable range (checked once per weight version: one
        launch + one read-back, never under capture)."""
        if frozen_math_mode() != "auto" or os.environ.get("SLU_DTYPE", "f32") == "bf16":
            return False
        if getattr(self, "_f16x2_pin", None) is not None:
            return False
        sig = tuple((q.data_ptr(), q._version, q.requires_grad) for q in self.parameters())
        cached = getattr(self, "_f16x2_weights", None)
        if cached is None or cached[0] != sig:
            if torch.cuda.is_current_stream_capturing() or not next(self.parameters()).is_cuda:
                return False
            from slu_hip import guard as _guard
            with torch.no_grad():
                ok, worst = _guard.weights_in_range(self._frozen_split_tensors())
            if not ok:
                print("frozen stages on bf16x3: a frozen weight tensor's largest entry (%.3g) is outside the f16x2 "
                      "scheme's range [%.3g, %.0f)" % (worst, _guard.WEIGHT_MIN, _guard.F16X2_LIMIT))
            cached = self._f16x2_weights = (sig, ok)
        return cached[1]

    def pin_bf16x3(self, why):
        """A range violation was observed: this model's frozen stages stay on bf16x3 from now on."""
        if getattr(self, "_f16x2_pin", None) is None:
            print("frozen stages pinned to bf16x3: %s" % why)
        self._f16x2_pin = why

    def range_guard(self):
        from slu_hip import guard as _guard
        dev = next(self.parameters()).device
        g = getattr(self, "_range_guard", None)
        if g is None or g.device != dev:
            g = self._range_guard = _guard.RangeGuard(dev)
        return g

    def run_stages(self, h, first, last):
        """Run fused stages [first, last) of the encoder.  In the default arithmetic mode an evaluation that contains
        FROZEN stages and that nobody else guards (the look-ahead pipeline guards its super-batches itself) runs them on
        f16x2 under this model's range guard: the range words are read back before the result is returned (one stream
        synchronisation — these are the eager paths: inference, evaluation, un-captured steps) and a violation repeats
        the evaluation on bf16x3.  Under hipGraph capture no guard can act: frozen stages then run on bf16x3."""
        stages = self._stages()
        if (_FrozenMath.scope is None and not torch.cuda.is_current_stream_capturing()
                and any(not any(q.requires_grad for q in st.parameters()) for st in stages[first:last])
                and self.f16x2_allowed()):
            g = self.range_guard()
            g.arm()
            with frozen_math_scope(g):
                out = self._run_stages(h, first, last)
            g.collect()
            torch.cuda.current_stream().synchronize()
            overflow, quiet, seen = g.verdict()
            if overflow or quiet:
                if overflow:
                    self.pin_bf16x3("a split-precision stage saw |value| = %.3g (limit %.0f)" % (max(seen), 65504.0))
                out = self._run_stages(h, first, last)          # no scope: bf16x3
            return out
        return self._run_stages(h, first, last)

    def _run_stages(self, h, first, last):
        """Run fused stages [first, last) of the encoder (CNN blocks, then phoneme and word RNN
        layers).  Hand-off layouts: (B,T) waveform -> channels-last (B,L,C) between CNN blocks ->
        time-major (T,B,C) from the last CNN block on."""
        self._cnn_stages[-1].time_major = True
        if first == 0 and h.dtype != torch.int16:       # PCM16 batches stay int16: the first block scales them (ops.PCM16_SCALE)
            h = h.float()
        stages = self._stages()
        for k in range(first, last):
            st = stages[k]
            if isinstance(st, _RnnStage):
                # two consecutive frozen GRU layers inside this call: the activation between them stays in the
                # split-precision format (bf16 planes written by the dropout+pool kernel, read by the GEMM)
                nxt = stages[k + 1] if k + 1 < last else None
                planes = (isinstance(nxt, _RnnStage) and st.gru.split_frozen() > 0
                          and nxt.gru.split_frozen() == st.gru.split_frozen())
                h = st.run(h, self.training, planes)
            else:
                # the last CNN block feeding a frozen split-precision GRU layer: bf16 planes straight from the
                # convolution's epilogue (no fp32 round trip, no split pass)
                nxt = stages[k + 1] if k + 1 < last else None
                nsplit_c = contraction_nsplit(True) if not any(q.requires_grad for q in st.parameters()) else 0
                planes = (st is self._cnn_stages[-1] and isinstance(nxt, _RnnStage) and nsplit_c > 0
                          and nxt.gru.split_frozen() == nsplit_c)
                h = st.run(h, self.training, planes)
        return h

    def stage_parameters(self):
        """[[parameters of stage 0], [of stage 1], ...] — the stages and their parameter OBJECTS are fixed after
        construction (requires_grad / versions are read live by the callers), so the module-tree walk is done once: the
        trainer asks at the start of every run, in front of the first launch."""
        cached = getattr(self, "_stage_params", None)
        stages = self._stages()
        if cached is None or len(cached) != len(stages):
            cached = self._stage_params = [list(st.parameters()) for st in stages]
        return cached

    def frozen_prefix_len(self):
        """Number of leading stages none of whose parameters is trainable."""
        n = 0
        for ps in self.stage_parameters():
            if any(p.requires_grad for p in ps):
                break
            n += 1
        return n

    def accepts_row_table(self):
        """Can stage 0 read a look-ahead super-batch through a row-pointer table (ops.RowTable) instead of a
        concatenated copy?  Only the split-precision convolution kernel of a FROZEN first block does."""
        st = self._cnn_stages[0]
        if any(q.requires_grad for q in st.parameters()) or not contraction_nsplit(True):
            return False
        fused_pool = st.pool in (1, 2)
        k_t = st.conv.Filt_dim if st.is_sinc else st.conv.kernel_size
        return fused_pool and _ops.wconv_bf16_supported(1, st.conv.stride, st.pool if fused_pool else 1, k_t,
                                                        contraction_nsplit(True))

    def warm_weight_caches(self):
        """Establish the direction-stacked input-projection storage of every GRU layer on the current
        stream (so that side streams only ever read it)."""
        for st in self._phone_stages + self._word_stages:
            st.gru._stacked_ih()

    def _phoneme_features_tm(self, x):
        """x (B,T) on device -> time-major (T', B, C) output of the phoneme module."""
        return self.run_stages(x, 0, len(self._cnn_stages) + len(self._phone_stages))

    def _word_features_tm(self, h):
        n = len(self._cnn_stages) + len(self._phone_stages)
        return self.run_stages(h, n, n + len(self._word_stages))

    def _features_tm(self, x):
        return self.run_stages(x, 0, len(self._stages()))

    # -- reference API --------------------------------------------------------------------------
    def forward(self, x, y_phoneme, y_word, rng_step=None):
        """x (B,T), y_phoneme (B,T'), y_word (B,T'') -> (phoneme_loss, word_loss, phoneme_acc,
        word_acc); cross-entropy ignores label -1 (reference models.py:291-331).
        rng_step (not in the reference): the dropout stream index of this forward — None = the next
        one, or a 1-element int64 CUDA tensor holding step*16 (hipGraph-captured steps)."""
        x, y_phoneme, y_word = self._to_device(x, y_phoneme, y_word)
        if torch.is_tensor(rng_step):
            _DropoutState.current_dev = rng_step
        else:
            _DropoutState.current = next_rng_step() if rng_step is None else rng_step
        try:
            ph_tm = self._phoneme_features_tm(x)                         # (T',B,C)
            # Linear + cross-entropy(ignore_index=-1) + frame accuracy: slu_gemm_f32 + slu_frame_ce_fwd
            pl = self.phoneme_linear
            phoneme_loss, phoneme_acc = _ops.FrameHeadFn.apply(ph_tm, pl.weight, pl.bias, y_phoneme)
            if self.pretraining_type == 1:
                return phoneme_loss, torch.tensor([0.]), phoneme_acc, torch.tensor([0.])
            wd_tm = self._word_features_tm(ph_tm)
            wl = self.word_linear
            word_loss, word_acc = _ops.FrameHeadFn.apply(wd_tm, wl.weight, wl.bias, y_word)
            return phoneme_loss, word_loss, phoneme_acc, word_acc
        finally:
            _DropoutState.current_dev = None

    def compute_posteriors(self, x):
        (x,) = self._to_device(x)
        _DropoutState.current = next_rng_step()
        ph_tm = self._phoneme_features_tm(x)
        wd_tm = self._word_features_tm(ph_tm)

        def head(lin, h_tm):                       # Linear on every frame: slu_gemm_f32 (no gradient path: inference)
            T, B, C = h_tm.shape
            out = _ops.gemm(h_tm.detach().contiguous().view(T * B, C), lin.weight.detach().t(), lin.bias.detach())
            return out.view(T, B, -1).transpose(0, 1)
        return head(self.phoneme_linear, ph_tm), head(self.word_linear, wd_tm)

    def compute_features(self, x):
        """(B,T) waveform -> (B,T',C) encoder features (reference models.py:349-361)."""
        (x,) = self._to_device(x)
        _DropoutState.current = next_rng_step()
        return self._features_tm(x).transpose(0, 1)


def freeze_layer(layer):
    for param in layer.parameters():
        param.requires_grad = False


def unfreeze_layer(layer):
    for param in layer.parameters():
        param.requires_grad = True


def has_params(layer):
    return sum(p.numel() for p in layer.parameters()) > 0


def is_frozen(layer):
    return not any(p.requires_grad for p in layer.parameters())


class Model(torch.nn.Module):
    """End-to-end SLU model: pre-trained encoder + intent module (reference models.py:653-874): fixed-length
    multi-slot output, or (config.seq2seq) the attention decoder over output characters."""

    def __init__(self, config):
        super().__init__()
        self.is_cuda = torch.cuda.is_available()
        self.Sy_intent = config.Sy_intent
        pretrained_model = PretrainedModel(config)
        if config.pretraining_type != 0:
            path = os.path.join(config.folder, "pretraining", "model_state.pth")
            # the file was written by rank 0 from cuda:0: deserialise onto the host, copy into this rank's
            # parameters (no allocation on a device the rank does not own)
            pretrained_model.load_state_dict(torch.load(path, map_location="cpu"))
        self.pretrained_model = pretrained_model
        self.unfreezing_type = config.unfreezing_type
        self.unfreezing_index = config.starting_unfreezing_index
        if config.pretraining_type != 0:
            self.freeze_all_layers()
        self.seq2seq = config.seq2seq
        out_dim = config.word_rnn_num_hidden[-1] * (2 if config.word_rnn_bidirectional else 1)
        if self.seq2seq:
            # character-level decoder instead of the fixed slots (reference models.py:718-725)
            self.intent_layers = []
            self.SOS = config.Sy_intent.index("<sos>")
            self.num_labels = len(config.Sy_intent)
            self.encoder = Seq2SeqEncoder(out_dim, config.num_intent_encoder_layers, config.intent_encoder_dim)
            self.decoder = Seq2SeqDecoder(self.num_labels, config.num_intent_decoder_layers, config.intent_encoder_dim,
                                          config.intent_decoder_dim, config.intent_decoder_key_dim,
                                          config.intent_decoder_value_dim, self.SOS)
            self._intent_stages = self.encoder._stages
            if self.is_cuda:
                self.cuda()
            return
        self.values_per_slot = config.values_per_slot
        self.num_values_total = sum(self.values_per_slot)
        intent_layers, self._intent_stages = [], []
        out_dim = _build_rnn_stack(intent_layers, self._intent_stages, "intent", out_dim,
                                   config.intent_rnn_num_hidden, config.intent_rnn_bidirectional,
                                   config.intent_rnn_drop, config.intent_downsample_type,
                                   config.intent_downsample_len)
        intent_layers.append(_named(torch.nn.Linear(out_dim, self.num_values_total), "final_classifier"))
        intent_layers.append(_named(FinalPool(), "final_pool"))
        self.intent_layers = torch.nn.ModuleList(intent_layers)
        if self.is_cuda:
            self.cuda()

    # -- freezing schedule (reference models.py:738-795) -----------------------------------------
    def freeze_all_layers(self):
        for layer in list(self.pretrained_model.phoneme_layers) + list(self.pretrained_model.word_layers):
            freeze_layer(layer)

    def print_frozen(self):
        for layer in list(self.pretrained_model.phoneme_layers) + list(self.pretrained_model.word_layers):
            if has_params(layer):
                print(layer.name + ": " + ("frozen" if is_frozen(layer) else "unfrozen"))

    def unfreeze_one_layer(self):
        """ULMFiT-style: each call unfreezes, from the top of the encoder down, every layer up to the
        `unfreezing_index`-th parametrised one, then advances the index.  Type 1 walks the word
        module only, type 2 continues into the phoneme module, type 0 does nothing."""
        if self.unfreezing_type not in (1, 2):
            return
        groups = [self.pretrained_model.word_layers]
        if self.unfreezing_type == 2:
            groups.append(self.pretrained_model.phoneme_layers)
        trainable_seen = 0
        for group in groups:
            for layer in reversed(list(group)):
                unfreeze_layer(layer)
                if has_params(layer):
                    trainable_seen += 1
                if trainable_seen == self.unfreezing_index:
                    self.unfreezing_index += 1
                    return

    # -- forward paths ---------------------------------------------------------------------------
    def _intent_features_tm(self, x):
        _DropoutState.current = next_rng_step()
        h = self.pretrained_model._features_tm(self.pretrained_model._to_device(x)[0])
        for st in self._intent_stages:
            h = st.run(h, self.training)
        return h                                                 # (T,B,C) time-major

    # -- split execution for the trainer's encoder look-ahead pipeline ---------------------------
    def frozen_prefix_len(self):
        """Leading encoder stages with no trainable parameter (the whole encoder for
        unfreezing_type 0): their outputs do not depend on earlier optimisation steps, so they may
        be computed ahead of time for upcoming batches."""
        return self.pretrained_model.frozen_prefix_len()

    def prefix_features(self, x, n_stages, rng_step, sub_batch=0):
        """Frozen stages [0, n_stages) without autograd, on the CURRENT stream.
        rng_step: int, or a 1-element int64 CUDA tensor holding step*16 (for hipGraph capture: the
        dropout kernels then read the step from device memory at replay time).
        sub_batch > 0: x is the concatenation of several upcoming batches of that size belonging to
        consecutive steps rng_step, rng_step+1, ...; each draws its own step's dropout masks."""
        with torch.no_grad():
            if torch.is_tensor(rng_step):
                _DropoutState.current_dev = rng_step
            else:
                _DropoutState.current = rng_step
            _DropoutState.sub_batch = sub_batch
            try:
                return self.pretrained_model.run_stages(self.pretrained_model._to_device(x)[0], 0, n_stages)
            finally:
                _DropoutState.current_dev = None
                _DropoutState.sub_batch = 0

    def forward_from(self, h, n_stages, y_intent, rng_step):
        """The rest of Model.forward given the output of stages [0, n_stages)."""
        pm = self.pretrained_model
        if torch.is_tensor(rng_step):              # hipGraph capture: step*16 lives on the device
            _DropoutState.current_dev = rng_step
        else:
            _DropoutState.current = rng_step
        try:
            h = pm.run_stages(h, n_stages, len(pm._stages()))
            drop = None
            for st in self._intent_stages:
                last = st is self._intent_stages[-1] and not self.seq2seq
                if last:
                    # the Dropout in front of the classifier is drawn inside the head kernels (same Philox stream, same
                    # bits as the stand-alone launch; two launches fewer per step) where ops.head_dropout_fusable allows
                    p, mask, seed, offset = _dropout_args(st.drop_name, st.site, st.p, self.training)
                    if _ops.head_dropout_fusable(self.intent_layers[-2].weight, p, mask, st.method, st.factor):
                        off, off_dev, sub = offset if isinstance(offset, tuple) else (offset, None, 0)
                        if sub == 0:
                            drop = (p, seed, off, off_dev)
                            h = st.gru.run_time_major(h, 0.0, None, 0, 0, st.method, st.factor, False)
                            continue
                h = st.run(h, self.training)
            if self.seq2seq:                       # teacher-forced decoder on the same dropout step
                loss_acc, _ = self.decoder.teacher_forced(h, y_intent.to(h.device))
        finally:
            _DropoutState.current_dev = None
        if self.seq2seq:
            # loss = -mean log p(y|x); the reference returns a host zero for the accuracy (models.py:825-828)
            self.last_loss_acc = loss_acc
            return loss_acc[0], torch.tensor([0.])
        cls = self.intent_layers[-2]
        loss, acc, _, _ = _ops.IntentHeadFn.apply(h, cls.weight, cls.bias, y_intent.to(h.device),
                                                  tuple(self.values_per_slot), drop)
        self.last_loss_acc = _ops.IntentHeadFn.last_loss_acc
        return loss, acc

    def forward(self, x, y_intent, *, rng_step=None, n_prefix=0):
        """x (B,T), y_intent (B,num_slots) -> (loss = sum of per-slot CE, acc = all slots right)
        (reference models.py:797-823); classifier, max over time, CE and accuracy are one fused op.
        seq2seq: y_intent (B,U,num_labels) one-hot -> (-mean log p(y|x), host zero) (models.py:825-828).
        Keyword-only extras (not in the reference) for training.Trainer: rng_step = the dropout-stream index of this
        forward (None = the next one; or a 1-element int64 CUDA tensor holding step * 16 for captured steps),
        n_prefix > 0: x is the output of the first n_prefix encoder stages (look-ahead pipeline)."""
        if n_prefix == 0:
            x = self.pretrained_model._to_device(x)[0]
        return self.forward_from(x, n_prefix, y_intent, next_rng_step() if rng_step is None else rng_step)

    def one_hot_to_string(self, input, S):
        """input (T, |S|) one-hot rows, S list of labels -> the string (reference models.py:731-737; the strips are
        character-set strips, as the reference applies them)."""
        return "".join([S[c] for c in input.max(dim=1)[1]]).lstrip("<sos>").rstrip("<eos>")

    def eval_group(self, xs, ys):
        """Evaluation (no dropout, no autograd) of several equally-shaped batches in ONE pass through the
        encoder and the intent GRU (concatenated along the batch axis: 8 x more recurrence workgroups at
        the same latency), then the per-batch loss/accuracy.  Returns [(loss, acc), ...] — the values
        forward() gives batch by batch."""
        assert not self.training
        pm = self.pretrained_model
        with torch.no_grad():
            dev = next(self.parameters()).device
            B = xs[0].shape[0]
            x_cat = torch.cat([x.to(dev, non_blocking=True) for x in xs]) if len(xs) > 1 else pm._to_device(xs[0])[0]
            h = pm.run_stages(x_cat, 0, len(pm._stages()))
            for st in self._intent_stages:
                h = st.run(h, False)
            cls = self.intent_layers[-2]
            out = []
            for k, y in enumerate(ys):
                hk = h[:, k * B:(k + 1) * B].contiguous() if len(xs) > 1 else h.contiguous()
                la, _, _, _, _ = _ops.cls_maxpool_ce_fwd(hk, cls.weight, cls.bias, y.to(dev).contiguous(),
                                                         tuple(self.values_per_slot), False)
                out.append((la[0], la[1]))
        return out

    def predict_intents(self, x):
        """-> (intent_logits (B, num_values_total), predicted_intent (B, num_slots)) (models.py:830-846)"""
        h = self._intent_features_tm(x).contiguous()
        if self.seq2seq:                                         # beam search, width 4 (models.py:848-851)
            return self.decoder.infer(h.transpose(0, 1), self.Sy_intent, B=4)
        cls = self.intent_layers[-2]
        _, logits, pred, _, _ = _ops.cls_maxpool_ce_fwd(h.detach(), cls.weight.detach(), cls.bias.detach(), None,
                                                        tuple(self.values_per_slot), False)
        return logits, pred

    def decode_intents(self, x):
        """-> list (batch) of lists (slots) of slot-value strings (reference models.py:853-865)."""
        _, pred = self.predict_intents(x)
        pred = pred.cpu()
        if self.seq2seq:                                         # best hypothesis of every utterance (models.py:866-874)
            return [self.one_hot_to_string(pred[0, i], self.Sy_intent) for i in range(pred.shape[1])]
        inverse = [{idx: value for value, idx in self.Sy_intent[slot].items()} for slot in self.Sy_intent]
        return [[inverse[s][int(row[s])] for s in range(len(inverse)) if int(row[s]) in inverse[s]]
                for row in pred]
