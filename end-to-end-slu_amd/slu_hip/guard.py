"""Range guard of the f16x2 split scheme (csrc/slu_bf16.h).

f16x2 writes an fp32 operand as two fp16 terms: 22 bits of significand at 3/16 of the fp32-MFMA cycles, but fp16's
EXPONENT range — an operand of 65504 or more becomes infinite, and a tensor whose entries all sit below fp16's smallest
normal (6.1e-5) keeps 11 bits.  The reference computes these stages in plain fp32 ATen kernels (models.py:108, :200,
:232), whose domain has neither limit.  So the default arithmetic of the frozen stages (SLU_FROZEN_MATH=auto) is:
f16x2 WHERE A GUARD WATCHES IT, bf16x3 (three bf16 terms: fp32's 24 bits and fp32's exponent range) everywhere else.

What the guard watches:
  * activations — every split-precision convolution launch raises a device word to the IEEE bit pattern of the largest
    |v| it splits (its input window; its plane output), slu_wconv_fwd_bf16(absmax_word).  The recurrences need no word:
    their operands are h (|h| <= 1), dropout-scaled h (<= 1 / (1 - p)) and weights.  The words travel to pinned host
    memory at the end of the guarded evaluation (inside the captured hipGraph of a look-ahead super-batch) and are read
    when the result is consumed: a pattern at or above 65504.0f (NaN and infinity rank higher) re-runs the evaluation on
    bf16x3 and pins the model to it; a first-stage input (waveform) whose maximum is below 2^-8 — samples carried by
    the lo term alone — re-runs on bf16x3 without pinning;
  * weights — once per weight version, slu_absmax_multi over the frozen stages' filters and GRU matrices: a tensor
    whose largest entry is >= 65504, non-finite or below 2^-5 (entries under 6.1e-5 then lose more than an fp32
    accumulation chain does) pins the model to bf16x3 before anything runs.

An explicit SLU_FROZEN_MATH=f16x2 runs the scheme unguarded (the caller vouches for the range), bf16x3 / fp32 as before.
"""
import struct

import torch

from . import lib as _lib
from . import ops as _ops

N_WORDS = 8
F16X2_LIMIT = 65504.0                 # operands must stay below fp16's largest finite value
QUIET_INPUT = 2.0 ** -8               # first-stage input: below this maximum the samples live in the lo term alone
WEIGHT_MIN = 2.0 ** -5                # a weight tensor's largest entry: below it the sub-normal-hi entries cost accuracy


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def _value(bits):
    return struct.unpack("<f", struct.pack("<I", bits & 0xFFFFFFFF))[0]


LIMIT_BITS = _bits(F16X2_LIMIT)


class RangeGuard:
    """Range words of one guarded evaluation of frozen stages: word k belongs to convolution stage k (the last word
    collects every further stage)."""

    def __init__(self, device):
        self.device = device
        self.words = torch.zeros(N_WORDS, dtype=torch.int32, device=device)
        self.host = torch.zeros(N_WORDS, dtype=torch.int32).pin_memory()
        self.trips = 0

    def arm(self):
        self.words.zero_()

    def word(self, k):
        """Device address of the word of convolution stage k (for slu_wconv_fwd_bf16's absmax_word)."""
        return self.words.data_ptr() + 4 * min(int(k), N_WORDS - 1)

    def collect(self):
        """Enqueue the device -> pinned host copy of the words on the current stream (capturable)."""
        self.host.copy_(self.words, non_blocking=True)

    def verdict(self):
        """After the stream that ran collect() has been synchronised: (overflow, quiet, [max |v| per word])."""
        pats = [int(w) & 0xFFFFFFFF for w in self.host.tolist()]
        overflow = any(p >= LIMIT_BITS for p in pats)
        first = _value(pats[0]) if pats[0] < LIMIT_BITS else float("inf")
        quiet = 0.0 < first < QUIET_INPUT
        if overflow or quiet:
            self.trips += 1
        return overflow, quiet, [_value(p) if p < 0x7F800000 else float("inf") for p in pats]


def weights_in_range(tensors):
    """tensors: fp32 CUDA tensors (the filters / matrices a model's frozen stages split).  One launch + one read-back:
    -> (ok, offending maximum or None).  ok: every tensor's largest |entry| is finite, < 65504 and >= 2^-5."""
    if not tensors:
        return True, None
    import ctypes
    L = _lib.load()
    dev = tensors[0].device
    ok, worst = True, None
    mx = L.slu_multi_max()
    for i in range(0, len(tensors), mx):
        chunk = [t.detach().contiguous() for t in tensors[i:i + mx]]
        words = torch.zeros(len(chunk), dtype=torch.int32, device=dev)
        ptrs = (ctypes.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        numel = (ctypes.c_int64 * len(chunk))(*[t.numel() for t in chunk])
        _lib.check(L.slu_absmax_multi(ptrs, numel, len(chunk), words.data_ptr(), _ops._stream()), "slu_absmax_multi")
        for w in words.tolist():
            p = int(w) & 0xFFFFFFFF
            v = _value(p) if p < 0x7F800000 else float("inf")
            if p >= LIMIT_BITS or v < WEIGHT_MIN:
                ok, worst = False, v
    return ok, worst
