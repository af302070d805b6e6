"""The GUARDED fast arithmetic of the frozen stages (SLU_FROZEN_MATH=auto, opt-in since round 5 — the default is bf16x3
everywhere: f16x2 under a range guard, bf16x3 otherwise — slu_hip/guard.py) must have fp32's DOMAIN, like the reference's plain fp32 ATen kernels (models.py:108, :200, :232):
whatever the scale of the waveforms or of the checkpoint, the encoder's output stays within the parity bound of the CPU
ORACLE (not merely of the package's own exact kernels), through every entry point that evaluates frozen stages —
compute_features / predict_intents (eager, guarded), the look-ahead training loop (guard read when a slot is consumed)."""
import contextlib
import os
import sys

import pytest
import torch

from oracle import slu_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-slu_amd")
sys.path.insert(0, PKG)

TOL = 1e-4          # BASELINE.json north_star: max-abs deviation from the CPU reference path, fp32


def _cfg(tmp_path):
    import data
    cfg = O.OracleConfig(pretraining_type=2)                 # experiments/no_unfreezing.cfg
    cfg.folder = str(tmp_path)
    cfg.training_lr = 0.001
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    os.makedirs(tmp_path / "training", exist_ok=True)
    return cfg


def _model(tmp_path, cfg, scale_weights=None):
    import models
    torch.manual_seed(1)
    pre = O.init_pretrained_state_dict(cfg)
    if scale_weights is not None:
        # a "scaled checkpoint": every fp32 weight and bias of the encoder (the float64 Sinc cut-offs are frequencies)
        pre = {k: (v * scale_weights if v.dtype == torch.float32 else v) for k, v in pre.items()}
    torch.save(pre, tmp_path / "pretraining" / "model_state.pth")
    torch.manual_seed(2)
    return models.Model(cfg), pre


@pytest.mark.parametrize("amp", [1e-6, 1e-5, 1e-3, 0.1, 30.0, 1e3, 32768.0])
def test_default_arithmetic_over_the_input_dynamic_range_vs_oracle(tmp_path, monkeypatch, amp):
    """Waveform amplitudes from 1e-6 (every sample far below fp16's smallest normal) to 32768 (un-normalised int16
    audio: the Sinc layer's output reaches 1e6, beyond fp16): encoder features in the GUARDED mode (SLU_FROZEN_MATH=auto) against the CPU
    oracle, eval mode, full-size architecture, 8 x 1 s.  Which arithmetic ran is asserted too: f16x2 for ordinary audio,
    bf16x3 after a quiet-input trip (no pin) or an overflow trip (pinned) — and SLU_FROZEN_MATH=f16x2, the UNGUARDED form,
    does break at 32768, which is what the guard is for."""
    import models
    monkeypatch.setenv("SLU_FROZEN_MATH", "auto")
    cfg = _cfg(tmp_path)
    model, pre = _model(tmp_path, cfg)
    model.eval()
    pm = model.pretrained_model
    g = torch.Generator().manual_seed(5)
    x = amp * torch.randn(8, 16000, generator=g)
    ref = O.encoder_stages(pre, x, cfg, None, explicit_gru=False)["features"]        # (8, T', 256)
    with torch.no_grad():
        got = pm.compute_features(x).float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    guard = pm.range_guard()
    pinned = getattr(pm, "_f16x2_pin", None)
    print("amplitude %g: max-abs deviation from the ORACLE %.3e; guard trips %d, pinned: %s" % (amp, err, guard.trips, pinned))
    if amp < 1e3:
        assert err <= TOL
    else:
        # pre-activations of 1e5 and more saturate the gates: two correct fp32 evaluations then differ wherever a
        # pre-activation lands within its own round-off of zero — the bound is what the package's EXACT fp32 kernels achieve
        _assert_as_close_as_exact_fp32(monkeypatch, pm, x, ref, got)
    from slu_hip import guard as G
    peak = x.abs().max().item()
    if amp in (0.1, 30.0):
        assert guard.trips == 0 and pinned is None and models.guarded_frozen_nsplit(model) == 2
    elif 0.0 < peak < G.QUIET_INPUT:
        assert guard.trips == 1 and pinned is None            # quiet input: this evaluation re-ran on bf16x3, no pin
    elif peak >= G.F16X2_LIMIT:
        assert guard.trips == 1 and pinned is not None and models.guarded_frozen_nsplit(model) == 3
    if pinned is not None:
        # pinned: the next evaluation does not even try f16x2
        with torch.no_grad():
            again = pm.compute_features(x).float().cpu()
        assert guard.trips == 1 and torch.equal(again, got)
    assert amp > 1e-4 or guard.trips == 1                     # 1e-6 / 1e-5: quiet by construction
    assert amp < 3e4 or pinned is not None                    # int16 scale: overflow by construction
    # the same through the inference entry point of the reference API (models.py:830-846)
    logits, pred = model.predict_intents(x)
    sd = {("pretrained_model." + k): v for k, v in pre.items()}
    sd.update({k: v.detach().cpu() for k, v in model.state_dict().items() if not k.startswith("pretrained_model.")})
    want = O.intent_logits(sd, ref, cfg, None, explicit_gru=False)
    want = want[0] if isinstance(want, tuple) else want
    if amp < 1e3:
        assert (logits.cpu() - want).abs().max().item() <= TOL
    if amp == 32768.0:
        monkeypatch.setenv("SLU_FROZEN_MATH", "f16x2")        # unguarded: the caller vouches for the range — wrongly here
        with torch.no_grad():
            bad = pm.compute_features(x).float().cpu()
        assert (not torch.isfinite(bad).all()) or (bad - ref).abs().max().item() > TOL


def _assert_as_close_as_exact_fp32(monkeypatch, pm, x, ref, got, slack=0.002):
    """For ill-conditioned inputs (saturated gates): the default arithmetic must deviate from the oracle on no more elements
    than the package's exact fp32 kernels (SLU_FROZEN_MATH=fp32) do, and — when the guard switched the model to bf16x3 —
    equal the explicit bf16x3 mode bit for bit."""
    switched = getattr(pm, "_f16x2_pin", None) is not None or not pm.f16x2_allowed()
    monkeypatch.setenv("SLU_FROZEN_MATH", "fp32")
    with torch.no_grad():
        exact = pm.compute_features(x).float().cpu()
    monkeypatch.setenv("SLU_FROZEN_MATH", "bf16x3")
    with torch.no_grad():
        b3 = pm.compute_features(x).float().cpu()
    monkeypatch.setenv("SLU_FROZEN_MATH", "auto")
    frac = lambda a: ((a - ref).abs() > TOL).float().mean().item()
    print("   fraction of features further than 1e-4 from the oracle: default %.4f, exact fp32 kernels %.4f" % (frac(got), frac(exact)))
    assert not switched or torch.equal(b3, got)
    assert frac(got) <= 2.0 * frac(exact) + slack


@pytest.mark.parametrize("case", ["conv1_x1e5_conv2_x1e-5", "all_x1e-4", "all_x1e3"])
def test_default_arithmetic_with_a_scaled_checkpoint_vs_oracle(tmp_path, monkeypatch, case):
    """Checkpoints outside fp16's comfortable range, features against the ORACLE on the same weights:
      * conv1 x 1e5, conv2 x 1e-5 — the network computes the same function up to round-off, but the activation between
        the two layers reaches 1e5: only the guard's activation words can see that (the weights' own maxima, 5e3 and 6e-7,
        also fail the pack-time check, so that check is switched off for this case to exercise the activation path);
      * every weight and bias x 1e-4 — all entries below fp16's smallest normal: the pack-time weight check sends the model
        to bf16x3 before anything runs;
      * every weight and bias x 1e3 — pre-activations of 1e6 and more saturate every gate; two correct fp32 evaluations
        then differ wherever a pre-activation lands within its own round-off of zero (the recurrence amplifies that to
        sign flips), so the bound is relative: no more features off the oracle than with the package's exact fp32
        kernels, and agreement with the explicit bf16x3 mode bit for bit (the guard switched)."""
    import models
    from slu_hip import guard as G
    monkeypatch.setenv("SLU_FROZEN_MATH", "auto")
    cfg = _cfg(tmp_path)
    if case == "conv1_x1e5_conv2_x1e-5":
        model, pre = _model(tmp_path, cfg)
        idx = O.phoneme_layer_index(cfg)
        with torch.no_grad():
            for name, f in (("conv1", 1e5), ("conv2", 1e-5)):
                li = idx[name]
                for part in ("weight", "bias"):
                    key = "phoneme_layers.%d.%s" % (li, part)
                    # conv2's bias keeps its scale: (1e-5 W2) * (1e5 a1) + b2 = W2 a1 + b2
                    g = f if not (name == "conv2" and part == "bias") else 1.0
                    pre[key] = pre[key] * g
                    getattr(model.pretrained_model.phoneme_layers[li], part).mul_(g)
        monkeypatch.setattr(G, "WEIGHT_MIN", 0.0)
        monkeypatch.setattr(G, "weights_in_range", lambda tensors: (True, None))
    else:
        model, pre = _model(tmp_path, cfg, scale_weights=1e-4 if case == "all_x1e-4" else 1e3)
    model.eval()
    pm = model.pretrained_model
    g = torch.Generator().manual_seed(6)
    x = 0.1 * torch.randn(8, 16000, generator=g)
    ref = O.encoder_stages(pre, x, cfg, None, explicit_gru=False)["features"]
    with torch.no_grad():
        got = pm.compute_features(x).float().cpu()
    dev = (got - ref).abs()
    print("checkpoint %s: max-abs deviation from the ORACLE %.3e (99.9 %% quantile %.3e); guard trips %d, pinned: %s"
          % (case, dev.max().item(), dev.flatten().kthvalue(int(0.999 * dev.numel())).values.item(),
             pm.range_guard().trips, getattr(pm, "_f16x2_pin", None)))
    assert torch.isfinite(got).all()
    if case == "all_x1e3":
        _assert_as_close_as_exact_fp32(monkeypatch, pm, x, ref, got)
    else:
        assert dev.max().item() <= TOL
    assert models.guarded_frozen_nsplit(model) == 3                   # weights out of range, or pinned by the trip
    if case == "all_x1e-4":
        assert pm.range_guard().trips == 0 and getattr(pm, "_f16x2_pin", None) is None     # the weight check decided
    if case == "conv1_x1e5_conv2_x1e-5":
        assert pm.range_guard().trips == 1 and getattr(pm, "_f16x2_pin", None) is not None


def test_lookahead_loop_reads_the_guard_before_it_uses_a_super_batch(tmp_path, monkeypatch):
    """Training on un-normalised (int16-scale) waveforms in the default mode: the first look-ahead super-batch trips its
    slot's range guard, is re-run on bf16x3 before any of its steps, and the model is pinned — losses and final
    parameters equal the eager sequential loop under SLU_FROZEN_MATH=bf16x3 bit for bit, and stay finite."""
    import data
    import models
    import training
    cfg = _cfg(tmp_path)
    ds = data.SyntheticSLUDataset(3, 8, 16000, cfg.values_per_slot, seed=9)
    batches = [((x * 3e4 / 0.1).cuda(), y.cuda()) for x, y in ds.batches]      # the synthetic sets draw 0.1 * randn
    loader = [batches[i % 3] for i in range(12)]

    def run(lookahead, graphs, math):
        monkeypatch.setenv("SLU_LOOKAHEAD", lookahead)
        monkeypatch.setenv("SLU_GRAPHS", graphs)
        if math is None:
            monkeypatch.setenv("SLU_FROZEN_MATH", "auto")
        else:
            monkeypatch.setenv("SLU_FROZEN_MATH", math)
        model, _ = _model(tmp_path, cfg)
        models.set_dropout_seed(4321)
        trainer = training.Trainer(model, cfg)
        model.train()
        losses = []
        with contextlib.closing(trainer._iterate(loader, True, False, accumulate=True)) as it:
            for v, _ in it:
                losses.append(float(v[0]))
        torch.cuda.synchronize()
        return trainer, losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    ref_tr, ref_losses, ref_sd = run("0", "0", "bf16x3")
    tr, losses, sd = run("4", "1", None)
    pm = tr.model.pretrained_model
    assert all(abs(l) < 1e3 for l in losses) and all(l == l for l in losses)
    assert sum(slot.guard.trips for slot in tr._slots) >= 1
    assert getattr(pm, "_f16x2_pin", None) is not None and not pm.f16x2_allowed()
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k
