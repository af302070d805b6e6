"""GPU parity of the EXACT path bench.py times (BASELINE.json configs[3] at full size: no_unfreezing
architecture, H = 128, B = 64, 3 s): look-ahead super-batches of 20 batches (1280 sequences on the 160-CU partition, split-precision
(f16x2) MFMA input projections and 16-sequence recurrence kernels for the frozen layers, sub-batch Philox streams) replayed from captured hipGraphs + the captured training
step, against (1) the plain sequential eager loop, bit for bit, and (2) the CPU oracle (<= 1e-4).

Chain proven here:  oracle == HIP kernels at super-batch size (mask-in)  and  sequential eager ==
sequential captured == pipelined + captured (Philox), so the benchmarked path computes what the
reference computes (reference models.py:797-823, training.py:84-119)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import slu_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-slu_amd")
sys.path.insert(0, PKG)


def _full_cfg(tmp_path, **kw):
    import data
    cfg = O.OracleConfig(pretraining_type=2, **kw)           # defaults = experiments/no_unfreezing.cfg
    cfg.folder = str(tmp_path)
    cfg.training_lr = 0.001
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    os.makedirs(tmp_path / "training", exist_ok=True)
    return cfg


def _run_training(cfg, loader, monkeypatch, lookahead, graphs, n_steps, math=None):
    """math: SLU_FROZEN_MATH for this run (None = the default: bf16x3, reference-width; "auto" = guarded f16x2 where a
    guard can act, bf16x3 where none can)."""
    import models
    import training
    monkeypatch.setenv("SLU_LOOKAHEAD", lookahead)
    monkeypatch.setenv("SLU_GRAPHS", graphs)
    if math is None:
        monkeypatch.delenv("SLU_FROZEN_MATH", raising=False)
    else:
        monkeypatch.setenv("SLU_FROZEN_MATH", math)
    torch.manual_seed(2)
    model = models.Model(cfg)
    models.set_dropout_seed(1234)
    trainer = training.Trainer(model, cfg)
    model.train()
    vals = []
    import contextlib
    # accumulate=True as bench.run_steps / Trainer.train() call it: the epoch statistics are kept on the device by
    # the loop (inside the captured step's intent-head launch)
    with contextlib.closing(trainer._iterate(loader, True, False, accumulate=True)) as it:
        for v, _ in it:
            # captured steps return their static (2,) metrics buffer: snapshot it on the current stream
            vals.append(v.clone() if torch.is_tensor(v) else torch.stack([v[0].detach(), v[1].detach()]))
    torch.cuda.synchronize()
    losses = [float(v[0]) for v in vals]
    assert len(losses) == n_steps
    # the device-side epoch sums = sum over steps of batch_size * (loss, acc), accumulated in float64 in step order
    want = torch.zeros(2, dtype=torch.float64)
    for v, (x, _) in zip(vals, loader):
        want += v.detach().cpu().double()[:2] * len(x)
    assert torch.equal(trainer.epoch_sums[:2].cpu(), want)
    return trainer, losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("math", [None, "auto"])
def test_benchmarked_pipeline_equals_sequential_eager_at_full_size(tmp_path, monkeypatch, math):
    """120 steps of B = 64 x 3 s: six 20-batch super-batches, three per look-ahead slot: each slot captures its shape
    on the second appearance and REPLAYS it on the third; the training step is captured after three eager steps.
    Per-step losses and final parameters must be bit-equal to the eager sequential loop.  math None = the default
    arithmetic of frozen stages (bf16x3: what bench.py's `value` runs), "auto" = the opt-in guarded f16x2."""
    import data
    import models
    import training

    def run(*a):
        return _run_training(*a, math=math)
    cfg = _full_cfg(tmp_path)
    torch.manual_seed(1)
    torch.save(O.init_pretrained_state_dict(cfg), tmp_path / "pretraining" / "model_state.pth")
    ds = data.SyntheticSLUDataset(4, 64, 48000, cfg.values_per_slot, seed=1234)
    n_steps = 120
    # inputs resident in HBM, as bench.py holds them: the look-ahead super-batches then read the batches in place
    # through the row-pointer table (no concatenation copy)
    dev_batches = [tuple(t.cuda() for t in b) for b in ds.batches]
    loader = [dev_batches[i % 4] for i in range(n_steps)]
    ref_tr, ref_losses, ref_sd = run(cfg, loader, monkeypatch, "0", "0", n_steps)
    assert ref_tr.graph_stats() == {"step_graphs": 0, "prefix_graphs": 0, "capture_failures": 0}
    assert len(set(ref_losses)) == n_steps                      # dropout and the optimiser really moved

    # sequential steps, each captured as a hipGraph (what --workload unfreeze_all / SLU_LOOKAHEAD=0 run).  The frozen stages
    # sit INSIDE the captured step there: it is captured as forward + backward | range check | Adam, in the same guarded
    # f16x2 arithmetic as the eager loop and the look-ahead pipeline
    tr, losses, sd = run(cfg, loader, monkeypatch, "0", "1", n_steps)
    assert tr.graph_stats() == {"step_graphs": 1, "prefix_graphs": 0, "capture_failures": 0}
    if math == "auto":
        assert next(iter(tr._step_graphs.values())).guard is not None and tr.model.pretrained_model.range_guard().trips == 0
    else:
        assert next(iter(tr._step_graphs.values())).guard is None       # bf16x3 needs no guard: one graph per step
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k

    # the bench.py default: automatic look-ahead width (20 batches = 1280 sequences on 160 CUs) + graphs
    tr, losses, sd = run(cfg, loader, monkeypatch, "auto", "1", n_steps)
    from slu_hip import pipeline as _pl
    width = training._lookahead_width(-1, 64)
    assert width == 8 * (_pl.n_compute_units(0) - _pl.cu_split()) // 64       # one 16-sequence recurrence workgroup per CU and direction
    stats = tr.graph_stats()
    assert stats["step_graphs"] == 1 and stats["capture_failures"] == 0
    assert all(len([g for g in slot.graphs.values() if g is not None]) >= 1 for slot in tr._slots)
    assert stats["prefix_graphs"] >= 2
    # every slot replayed its captured graph at least once (seen >= 3 for the full-width key)
    if n_steps >= 6 * width:
        assert all(max(slot.seen.values()) >= 3 for slot in tr._slots)
    from slu_hip import ops as _ops
    if models.guarded_frozen_nsplit(tr.model):      # the split-precision first stage reads the batches in place (row-pointer table)
        assert all(isinstance(g[1], _ops.RowTable) for slot in tr._slots for g in slot.graphs.values() if g is not None)
    if math == "auto":
        # the guarded mode ran f16x2 under the slots' range guards, and nothing tripped them
        assert models.guarded_frozen_nsplit(tr.model) == 2 and getattr(tr.model.pretrained_model, "_f16x2_pin", None) is None
        assert all(slot.guard.trips == 0 for slot in tr._slots)
    else:
        assert models.guarded_frozen_nsplit(tr.model) == 3             # reference-width by default
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k


def test_first_super_batch_on_the_whole_chip_replayed_over_repeated_runs_is_bit_identical(tmp_path, monkeypatch):
    """bench.py's K = 20 command, repeated: every run's FIRST super-batch (13 batches) goes to the unmasked stream, is
    captured there on the second run and replayed from the third on, and the host waits for it before it enqueues the
    steps (training._iterate).  Five runs of 20 steps on one trainer against the same 100 steps of the sequential eager
    loop: per-step losses and final parameters bit-equal."""
    import contextlib
    import data
    import models
    import training
    cfg = _full_cfg(tmp_path)
    torch.manual_seed(1)
    torch.save(O.init_pretrained_state_dict(cfg), tmp_path / "pretraining" / "model_state.pth")
    ds = data.SyntheticSLUDataset(4, 64, 48000, cfg.values_per_slot, seed=1234)
    dev_batches = [tuple(t.cuda() for t in b) for b in ds.batches]
    runs, per_run = 5, 20

    def go(lookahead, graphs):
        monkeypatch.setenv("SLU_LOOKAHEAD", lookahead)
        monkeypatch.setenv("SLU_GRAPHS", graphs)
        monkeypatch.delenv("SLU_FROZEN_MATH", raising=False)
        torch.manual_seed(2)
        model = models.Model(cfg)
        models.set_dropout_seed(1234)
        trainer = training.Trainer(model, cfg)
        model.train()
        losses = []
        for r in range(runs):
            loader = [dev_batches[(r * per_run + i) % 4] for i in range(per_run)]
            with contextlib.closing(trainer._iterate(loader, True, False, accumulate=True)) as it:
                for v, _ in it:
                    losses.append(v.clone() if torch.is_tensor(v) else torch.stack([v[0].detach(), v[1].detach()]))
            torch.cuda.synchronize()
        return trainer, [float(v[0]) for v in losses], {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    _, ref_losses, ref_sd = go("0", "0")
    tr, losses, sd = go("auto", "1")
    whole = [(k, v) for k, v in tr._slots[0].graphs.items() if k[-1]]
    assert whole and all(v is not None for _, v in whole), "the whole-chip variant of the first super-batch was not captured"
    assert max(tr._slots[0].seen[k] for k, _ in whole) == runs          # captured on run 2, replayed on runs 3 - 5
    assert whole[0][0][0] == training._ramp_plan(per_run, 20, 2)[0][0] == 13          # ceil(5 + 0.4 * 20) batches
    assert losses == ref_losses
    for k, v in ref_sd.items():
        assert torch.equal(v, sd[k]), k


@pytest.mark.parametrize("n_utt,math", [(1024, "f16x2"), (1280, "f16x2"), (768, "bf16x3"), (1280, "bf16x3")])
def test_super_batch_prefix_vs_oracle_with_injected_masks(tmp_path, monkeypatch, n_utt, math):
    """One look-ahead super-batch through the frozen encoder with the oracle's dropout masks, against the CPU oracle's
    encoder (models.py:349-361), features within 1e-4.  1024 = 16 x 64 utterances of 3 s is exactly what bench.py's
    default launches: split-precision convolutions, the first GRU layer's recurrence with the fused K = 60 input
    projection, the 96-row panel GEMM (M = 150 x 1024 = 153 600 rows >= 131 072, K = 256), the tiled GEMM for the shorter
    layers and the 16-sequence split-precision (f16x2) recurrence on 64 tiles x 2 directions; 1280 = the 20-batch
    super-batch of the 96 + 160 CU partition (rounds 1-2); 768 = a 12-batch super-batch (all launches below the panel
    threshold), on the bf16x3 scheme (whose first layer takes the row-panel GEMM for K = 60 + the plain recurrence); 1280 on
    bf16x3 = THE DEFAULT arithmetic at the default width (round 6: it had been compared with the oracle at 768 only)."""
    import models
    monkeypatch.setenv("SLU_FROZEN_MATH", math)
    cfg = _full_cfg(tmp_path)
    torch.manual_seed(1)
    pre = O.init_pretrained_state_dict(cfg)
    torch.save(pre, tmp_path / "pretraining" / "model_state.pth")
    torch.manual_seed(2)
    model = models.Model(cfg)
    model.train()
    g = torch.Generator().manual_seed(7)
    x = 0.1 * torch.randn(n_utt, 48000, generator=g)
    masks = O.draw_dropout_masks(cfg, x, seed=11, include_intent=False)
    models.set_dropout_masks({k: v.cuda() for k, v in masks.items()})
    try:
        n = model.frozen_prefix_len()
        assert n == 7
        feats = model.prefix_features(x.cuda(), n, 1)              # (19, n_utt, 256) time-major
        torch.cuda.synchronize()
    finally:
        models.set_dropout_masks(None)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = O.encoder_stages(pre, x, cfg, masks, explicit_gru=False)["features"]       # (n_utt, 19, 256)
    err = (feats.transpose(0, 1).cpu() - ref).abs().max().item()
    print("super-batch (%d x 3 s, %s) encoder features max-abs deviation vs oracle: %.3e" % (n_utt, math, err))
    assert err <= 1e-4


@pytest.mark.parametrize("tile", ["f16x2", "bf16x3", "4", "16"])
def test_recurrence_at_super_batch_size_vs_oracle(tile, monkeypatch):
    """The GRU layer at B = 768, T = 300, I = 60, H = 128, both directions (phone_rnn0 of a super-batch) against
    the oracle's GRU (torch.nn.GRU semantics, models.py:232): the split-precision path the bench runs for
    frozen layers (slu_gemm_bf16 + slu_gru_seq_fwd_bf16: two fp16 terms — the default — or three bf16 terms) and the
    exact-fp32 kernels forced onto 4- and 16-sequence workgroups."""
    from slu_hip import ops
    ns = {"f16x2": 2, "bf16x3": 3}.get(tile, 0)
    if not ns:
        monkeypatch.setenv("SLU_GRU_TILE", tile)
    T, B, I, H = 300, 768, 60, 128
    torch.manual_seed(3)
    p = {}
    for sfx in ("", "_reverse"):
        p["weight_ih_l0" + sfx] = (torch.rand(3 * H, I) * 2 - 1) / H ** 0.5
        p["weight_hh_l0" + sfx] = (torch.rand(3 * H, H) * 2 - 1) / H ** 0.5
        p["bias_ih_l0" + sfx] = (torch.rand(3 * H) * 2 - 1) / H ** 0.5
        p["bias_hh_l0" + sfx] = (torch.rand(3 * H) * 2 - 1) / H ** 0.5
    x = torch.randn(B, T, I)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    with torch.no_grad():
        ref = O.gru_layer(x, p, bidirectional=True, explicit=False)                 # (B, T, 2H)
    d = {k: v.cuda() for k, v in p.items()}
    xt = x.transpose(0, 1).contiguous().cuda()
    w_ih = torch.cat([d["weight_ih_l0"], d["weight_ih_l0_reverse"]])
    b_ih = torch.cat([d["bias_ih_l0"], d["bias_ih_l0_reverse"]])
    if ns:
        gx = ops.gemm_bf16(ops.split_bf16(xt.view(T * B, I), ns), ops.gemm_bf16_pack(w_ih, ns), b_ih, 6 * H, I)
        out, _ = ops.gru_seq_fwd_bf16(gx, d["weight_hh_l0"], d["weight_hh_l0_reverse"], d["bias_hh_l0"],
                                      d["bias_hh_l0_reverse"], T, B, H, 2, ns)
    else:
        gx = ops.gemm(xt.view(T * B, I), w_ih.t(), b_ih)
        out, _ = ops.gru_seq_fwd(gx, d["weight_hh_l0"], d["weight_hh_l0_reverse"], d["bias_hh_l0"],
                                 d["bias_hh_l0_reverse"], T, B, H, 2, False)
    torch.cuda.synchronize()
    err = (out.transpose(0, 1).cpu() - ref).abs().max().item()
    print("GRU B=768 T=300 tile %s: max-abs deviation vs oracle %.3e" % (tile, err))
    assert err <= 1e-4


def test_captured_asr_pretraining_step_equals_eager(tmp_path, monkeypatch):
    """ASR pre-training (BASELINE configs[2] shape family: every layer trainable, both CE heads): the
    hipGraph-captured step loop gives bit-identical losses and parameters to the eager loop."""
    import contextlib
    import data
    import models
    import training
    cfg = O.OracleConfig(cnn_N_filt=[16, 12, 12], cnn_len_filt=[101, 5, 5], cnn_stride=[20, 1, 1],
                         phone_rnn_num_hidden=[32, 32], word_rnn_num_hidden=[32, 32],
                         intent_rnn_num_hidden=[32], vocabulary_size=60, num_phonemes=20, pretraining_type=2)
    cfg.folder = str(tmp_path)
    cfg.pretraining_lr = 0.002
    cfg.phone_downsample_factor, cfg.word_downsample_factor = 20 * 2 * 4, 20 * 2 * 16
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    ds = data.SyntheticASRDataset(3, 8, 8000, cfg, seed=3)
    loader = [ds.batches[i % 3] for i in range(10)]
    results = {}
    for graphs in ("0", "1"):
        monkeypatch.setenv("SLU_GRAPHS", graphs)
        torch.manual_seed(9)
        pm = models.PretrainedModel(cfg)
        models.set_dropout_seed(5)
        trainer = training.Trainer(pm, cfg)
        pm.train()
        vals = []
        with contextlib.closing(trainer._iterate(loader, True, True)) as it:
            for v, _ in it:
                vals.append(v.clone() if torch.is_tensor(v) else
                            torch.stack([t.detach().float().reshape(()).cuda() for t in v]))
        torch.cuda.synchronize()
        assert trainer.graph_stats()["step_graphs"] == (1 if graphs == "1" else 0)
        assert trainer.graph_stats()["capture_failures"] == 0
        results[graphs] = (torch.stack(vals).cpu(), {k: v.detach().cpu().clone() for k, v in pm.state_dict().items()})
    assert torch.equal(results["0"][0], results["1"][0])
    assert len(set(results["0"][0][:, 0].tolist())) == 10
    for k, v in results["0"][1].items():
        assert torch.equal(v, results["1"][1][k]), k


@pytest.mark.parametrize("scale", [1e-3, 0.1, 30.0])
def test_split_schemes_hold_fp32_class_over_the_input_dynamic_range(tmp_path, monkeypatch, scale):
    """f16x2 (the default of the frozen stages) has fp16's exponent range: quiet recordings (amplitude 1e-3: most
    samples below fp16's smallest normal in the first convolution) and loud ones (30: far beyond [-1, 1]) must still give
    encoder features within fp32 round-off of the exact-fp32 kernels, like bf16x3 (no range limit).  Eval mode (no
    dropout), full-size architecture, 16 x 3 s."""
    import models
    cfg = _full_cfg(tmp_path)
    torch.manual_seed(1)
    pre = O.init_pretrained_state_dict(cfg)
    torch.save(pre, tmp_path / "pretraining" / "model_state.pth")
    torch.manual_seed(2)
    model = models.Model(cfg)
    model.eval()
    g = torch.Generator().manual_seed(5)
    x = (scale * torch.randn(16, 48000, generator=g)).cuda()
    n = model.frozen_prefix_len()
    feats = {}
    with torch.no_grad():
        for math in ("fp32", "f16x2", "bf16x3"):
            monkeypatch.setenv("SLU_FROZEN_MATH", math)
            feats[math] = model.prefix_features(x, n, 1).clone()
    torch.cuda.synchronize()
    ref = feats["fp32"]
    span = ref.abs().max().item()
    for math in ("f16x2", "bf16x3"):
        assert torch.isfinite(feats[math]).all()
        err = (feats[math] - ref).abs().max().item()
        print("input amplitude %g, %s: max-abs deviation from the exact-fp32 kernels %.3e (feature range %.3f)" % (scale, math, err, span))
        assert err <= 2e-5 * max(span, 1.0)


def test_weight_gradient_branches_of_fully_trainable_loops(tmp_path, monkeypatch):
    """Fully trainable loops (round 5): the batched weight-gradient launch of every long GRU layer runs on an auxiliary stream
    / graph branch with a workgroup budget.  SLU_WGRAD_BRANCH=layer (default) joins it at the end of its layer's backward
    (beside the data-gradient GEMM); =pass leaves the very same launches open until ONE join after the whole backward pass
    (beside the BPTT of the layer below).  If the open form had a hazard — a gradient cloned or packed before its branch wrote
    it, an operand's memory handed out again too early — the two would differ: they must be BIT-IDENTICAL, losses and final
    parameters, over eager steps, the capture and replays.  SLU_WGRAD_BRANCH=0 (in line, no budget: another split count, i.e.
    another summation order) must agree to fp32 round-off."""
    import data
    import models
    import training
    cfg = _full_cfg(tmp_path)
    torch.manual_seed(1)
    torch.save(O.init_pretrained_state_dict(cfg), tmp_path / "pretraining" / "model_state.pth")
    ds = data.SyntheticSLUDataset(2, 32, 48000, cfg.values_per_slot, seed=77)
    dev_batches = [tuple(t.cuda() for t in b) for b in ds.batches]
    n_steps = 7                                          # three eager steps, the capture, three replays
    loader = [dev_batches[i % 2] for i in range(n_steps)]

    def run(mode):
        monkeypatch.setenv("SLU_WGRAD_BRANCH", mode)
        monkeypatch.setenv("SLU_GRAPHS", "1")
        monkeypatch.setenv("SLU_LOOKAHEAD", "auto")
        torch.manual_seed(2)
        model = models.Model(cfg)
        for p in model.parameters():
            p.requires_grad = True                        # unfreeze_all_layers end state: nothing to look ahead to
        models.set_dropout_seed(99)
        trainer = training.Trainer(model, cfg)
        model.train()
        losses = []
        import contextlib
        with contextlib.closing(trainer._iterate(loader, True, False)) as it:
            for v, _ in it:
                losses.append(float(v[0]))
        torch.cuda.synchronize()
        assert trainer.graph_stats()["step_graphs"] == 1 and trainer.graph_stats()["capture_failures"] == 0
        return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    l_def, sd_def = run("pass")
    l_join, sd_join = run("layer")
    assert l_def == l_join
    for k, v in sd_join.items():
        assert torch.equal(v, sd_def[k]), k
    l_off, sd_off = run("0")
    assert max(abs(a - b) for a, b in zip(l_def, l_off)) <= 1e-4
    worst = max((sd_off[k].double() - sd_def[k].double()).abs().max().item() for k in sd_off)
    print("branched vs in-line weight-gradient launches after %d steps: worst parameter difference %.2e" % (n_steps, worst))
    assert worst <= 5e-3          # Adam normalises by |g|: a near-zero gradient entry may flip sign between summation orders
