"""PCM16 batches (SLU_PCM16_BATCHES=1: the loaders hand wavs over as int16, half the bytes on PCIe) through the HIP path:
the first stage computes on sample / 32768 — exactly the float32 value the reference's loaders produce (data.py:273-293:
sox / soundfile decode PCM16 as sample / 32768) — so everything downstream is bit-identical to float batches."""
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


def _cfg(tmp_path):
    import data
    cfg = O.OracleConfig(pretraining_type=2)                 # experiments/no_unfreezing.cfg
    cfg.folder = str(tmp_path)
    cfg.training_lr = 0.001
    cfg.starting_unfreezing_index = 1
    cfg.Sy_intent = data.synthetic_Sy_intent(cfg.values_per_slot)
    os.makedirs(tmp_path / "pretraining", exist_ok=True)
    os.makedirs(tmp_path / "training", exist_ok=True)
    torch.manual_seed(1)
    torch.save(O.init_pretrained_state_dict(cfg), tmp_path / "pretraining" / "model_state.pth")
    return cfg


def _pcm(x):
    return (x * 32768.0).round().clamp(-32768, 32767).to(torch.int16)


def test_pcm16_waveforms_equal_their_float_conversion(tmp_path, monkeypatch):
    """Encoder features and logits from int16 samples == from sample / 32768 as float32, bit for bit: frozen first block
    (the split-precision kernel reads the int16 samples itself), trainable first block (slu_pcm16_to_f32 + exact kernels),
    and the CPU oracle agrees on the converted waveform."""
    import models
    monkeypatch.delenv("SLU_FROZEN_MATH", raising=False)
    cfg = _cfg(tmp_path)
    torch.manual_seed(2)
    model = models.Model(cfg)
    model.eval()
    pm = model.pretrained_model
    g = torch.Generator().manual_seed(3)
    x16 = _pcm(0.1 * torch.randn(6, 16000, generator=g))
    xf = x16.float() / 32768.0
    with torch.no_grad():
        a = pm.compute_features(x16).float().cpu()
        b = pm.compute_features(xf).float().cpu()
    assert torch.equal(a, b)
    la, pa = model.predict_intents(x16)
    lb, pb = model.predict_intents(xf)
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    pre = torch.load(tmp_path / "pretraining" / "model_state.pth")
    ref = O.encoder_stages(pre, xf, cfg, None, explicit_gru=False)["features"]
    assert (a - ref).abs().max().item() <= 1e-4
    # every layer trainable: the first block runs on the exact fp32 kernels behind the conversion kernel
    for p in model.parameters():
        p.requires_grad_(True)
    with torch.no_grad():
        c = pm.compute_features(x16).float().cpu()
        d = pm.compute_features(xf).float().cpu()
    assert torch.equal(c, d) and (c - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize("where", ["device", "host"])
def test_lookahead_training_on_pcm16_batches_equals_float_batches(tmp_path, monkeypatch, where):
    """The look-ahead loop fed int16 batches (device-resident: read in place through the row-pointer table; pinned host
    batches: copied as int16 — half the H2D bytes) ends with the losses and parameters of the same loop on float batches."""
    import data
    import models
    import training
    monkeypatch.delenv("SLU_FROZEN_MATH", raising=False)
    cfg = _cfg(tmp_path)
    ds = data.SyntheticSLUDataset(3, 8, 16000, cfg.values_per_slot, seed=9)
    b16 = [(_pcm(x), y) for x, y in ds.batches]
    bf = [(x.float() / 32768.0, y) for x, y in b16]
    place = (lambda t: t.cuda()) if where == "device" else (lambda t: t.pin_memory())

    def run(batches):
        loader = [tuple(place(t) for t in batches[i % 3]) for i in range(12)]
        monkeypatch.setenv("SLU_LOOKAHEAD", "4")
        torch.manual_seed(2)
        model = models.Model(cfg)
        models.set_dropout_seed(4321)
        trainer = training.Trainer(model, cfg)
        model.train()
        losses = []
        with contextlib.closing(trainer._iterate(loader, True, False, accumulate=True)) as it:
            for v, _ in it:
                losses.append(float(v[0]))
        torch.cuda.synchronize()
        return trainer, losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    tr_f, lf, sf = run(bf)
    tr_i, li, si = run(b16)
    assert li == lf and len(set(li)) == 12
    for k, v in sf.items():
        assert torch.equal(v, si[k]), k
    assert tr_i.graph_stats()["capture_failures"] == 0 and tr_i.graph_stats()["prefix_graphs"] >= 1
    if where == "device":
        from slu_hip import ops
        tables = [g[1] for slot in tr_i._slots for g in slot.graphs.values() if g is not None]
        assert tables and all(isinstance(t, ops.RowTable) and t.dtype == torch.int16 for t in tables)
