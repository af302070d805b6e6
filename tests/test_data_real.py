"""Real-data SLU input pipeline against fixture g10: the REFERENCE's
get_SLU_datasets / SLUDataset / CollateWavsSLU (data.py:132-376) run by tests/golden/make_goldens.py on
the tiny FSC-shaped tree of tests/slu_data_fixture.py."""
import json
import os
import types

import numpy as np
import pytest
import torch

import data
import slu_data_fixture as fx

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g10_slu_data.json")))


def _config(root, over):
    cfg = types.SimpleNamespace(
        slu_path=root, folder=root, seq2seq=False, training_batch_size=4, seed=1,
        real_speaker_subset_percentage=1.0, synthetic_speaker_subset_percentage=1.0,
        real_dataset_subset_percentage=1.0, synthetic_dataset_subset_percentage=1.0,
        train_wording_path=None, test_wording_path=None, dataset_upsample_factor=1)
    for k, v in over.items():
        setattr(cfg, k, os.path.join(root, v) if k.endswith("_path") else v)
    return cfg


@pytest.mark.parametrize("name", sorted(fx.VARIANTS))
def test_get_slu_datasets_matches_reference(name, tmp_path, capsys, monkeypatch):
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    over, np_seed, tree_kw = fx.VARIANTS[name]
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=11, **tree_kw)
    cfg = _config(root, over)
    np.random.seed(np_seed)
    tr, va, te = data.get_SLU_datasets(cfg)
    g = GOLD[name]
    assert capsys.readouterr().out == g["stdout"]
    assert cfg.values_per_slot == g["values_per_slot"]
    assert cfg.Sy_intent == g["Sy_intent"]
    assert [len(tr), len(va), len(te)] == g["len"]
    for tag, ds in (("train", tr), ("valid", va), ("test", te)):
        assert [str(p) for p in ds.df.path.tolist()] == g[tag + "_paths"]
        assert [int(i) for i in ds.df.index.tolist()] == g[tag + "_index"]
    for it in g.get("items", []):
        x, y = tr[it["idx"]]
        assert x.dtype == np.float32 and str(x.dtype) == it["dtype"]
        assert len(x) == it["n"] and [int(v) for v in y] == it["y"]
        assert [float(v) for v in x[:4]] == it["first"]
        assert float(np.float64(x).sum()) == it["sum"]


def test_seq2seq_datasets_match_reference(tmp_path, capsys, monkeypatch):
    """config.seq2seq (reference data.py:143-146, 186-187, 201-208, 318-326): the *_seq2seq.csv splits, the output
    alphabet (the reference's set of characters; its order is hash-salted there, sorted here), <sos> ... <eos> label
    sequences."""
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=11, seq2seq=True)
    cfg = _config(root, {})
    cfg.seq2seq = True
    np.random.seed(0)
    tr, va, te = data.get_SLU_datasets(cfg)
    capsys.readouterr()
    g = GOLD["seq2seq"]
    Sy = cfg.Sy_intent
    assert Sy[0] == g["first"] == "<sos>" and Sy[-1] == g["last"] == "<eos>"
    assert Sy[1:-1] == g["alphabet_sorted"]
    assert [len(tr), len(va), len(te)] == g["len"]
    assert [str(p) for p in tr.df.path.tolist()] == g["train_paths"]
    for it in g["items"]:
        x, y = tr[it["idx"]]
        assert len(x) == it["n"] and [Sy[k] for k in y] == it["labels"]
    x, y = next(iter(tr.loader))
    assert y.dtype == torch.float32 and y.dim() == 3 and y.shape[2] == len(Sy) and float(y.sum()) == y.shape[0] * y.shape[1]
    assert torch.equal(y[:, 0].argmax(1), torch.zeros(y.shape[0], dtype=torch.int64))          # <sos> first


def test_seq2seq_collate_matches_reference():
    g = GOLD["collate_seq2seq"]
    rs = np.random.RandomState(g["seed"])
    batch = [(rs.randn(n).astype(np.float32), [0] + [int(rs.randint(1, 9)) for _ in range(u)] + [9])
             for n, u in zip(g["lens"], g["ulens"])]
    x, y = data.CollateWavsSLU(g["labels"], True)(batch)
    assert torch.equal(x, torch.tensor(g["x"], dtype=torch.float32))
    assert list(y.shape) == g["y_shape"] and str(y.dtype) == g["y_dtype"] and float(y.sum()) == g["y_sum"]
    assert y.max(dim=2)[1].tolist() == g["y_idx"]                                              # <eos> padding


def test_collate_matches_reference():
    g = GOLD["collate"]
    rs = np.random.RandomState(g["seed"])
    batch = [(rs.randn(n).astype(np.float32), [int(rs.randint(6)), int(rs.randint(14)), int(rs.randint(4))])
             for n in g["lens"]]
    x, y = data.CollateWavsSLU({"action": {}, "object": {}, "location": {}}, False)(batch)
    assert str(x.dtype) == g["x_dtype"] and str(y.dtype) == g["y_dtype"]
    assert torch.equal(x, torch.tensor(g["x"], dtype=torch.float32))
    assert torch.equal(y, torch.tensor(g["y"], dtype=torch.int64))


def test_loader_batches_and_wav_formats(tmp_path, monkeypatch):
    from scipy.io import wavfile
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=3)
    cfg = _config(root, {})
    tr, va, te = data.get_SLU_datasets(cfg)
    seen = 0
    for x, y in va.loader:
        assert x.dtype == torch.float32 and y.dtype == torch.int64 and y.shape == (x.shape[0], 3)
        assert x.shape[0] <= cfg.training_batch_size
        assert (x[:, -1] != 0).any()                      # padded to the longest item of the batch, not beyond
        seen += x.shape[0]
    assert seen == len(va)
    # other sample formats: first channel, full-scale conventions
    p = os.path.join(root, "fmt.wav")
    wavfile.write(p, 16000, np.array([[16384, 1], [-32768, 2]], dtype=np.int16))
    assert data.read_wav(p)[0].tolist() == [0.5, -1.0]
    wavfile.write(p, 16000, np.array([2 ** 30, -2 ** 31], dtype=np.int32))
    assert data.read_wav(p)[0].tolist() == [0.5, -1.0]
    wavfile.write(p, 16000, np.array([192, 0], dtype=np.uint8))
    assert data.read_wav(p)[0].tolist() == [0.5, -1.0]
    wavfile.write(p, 8000, np.array([0.25, -0.75], dtype=np.float32))
    x, fs = data.read_wav(p)
    assert x.tolist() == [0.25, -0.75] and fs == 8000
    # a slot value unseen in training fails at item access, as in the reference
    te._values[0] = ("unseen action",) + te._values[0][1:]
    with pytest.raises(KeyError):
        te[0]


def test_pcm16_batches_are_the_float_batches_times_32768(tmp_path, monkeypatch):
    """SLU_PCM16_BATCHES=1: PCM16 wavs travel as int16 batches (half the bytes on PCIe); sample / 32768 — what the model's
    first stage computes — is exactly the float32 batch of the default loader (and of the reference: data.py:273-293), padding
    included.  A batch with a non-PCM16 item falls back to float32."""
    from scipy.io import wavfile
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=3)
    cfg = _config(root, {})
    monkeypatch.delenv("SLU_PCM16_BATCHES", raising=False)
    assert not data.pcm16_batches()
    _, va, _ = data.get_SLU_datasets(cfg)
    items_f = [va[i] for i in range(len(va))]
    monkeypatch.setenv("SLU_PCM16_BATCHES", "1")
    _, va16, _ = data.get_SLU_datasets(cfg)
    items_i = [va16[i] for i in range(len(va16))]
    assert len(items_i) == len(items_f) > 4
    for (xi, yi), (xf, yf) in zip(items_i, items_f):
        assert np.asarray(xi).dtype == np.int16 and list(yi) == list(yf)
        assert np.array_equal(np.asarray(xi).astype(np.float32) / 32768.0, np.asarray(xf))
    collate = data.CollateWavsSLU(va.Sy_intent, False)
    x16, y16 = collate(items_i[:5])
    xf, yf = collate(items_f[:5])
    assert x16.dtype == torch.int16 and x16.shape == xf.shape and torch.equal(y16, yf)
    assert torch.equal(x16.float() / 32768.0, xf)                # padding included
    for x, _ in va16.loader:
        assert x.dtype == torch.int16
    p = os.path.join(root, "f.wav")
    wavfile.write(p, 16000, np.array([0.25, -0.75], dtype=np.float32))
    a, _ = data.read_wav(p, keep_pcm16=True)
    assert a.dtype == np.float32                              # not PCM16: stays float
    mixed = data._pad_waveforms([np.array([16384, -32768], dtype=np.int16), a], 3)
    assert mixed.dtype == torch.float32 and mixed.tolist() == [[0.5, -1.0, 0.0], [0.25, -0.75, 0.0]]


def test_trainer_consumes_real_loader_shapes(tmp_path, monkeypatch):
    """The batch tuples of the real loader are what Trainer._forward_losses expects (x (B,T), y (B,3))."""
    monkeypatch.setenv("SLU_DATA_WORKERS", "2")          # worker processes + collate in the workers
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=4)
    tr, _, _ = data.get_SLU_datasets(_config(root, {"dataset_upsample_factor": 2}))
    n = 0
    for x, y in tr.loader:
        assert x.ndim == 2 and y.shape[1] == 3
        n += len(x)
    assert n == len(tr) == 2 * len(tr.df)


# ------------------------------------------------------------------------------------------------
# ASR pre-training input pipeline (reference data.py:393-545) against fixture g11
# ------------------------------------------------------------------------------------------------
GOLD_ASR = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g11_asr_data.json")))


def _asr_config(root, base):
    folder = os.path.join(root, "exp")
    os.makedirs(os.path.join(folder, "pretraining"), exist_ok=True)
    return types.SimpleNamespace(asr_path=base, folder=folder, vocabulary_size=5, pretraining_batch_size=3,
                                 pretraining_length_mean=1.0, pretraining_length_var=0.4,
                                 phone_downsample_factor=40, word_downsample_factor=160, seed=1)


def test_get_asr_datasets_matches_reference(tmp_path, capsys, monkeypatch):
    import hashlib
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    root = str(tmp_path)
    base = fx.make_asr_tree(root, seed=5)
    cfg = _asr_config(root, base)
    g = GOLD_ASR
    tr, va, te = data.get_ASR_datasets(cfg)
    assert capsys.readouterr().out == g["stdout_first"]
    assert cfg.num_phonemes == g["num_phonemes"]
    assert sorted(tr.Sy_phoneme) == g["Sy_phoneme_sorted"] and sorted(tr.Sy_word) == g["Sy_word_set"]
    assert [len(tr), len(va), len(te)] == g["len"]
    rel = lambda p: os.path.relpath(p, base)
    for tag, ds in (("train", tr), ("valid", va), ("test", te)):
        assert sorted(rel(p) for p in ds.wav_paths) == g[tag + "_wavs"]
    # the vocabulary files written by the first call are what a second call reads back
    assert open(os.path.join(cfg.folder, "pretraining", "phonemes.txt")).read().split("\n")[:-1] == tr.Sy_phoneme
    with open(os.path.join(cfg.folder, "pretraining", "phonemes.txt"), "w") as f:
        f.write("\n".join(g["Sy_phoneme"]) + "\n")
    with open(os.path.join(cfg.folder, "pretraining", "words.txt"), "w") as f:
        f.write("\n".join(g["Sy_word"]) + "\n")
    tr, va, te = data.get_ASR_datasets(cfg)
    assert capsys.readouterr().out == g["stdout_second"]
    assert tr.Sy_phoneme == g["Sy_phoneme"] and tr.Sy_word == g["Sy_word"]
    for i in range(len(tr)):
        key = rel(tr.wav_paths[i])
        it = g["items"][key]
        torch.manual_seed(int(hashlib.sha256(key.encode()).hexdigest()[:6], 16))
        x, yp, yw = tr[i]
        assert str(np.asarray(x).dtype) == it["dtype"] and len(x) == it["n"]
        assert [float(v) for v in x[:3]] == it["first"] and float(np.sum(x)) == it["sum"]
        assert [int(v) for v in yp] == it["y_phoneme"] and [int(v) for v in yw] == it["y_word"]
    # loader: shapes and the ignore index as padding value
    seen = 0
    for x, yp, yw in tr.loader:
        assert x.dtype == torch.float32 and yp.dtype == torch.int64 and yw.dtype == torch.int64
        assert x.shape[0] == yp.shape[0] == yw.shape[0] <= 3
        seen += len(x)
    assert seen == len(tr)


def test_collate_asr_matches_reference():
    g = GOLD_ASR["collate"]
    rs = np.random.RandomState(g["seed"])
    batch = [(rs.randn(n), [int(v) for v in rs.randint(-1, 9, size=-(-n // 4))],
              [int(v) for v in rs.randint(-1, 5, size=-(-n // 16))]) for n in g["lens"]]
    x, yp, yw = data.CollateWavsASR()(batch)
    assert [str(x.dtype), str(yp.dtype), str(yw.dtype)] == g["dtypes"]
    assert torch.equal(x, torch.tensor(g["x"], dtype=torch.float32))
    assert torch.equal(yp, torch.tensor(g["yp"])) and torch.equal(yw, torch.tensor(g["yw"]))


def test_textgrid_reader_edge_cases(tmp_path):
    p = str(tmp_path / "a.TextGrid")
    fx.write_textgrid(p, 1.5, [("he said ""hi""", 0.5), ("", 1.5)], [("HH", 0.2), ("IY1", 0.5), ("sil", 1.5)])
    tg = data.read_textgrid(p)
    assert [m for _, _, m in tg["phones"]] == ["HH", "IY1", "sil"]
    assert tg["words"][0][:2] == (0.0, 0.5) and tg["words"][1] == (0.5, 1.5, "")
    assert tg["phones"][1][:2] == (0.2, 0.5)


def test_length_bucketed_batches(tmp_path, monkeypatch):
    """SLU_BUCKET_BATCHES=1 + SLU_PAD_TO_MULTIPLE: every utterance once per epoch, one shape per bucket,
    equal shapes adjacent (what the look-ahead pipeline groups on), a new order every epoch."""
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    monkeypatch.setenv("SLU_PAD_TO_MULTIPLE", "500")
    monkeypatch.setenv("SLU_BUCKET_BATCHES", "1")
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=8, sizes=(60, 30, 9, 9))
    torch.manual_seed(0)
    tr, va, te = data.get_SLU_datasets(_config(root, {"dataset_upsample_factor": 2}))
    assert isinstance(tr.loader.batch_sampler, data.LengthBucketBatchSampler)
    lengths = [data.wav_num_samples(p) for p in tr._paths]
    epochs = []
    for _ in range(2):
        order, shapes = [], []
        for batch in tr.loader.batch_sampler:
            order.append(list(batch))
        seen = sorted(i for b in order for i in b)
        assert seen == list(range(len(tr)))                       # every (upsampled) index exactly once
        for b in order:
            ts = {-(-lengths[i % len(lengths)] // 500) for i in b}
            assert len(ts) == 1 and len(b) <= 4
            shapes.append(ts.pop())
        changes = sum(1 for a, b in zip(shapes, shapes[1:]) if a != b)
        assert changes == len(set(shapes)) - 1                    # the batches of a bucket are consecutive
        epochs.append(order)
    assert epochs[0] != epochs[1]
    n = 0
    for x, y in tr.loader:                                        # collate pads to the bucket's multiple
        assert x.shape[1] % 500 == 0 and x.shape[1] - 500 < max(1, int((x != 0).any(0).nonzero().max()) + 1)
        n += len(x)
    assert n == len(tr)


@pytest.mark.parametrize("bucketed", [False, True])
def test_training_split_is_sharded_across_ranks(tmp_path, monkeypatch, bucketed):
    """Under data parallelism the ranks draw disjoint parts of each training epoch (same number of steps
    on every rank, a new partition per epoch); validation stays whole on every rank."""
    monkeypatch.setenv("SLU_DATA_WORKERS", "0")
    if bucketed:
        monkeypatch.setenv("SLU_PAD_TO_MULTIPLE", "500")
        monkeypatch.setenv("SLU_BUCKET_BATCHES", "1")
    root = str(tmp_path)
    fx.make_fsc_tree(root, seed=12, sizes=(41, 20, 9, 9))
    per_rank = {}
    for rank in (0, 1):
        monkeypatch.setattr(data, "_world", lambda r=rank: (r, 2))
        np.random.seed(0)
        tr, va, te = data.get_SLU_datasets(_config(root, {}))
        assert sum(len(x) for x, _ in va.loader) == len(va)                  # whole validation set
        epochs = []
        for epoch in (0, 1):
            smp = tr.loader.batch_sampler if bucketed else tr.loader.sampler
            smp.set_epoch(epoch)
            if bucketed:
                idx = [list(b) for b in tr.loader.batch_sampler]
            else:
                idx = [list(tr.loader.sampler)]
            epochs.append(idx)
        per_rank[rank] = epochs
    for epoch in (0, 1):
        a = [i for b in per_rank[0][epoch] for i in b]
        b = [i for b in per_rank[1][epoch] for i in b]
        assert len(per_rank[0][epoch]) == len(per_rank[1][epoch])            # same number of steps
        if bucketed:
            assert set(a) | set(b) == set(range(61))
        else:
            assert len(a) == len(b) == 31 and set(a) | set(b) == set(range(61))   # 61 items -> 31 + 31 (one repeated)
    assert per_rank[0][0] != per_rank[0][1]
