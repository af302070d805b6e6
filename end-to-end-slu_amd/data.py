"""Config reader and dataset surface of the reference's `data.py`, as far as the encoder hot path
needs it.

`read_config(path) -> Config` parses the reference's experiment .cfg files into the same attribute
bag (reference data.py:15-130: same attribute names, types, defaults, derived values and the same
side effects — experiment folder creation and a copy of the cfg).  It additionally accepts the nine
legacy cfgs the reference's own parser rejects (they carry `dataset_subset_percentage` instead of
the four `*_subset_percentage` keys): the legacy key maps to
`real_dataset_subset_percentage`, the other three default to 1.0.

The wav/CSV/TextGrid loaders of the reference (data.py:132-545) are IO outside the MI355X hot path
("next"); this module provides Fluent-Speech-Commands-shaped and
LibriSpeech-shaped SYNTHETIC datasets with the same duck type (`.loader` yielding the same batch
tuples; class names `SLUDataset` / `ASRDataset`, which `training.Trainer` dispatches on).
"""
import configparser
import os
import shutil

import torch


class Config:
    def __init__(self):
        self.use_sincnet = True


def _ints(s):
    return [int(v) for v in s.split(",")]


def _floats(s):
    return [float(v) for v in s.split(",")]


def _strs(s):
    return [v for v in s.split(",")]


def _flag(s):
    return s == "True"


def _none_or_str(s):
    return None if s == "None" else s


# (section, key, converter) in the reference's reading order; attribute name == key
_REQUIRED = [
    ("phoneme_module", "use_sincnet", _flag), ("phoneme_module", "fs", int),
    ("phoneme_module", "cnn_N_filt", _ints), ("phoneme_module", "cnn_len_filt", _ints),
    ("phoneme_module", "cnn_stride", _ints), ("phoneme_module", "cnn_max_pool_len", _ints),
    ("phoneme_module", "cnn_act", _strs), ("phoneme_module", "cnn_drop", _floats),
    ("phoneme_module", "phone_rnn_num_hidden", _ints), ("phoneme_module", "phone_downsample_len", _ints),
    ("phoneme_module", "phone_downsample_type", _strs), ("phoneme_module", "phone_rnn_drop", _floats),
    ("phoneme_module", "phone_rnn_bidirectional", _flag),
    ("word_module", "word_rnn_num_hidden", _ints), ("word_module", "word_downsample_len", _ints),
    ("word_module", "word_downsample_type", _strs), ("word_module", "word_rnn_drop", _floats),
    ("word_module", "word_rnn_bidirectional", _flag), ("word_module", "vocabulary_size", int),
    ("intent_module", "intent_rnn_num_hidden", _ints), ("intent_module", "intent_downsample_len", _ints),
    ("intent_module", "intent_downsample_type", _strs), ("intent_module", "intent_rnn_drop", _floats),
    ("intent_module", "intent_rnn_bidirectional", _flag),
]
_SEQ2SEQ_KEYS = ["intent_encoder_dim", "num_intent_encoder_layers", "intent_decoder_dim",
                 "num_intent_decoder_layers", "intent_decoder_key_dim", "intent_decoder_value_dim"]
_PRETRAINING = [("asr_path", str), ("pretraining_type", int), ("pretraining_lr", float),
                ("pretraining_batch_size", int), ("pretraining_num_epochs", int),
                ("pretraining_length_mean", float), ("pretraining_length_var", float)]
_TRAINING_HEAD = [("slu_path", str), ("unfreezing_type", int), ("training_lr", float),
                  ("training_batch_size", int), ("training_num_epochs", int)]
_SUBSETS = ["real_dataset_subset_percentage", "synthetic_dataset_subset_percentage",
            "real_speaker_subset_percentage", "synthetic_speaker_subset_percentage"]


def read_config(config_file):
    """Parse an experiment .cfg into a Config (reference data.py:19-130)."""
    parser = configparser.ConfigParser()
    parser.read(config_file)
    config = Config()
    config.seed = int(parser.get("experiment", "seed"))
    config.folder = parser.get("experiment", "folder")

    # side effects of the reference (data.py:29-33): experiment folder + a copy of the cfg
    if not os.path.isdir(config.folder):
        os.mkdir(config.folder)
        os.mkdir(os.path.join(config.folder, "pretraining"))
        os.mkdir(os.path.join(config.folder, "training"))
    try:
        shutil.copyfile(config_file, os.path.join(config.folder, "experiment.cfg"))
    except (shutil.SameFileError, OSError):
        pass

    for section, key, conv in _REQUIRED:
        setattr(config, key, conv(parser.get(section, key)))
    try:
        vals = [int(parser.get("intent_module", k)) for k in _SEQ2SEQ_KEYS]
        for k, v in zip(_SEQ2SEQ_KEYS, vals):
            setattr(config, k, v)
    except Exception:
        # partial seq2seq sections leave the earlier keys set, like the reference's sequential reads
        for k in _SEQ2SEQ_KEYS:
            try:
                setattr(config, k, int(parser.get("intent_module", k)))
            except Exception:
                break
        print("no seq2seq hyperparameters")

    for key, conv in _PRETRAINING[:2]:
        setattr(config, key, conv(parser.get("pretraining", key)))
    n_word, n_phone, n_cnn = (len(config.word_rnn_num_hidden), len(config.phone_rnn_num_hidden),
                              len(config.cnn_N_filt))
    start = {0: 1 + n_word + n_phone + n_cnn, 1: 1 + n_word, 2: 1, 3: 1}
    if config.pretraining_type in start:
        config.starting_unfreezing_index = start[config.pretraining_type]
    for key, conv in _PRETRAINING[2:]:
        setattr(config, key, conv(parser.get("pretraining", key)))

    for key, conv in _TRAINING_HEAD:
        setattr(config, key, conv(parser.get("training", key)))
    if parser.has_option("training", _SUBSETS[0]):
        for key in _SUBSETS:
            setattr(config, key, float(parser.get("training", key)))
    elif parser.has_option("training", "dataset_subset_percentage"):
        # legacy cfg (superset of the reference's behaviour, which raises NoOptionError here)
        config.real_dataset_subset_percentage = float(parser.get("training", "dataset_subset_percentage"))
        for key in _SUBSETS[1:]:
            config.__dict__[key] = float(parser.get("training", key)) if parser.has_option("training", key) else 1.0
    else:
        parser.get("training", _SUBSETS[0])          # raises configparser.NoOptionError like the reference
    config.train_wording_path = _none_or_str(parser.get("training", "train_wording_path"))
    config.test_wording_path = _none_or_str(parser.get("training", "test_wording_path"))
    config.augment = _flag(parser.get("training", "augment", fallback="False"))
    config.seq2seq = _flag(parser.get("training", "seq2seq", fallback="False"))
    try:
        config.dataset_upsample_factor = int(parser.get("training", "dataset_upsample_factor"))
    except Exception:
        config.dataset_upsample_factor = 1

    # total time decimation of the phoneme / word modules (reference data.py:121-128)
    config.phone_downsample_factor = 1
    for f in config.cnn_stride + config.cnn_max_pool_len + config.phone_downsample_len:
        config.phone_downsample_factor *= f
    config.word_downsample_factor = config.phone_downsample_factor
    for f in config.word_downsample_len:
        config.word_downsample_factor *= f
    return config


# ------------------------------------------------------------------------------------------------
# synthetic datasets (shape-compatible stand-ins for the reference's wav/CSV loaders)
# ------------------------------------------------------------------------------------------------
FSC_VALUES_PER_SLOT = [6, 14, 4]       # Fluent Speech Commands: action / object / location


def synthetic_Sy_intent(values_per_slot=FSC_VALUES_PER_SLOT):
    names = ["action", "object", "location"]
    return {names[s]: {"%s_%d" % (names[s], v): v for v in range(n)} for s, n in enumerate(values_per_slot)}


class _SyntheticLoader:
    """Iterable of pre-generated batches (one pass per __iter__, like a DataLoader)."""

    def __init__(self, batches):
        self.batches = batches

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


class SLUDataset(torch.utils.data.Dataset):
    """Synthetic SLU dataset: `.loader` yields (x (B,T) float32, y_intent (B,3) int64) like
    CollateWavsSLU (reference data.py:344-376): zero-mean 0.1-RMS noise waveforms, uniform labels."""

    def __init__(self, num_batches, batch_size, num_samples, values_per_slot=FSC_VALUES_PER_SLOT,
                 seed=1234, Sy_intent=None, pin=False):
        g = torch.Generator().manual_seed(seed)
        self.Sy_intent = Sy_intent or synthetic_Sy_intent(values_per_slot)
        self.seq2seq = False
        batches = []
        for _ in range(num_batches):
            x = 0.1 * torch.randn(batch_size, num_samples, generator=g)
            y = torch.stack([torch.randint(0, n, (batch_size,), generator=g) for n in values_per_slot], dim=1)
            if pin and torch.cuda.is_available():
                x, y = x.pin_memory(), y.pin_memory()
            batches.append((x, y))
        self.batches = batches
        self.loader = _SyntheticLoader(batches)

    def __len__(self):
        return sum(len(b[0]) for b in self.batches)

    def __getitem__(self, idx):
        bs = len(self.batches[0][0])
        x, y = self.batches[idx // bs]
        return x[idx % bs], y[idx % bs]


class ASRDataset(torch.utils.data.Dataset):
    """Synthetic ASR pre-training dataset: `.loader` yields (x (B,T), y_phoneme (B,ceil(T/640)),
    y_word (B,ceil(T/2560))) like CollateWavsASR (reference data.py:511-545); 10 % of the frame
    labels are the ignore index -1."""

    def __init__(self, num_batches, batch_size, num_samples, config, seed=1234):
        g = torch.Generator().manual_seed(seed)
        Tp = -(-num_samples // config.phone_downsample_factor)
        Tw = -(-num_samples // config.word_downsample_factor)
        batches = []
        for _ in range(num_batches):
            x = 0.1 * torch.randn(batch_size, num_samples, generator=g)
            yp = torch.randint(0, config.num_phonemes, (batch_size, Tp), generator=g)
            yw = torch.randint(0, config.vocabulary_size, (batch_size, Tw), generator=g)
            yp[torch.rand(batch_size, Tp, generator=g) < 0.1] = -1
            yw[torch.rand(batch_size, Tw, generator=g) < 0.1] = -1
            batches.append((x, yp, yw))
        self.batches = batches
        self.loader = _SyntheticLoader(batches)

    def __len__(self):
        return sum(len(b[0]) for b in self.batches)

    def __getitem__(self, idx):
        bs = len(self.batches[0][0])
        b = self.batches[idx // bs]
        return tuple(t[idx % bs] for t in b)


def _synthetic_spec(path):
    """"synthetic" or "synthetic:<batches>x<batch>x<samples>" -> (batches, batch, samples) or None."""
    if not path.startswith("synthetic"):
        return None
    spec = path.split(":", 1)[1] if ":" in path else ""
    if not spec:
        return 8, None, 48000
    nb, bs, ns = (int(v) for v in spec.split("x"))
    return nb, bs, ns


def get_SLU_datasets(config):
    """(train, valid, test) SLU datasets; also sets config.values_per_slot / Sy_intent /
    num_phonemes like the reference (data.py:132-240).  Only `slu_path = synthetic[:NxBxT]` is
    served here; reading Fluent Speech Commands wavs is the "next" row of SURVEY.md §8(f)."""
    spec = _synthetic_spec(config.slu_path)
    if spec is None:
        raise NotImplementedError(
            "real-data SLU loading (CSV + wav decoding, reference data.py:132-391) is outside the "
            "MI355X hot path of this package; set slu_path=synthetic[:<batches>x<batch>x<samples>]")
    nb, bs, ns = spec
    bs = bs or config.training_batch_size
    config.values_per_slot = list(FSC_VALUES_PER_SLOT)
    config.Sy_intent = synthetic_Sy_intent(config.values_per_slot)
    phonemes = os.path.join(config.folder, "pretraining", "phonemes.txt")
    if os.path.isfile(phonemes):
        with open(phonemes) as f:
            config.num_phonemes = len([ln for ln in f.read().split("\n") if ln != ""])
    else:
        print("No phoneme file found.")
        config.num_phonemes = 42
    mk = lambda n, seed: SLUDataset(n, bs, ns, config.values_per_slot, seed=seed, Sy_intent=config.Sy_intent)
    return mk(nb, config.seed), mk(max(1, nb // 4), config.seed + 1), mk(max(1, nb // 4), config.seed + 2)


def get_ASR_datasets(config):
    """(train, valid, test) ASR datasets; `asr_path = synthetic[:NxBxT]` only (see get_SLU_datasets)."""
    spec = _synthetic_spec(config.asr_path)
    if spec is None:
        raise NotImplementedError(
            "LibriSpeech + TextGrid loading (reference data.py:393-545) is outside the MI355X hot "
            "path of this package; set asr_path=synthetic[:<batches>x<batch>x<samples>]")
    nb, bs, ns = spec
    bs = bs or config.pretraining_batch_size
    config.num_phonemes = 42
    mk = lambda n, seed: ASRDataset(n, bs, ns, config, seed=seed)
    return mk(nb, config.seed), mk(max(1, nb // 4), config.seed + 1), mk(max(1, nb // 4), config.seed + 2)
