"""Config reader and dataset surface of the reference's `data.py`, as far as the encoder hot path
needs it.

`read_config(path) -> Config` parses the reference's experiment .cfg files into the same attribute
bag (reference data.py:15-130: same attribute names, types, defaults, derived values and the same
side effects — experiment folder creation and a copy of the cfg).  It additionally accepts the nine
legacy cfgs the reference's own parser rejects (they carry `dataset_subset_percentage` instead of
the four `*_subset_percentage` keys): the legacy key maps to
`real_dataset_subset_percentage`, the other three default to 1.0.

Datasets: `get_SLU_datasets` / `SLUDataset` / `CollateWavsSLU` read a Fluent-Speech-Commands tree
(CSV splits + wavs) with the reference's split / subset / label-dictionary logic (data.py:132-376;
SURVEY.md §8(f) rank 1), feeding pinned, zero-padded (B, T_max) batches to the encoder; with
`get_ASR_datasets` / `ASRDataset` / `CollateWavsASR` read a LibriSpeech + TextGrid alignment tree
(data.py:393-545).  With `slu_path` / `asr_path = synthetic[:NxBxT]` FSC- / LibriSpeech-shaped SYNTHETIC
datasets with the same duck type are served (`.loader` yielding the same batch tuples;
`training.Trainer` dispatches on `SLUDataset` / `ASRDataset`).
"""
import configparser
import os
import shutil
from collections import Counter

import numpy as np
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
    # (race-free: under torch.distributed every rank calls read_config; only rank 0 copies the cfg)
    for sub in ("", "pretraining", "training"):
        os.makedirs(os.path.join(config.folder, sub), exist_ok=True)
    if int(os.environ.get("RANK", "0")) == 0:
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


def pcm16_batches():
    """SLU_PCM16_BATCHES=1: loaders hand PCM16 audio over as int16 batches (B, T) — half the bytes on PCIe; the model's
    first stage computes on sample / 32768 (slu_hip.ops.PCM16_SCALE), which is exactly the float32 value the default
    loaders (and the reference's, data.py:273-293) produce, so losses and parameters are bit-identical.  Off by default:
    code that consumes `dataset.loader` directly keeps seeing the reference's float32 waveforms."""
    return os.environ.get("SLU_PCM16_BATCHES", "0") == "1"


def read_wav(path, keep_pcm16=False):
    """First channel of a wav file as float32 in [-1, 1) and its sample rate — the array
    `SoxEffectsChain.sox_build_flow_effects()` hands the reference (data.py:273-293; augmentation is
    hard-wired off there, so the chain is a plain decode): PCM16 / 32768, PCM32 / 2^31, unsigned 8-bit
    (v - 128) / 128, float wavs as stored.  keep_pcm16: PCM16 files come back as the int16 samples themselves."""
    from scipy.io import wavfile
    fs, a = wavfile.read(path)
    if a.ndim > 1:
        a = a[:, 0]
    if a.dtype == np.int16 and keep_pcm16:
        return np.ascontiguousarray(a), fs
    if a.dtype == np.int16:
        x = a.astype(np.float32) / 32768.0
    elif a.dtype == np.int32:
        x = (a.astype(np.float64) / 2147483648.0).astype(np.float32)
    elif a.dtype == np.uint8:
        x = (a.astype(np.float32) - 128.0) / 128.0
    else:
        x = a.astype(np.float32)
    return x, fs


def _pad_waveforms(waves, T):
    """list of 1-D waveforms -> (B, T) zero-padded at the end: int16 when every item is PCM16 samples (pcm16_batches),
    else float32 (int16 items are converted as read_wav would have)."""
    as_np = [np.asarray(w) for w in waves]
    if all(a.dtype == np.int16 for a in as_np):
        x = torch.zeros(len(as_np), T, dtype=torch.int16)
        for i, a in enumerate(as_np):
            x[i, :len(a)] = torch.from_numpy(a)
        return x
    x = torch.zeros(len(as_np), T, dtype=torch.float32)
    for i, a in enumerate(as_np):
        x[i, :len(a)] = torch.from_numpy(a.astype(np.float32) / 32768.0) if a.dtype == np.int16 else torch.as_tensor(a, dtype=torch.float32)
    return x


def one_hot(letters, S):
    """letters (B, U) int64 label indices -> (B, U, S) float32 one-hot rows (reference data.py:331-342)."""
    out = torch.zeros(letters.shape[0], letters.shape[1], S)
    out.scatter_(2, letters.long().unsqueeze(2), 1.0)
    return out


class CollateWavsSLU:
    """list of (waveform, [action, object, location]) -> (x (B, T_max) float32 zero-padded at the end,
    y_intent (B, 3) int64), as reference data.py:344-376.  seq2seq: the labels are <sos> ... <eos> index
    sequences, padded with <eos> to the longest of the batch and returned one-hot, (B, U_max, num_labels) float32
    (data.py:363-376).  The batch is assembled directly in ONE buffer instead of per-row pad + stack; the DataLoader
    pins it (pin_memory=True, in the parent process — never in a forked worker) so that the H2D copy of the
    look-ahead slots is asynchronous."""

    def __init__(self, Sy_intent, seq2seq, pad_multiple=None):
        self.Sy_intent = Sy_intent
        self.num_labels = len(self.Sy_intent)
        self.seq2seq = seq2seq
        if self.seq2seq:
            self.EOS = self.Sy_intent.index("<eos>")
        # Opt-in (SLU_PAD_TO_MULTIPLE=n samples): round T_max up to a multiple of n so that ragged real
        # data falls into a few batch shapes (the look-ahead super-batches and hipGraphs are per shape).
        # Off by default: the extra trailing zeros are seen by the recurrences, i.e. it is not the
        # reference's batch any more.
        self.pad_multiple = int(os.environ.get("SLU_PAD_TO_MULTIPLE", "0")) if pad_multiple is None else pad_multiple

    def __call__(self, batch):
        T = max(len(x) for x, _ in batch)
        if self.pad_multiple > 1:
            T = -(-T // self.pad_multiple) * self.pad_multiple
        x = _pad_waveforms([xi for xi, _ in batch], T)
        if self.seq2seq:
            U = max(len(yi) for _, yi in batch)
            idx = torch.full((len(batch), U), self.EOS, dtype=torch.int64)
            for i, (_, yi) in enumerate(batch):
                idx[i, :len(yi)] = torch.as_tensor(list(yi), dtype=torch.int64)
            return x, one_hot(idx, self.num_labels)
        y = torch.tensor([list(yi) for _, yi in batch], dtype=torch.int64)
        return x, y


def _world():
    """(rank, world size) of the data-parallel job (one process per GPU), (0, 1) outside one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _loader_workers():
    """The reference starts one DataLoader worker per core (data.py:261); on a 256-thread MI355X host
    that is all start-up cost, so the default is capped (SLU_DATA_WORKERS overrides, 0 = in-process)."""
    env = os.environ.get("SLU_DATA_WORKERS")
    if env is not None:
        return int(env)
    return min(16, os.cpu_count() or 1)


class ForkSafeDataLoader(torch.utils.data.DataLoader):
    """DataLoader whose worker processes are forked from a process that holds GPU objects (hipGraphs, CU-masked streams,
    events, IPC windows).  A forked child must never FINALISE such objects — their destructors call into a HIP runtime
    that does not exist in the child — yet CPython may run a garbage collection right inside the child's fork bootstrap
    (`threading._after_fork`), and any unreachable cycle that holds one of them is then destroyed there: seen once as
    "Fatal Python error: Segmentation fault ... Garbage-collecting" in a worker of the ASR loader.  So: collect in the
    PARENT (where the runtime is alive) right before the workers start, and freeze what exists (`gc.freeze`: the children
    inherit a permanent generation their collections never visit); the parent unfreezes afterwards."""

    def __iter__(self):
        if self.num_workers <= 0:
            return super().__iter__()
        import gc
        gc.collect()
        gc.freeze()
        try:
            return super().__iter__()            # the workers are forked here
        finally:
            gc.unfreeze()


def wav_num_samples(path):
    """Number of frames of a wav file from its header (falls back to decoding for non-PCM files)."""
    import wave
    try:
        with wave.open(path, "rb") as w:
            return w.getnframes()
    except (wave.Error, EOFError):
        return len(read_wav(path)[0])


class LengthBucketBatchSampler(torch.utils.data.Sampler):
    """Opt-in batch sampler for ragged real data (SLU_BUCKET_BATCHES=1 together with
    SLU_PAD_TO_MULTIPLE=n): utterances are bucketed by ceil(length / n), batches are drawn inside a bucket
    (shuffled every epoch) and the batches of one bucket are emitted consecutively, buckets in random
    order.  With the collate function padding to the same multiple, every batch of a bucket has the SAME
    shape and equal shapes are adjacent — which is what lets the look-ahead pipeline form super-batches
    and replay its hipGraphs on real data (it groups consecutive equally-shaped batches).  The price is
    less length mixing inside an epoch than the reference's plain shuffle; off by default.
    Under data parallelism every rank draws the same batch list (seed + epoch) and keeps every
    world-th batch, wrapping around so that all ranks run the same number of steps."""

    def __init__(self, lengths, batch_size, multiple, shuffle=True, generator=None, rank=0, world=1, seed=0):
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.generator = generator
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        buckets = {}
        for i, n in enumerate(lengths):
            buckets.setdefault(-(-int(n) // int(multiple)), []).append(i)
        self.buckets = [buckets[k] for k in sorted(buckets)]

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _total(self):
        return sum(-(-len(b) // self.batch_size) for b in self.buckets)

    def __len__(self):
        return -(-self._total() // self.world)

    def _batches(self, g):
        order = torch.randperm(len(self.buckets), generator=g).tolist() if self.shuffle else range(len(self.buckets))
        for k in order:
            idx = self.buckets[k]
            if self.shuffle:
                idx = [idx[j] for j in torch.randperm(len(idx), generator=g).tolist()]
            for s0 in range(0, len(idx), self.batch_size):
                yield idx[s0:s0 + self.batch_size]

    def __iter__(self):
        if self.world == 1:
            yield from self._batches(self.generator)
            return
        g = torch.Generator().manual_seed(self.seed + self.epoch)       # the same list on every rank
        batches = list(self._batches(g))
        for k in range(len(self)):
            yield batches[(self.rank + k * self.world) % len(batches)]


class SLUDataset(torch.utils.data.Dataset):
    """Fluent-Speech-Commands-style dataset (reference data.py:246-329): rows of `df` (columns path,
    action, object, location), wavs under `base_path`; item = (float32 waveform, [3 label indices]);
    `len` = rows x upsample_factor; `.loader` = shuffling DataLoader with CollateWavsSLU."""

    def __init__(self, df, base_path, Sy_intent, config, upsample_factor=1, shard=False):
        self.df = df
        self.base_path = base_path
        self.Sy_intent = Sy_intent
        self.upsample_factor = upsample_factor
        self.augment = False
        self.SNRs = [0, 5, 10, 15, 20]
        self.seq2seq = config.seq2seq
        # label lookup and paths as plain lists: no per-item DataFrame indexing in the workers
        self._paths = [os.path.join(base_path, p) for p in df["path"].tolist()]
        if self.seq2seq:
            self._values = [str(v) for v in df["semantics"].tolist()]
            self._index = {c: i for i, c in enumerate(self.Sy_intent)}       # label -> index (list.index per char otherwise)
        else:
            self._values = list(zip(df["action"].tolist(), df["object"].tolist(), df["location"].tolist()))
        collate = CollateWavsSLU(self.Sy_intent, self.seq2seq)
        pin = torch.cuda.is_available()
        # Data parallelism (`shard`: the training split only): every rank draws a disjoint 1/world of the
        # epoch (DistributedSampler pads by repetition so that all ranks run the same number of steps);
        # validation / test stay whole on every rank, so their metrics are the single-process ones.
        rank, world = _world() if shard else (0, 1)
        seed = getattr(config, "seed", 0)
        if os.environ.get("SLU_BUCKET_BATCHES", "0") == "1" and collate.pad_multiple > 1:
            lengths = [wav_num_samples(p) for p in self._paths] * self.upsample_factor
            self.loader = ForkSafeDataLoader(
                self, num_workers=_loader_workers(), collate_fn=collate, pin_memory=pin,
                batch_sampler=LengthBucketBatchSampler(lengths, config.training_batch_size, collate.pad_multiple,
                                                       rank=rank, world=world, seed=seed))
        elif world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self, num_replicas=world, rank=rank,
                                                                      shuffle=True, seed=seed)
            self.loader = ForkSafeDataLoader(
                self, batch_size=config.training_batch_size, num_workers=_loader_workers(), sampler=sampler,
                collate_fn=collate, pin_memory=pin)
        else:
            self.loader = ForkSafeDataLoader(
                self, batch_size=config.training_batch_size, num_workers=_loader_workers(), shuffle=True,
                collate_fn=collate, pin_memory=pin)

    def __len__(self):
        return len(self._paths) * self.upsample_factor

    def __getitem__(self, idx):
        idx = idx % len(self._paths)
        x, _fs = read_wav(self._paths[idx], keep_pcm16=pcm16_batches())
        if self.seq2seq:                         # <sos> + the characters of the semantics string + <eos> (data.py:322-326)
            y_intent = [self._index["<sos>"]] + [self._index[c] for c in self._values[idx]] + [self._index["<eos>"]]
            return x, y_intent
        # a slot value that never occurs in the training split raises KeyError here, as in the reference
        y_intent = [self.Sy_intent[slot][v] for slot, v in zip(("action", "object", "location"), self._values[idx])]
        return x, y_intent


class SyntheticSLUDataset(SLUDataset):
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


SYNTHETIC_SEQ2SEQ_LABELS = ["<sos>"] + list("abcdefghijklmnopqrstuvwxyz {}:'|,_") + ["<eos>"]


class SyntheticSeq2SeqDataset(SLUDataset):
    """Synthetic seq2seq SLU dataset: `.loader` yields (x (B,T) float32, y (B,U,num_labels) one-hot float32) like
    CollateWavsSLU's seq2seq branch (reference data.py:363-376): noise waveforms, random label strings of 3..U-2
    characters between <sos> and <eos>, padded with <eos>."""

    def __init__(self, num_batches, batch_size, num_samples, max_len=24, seed=1234, Sy_intent=None):
        g = torch.Generator().manual_seed(seed)
        self.Sy_intent = Sy_intent or list(SYNTHETIC_SEQ2SEQ_LABELS)
        self.seq2seq = True
        V = len(self.Sy_intent)
        sos, eos = self.Sy_intent.index("<sos>"), self.Sy_intent.index("<eos>")
        inner = [i for i in range(V) if i not in (sos, eos)]
        batches = []
        for _ in range(num_batches):
            x = 0.1 * torch.randn(batch_size, num_samples, generator=g)
            idx = torch.full((batch_size, max_len), eos, dtype=torch.int64)
            idx[:, 0] = sos
            lens = torch.randint(3, max_len - 1, (batch_size,), generator=g)
            pick = torch.randint(0, len(inner), (batch_size, max_len), generator=g)
            for b in range(batch_size):
                n = int(lens[b])
                idx[b, 1:1 + n] = torch.tensor([inner[int(k)] for k in pick[b, :n]])
            batches.append((x, one_hot(idx, V)))
        self.batches = batches
        self.loader = _SyntheticLoader(batches)

    def __len__(self):
        return sum(len(b[0]) for b in self.batches)

    def __getitem__(self, idx):
        bs = len(self.batches[0][0])
        x, y = self.batches[idx // bs]
        return x[idx % bs], y[idx % bs]


def read_textgrid(path):
    """Interval tiers of a Praat TextGrid (the long "ooTextFile" format the Montreal Forced Aligner writes
    for LibriSpeech): {tier name: [(minTime, maxTime, mark), ...]}; the first tier of a name wins, like
    `textgrid.TextGrid().getList(name)[0]` in the reference (data.py:478-497).  Point tiers are skipped."""
    import re
    tiers, cur, t0, t1, is_interval = {}, None, None, None, False
    num = r"([-+0-9.eE]+)"
    with open(path, "r", encoding="utf-8", errors="replace") as f:
        for raw in f:
            ln = raw.strip()
            m = re.match(r'class = "(\w+)"', ln)
            if m:
                is_interval = m.group(1) == "IntervalTier"
                cur = None
                continue
            m = re.match(r'name = "(.*)"$', ln)
            if m and is_interval:
                name = m.group(1)
                cur = tiers.setdefault(name, []) if name not in tiers else []    # later duplicates are ignored
                continue
            if cur is None:
                continue
            m = re.match(r"xmin = " + num, ln)
            if m:
                t0 = float(m.group(1))
                continue
            m = re.match(r"xmax = " + num, ln)
            if m:
                t1 = float(m.group(1))
                continue
            m = re.match(r'text = "(.*)"$', ln)
            if m:
                cur.append((t0, t1, m.group(1).replace('""', '"')))
    return tiers


def _strip_stress(mark):
    return mark.rstrip("0123456789")


class CollateWavsASR:
    """list of (waveform, phoneme labels, word labels) -> (x (B,T_max) float32 zero-padded, y_phoneme
    (B,U_p) int64, y_word (B,U_w) int64, both padded with the ignore index -1) — reference data.py:511-545.
    Assembled in one buffer per tensor (pinned by the DataLoader)."""

    def __call__(self, batch):
        n = len(batch)
        T = max(len(b[0]) for b in batch)
        Up = max(len(b[1]) for b in batch)
        Uw = max(len(b[2]) for b in batch)
        x = _pad_waveforms([b[0] for b in batch], T)
        yp = torch.full((n, Up), -1, dtype=torch.int64)
        yw = torch.full((n, Uw), -1, dtype=torch.int64)
        for i, (xi, pi, wi) in enumerate(batch):
            yp[i, :len(pi)] = torch.as_tensor(np.asarray(pi, dtype=np.int64))
            yw[i, :len(wi)] = torch.as_tensor(np.asarray(wi, dtype=np.int64))
        return x, yp, yw


class ASRDataset(torch.utils.data.Dataset):
    """LibriSpeech + forced-alignment dataset of ASR pre-training (reference data.py:453-509): item =
    a random snippet of the utterance (length ~ N(mean, var) seconds, at least 0.5 s; torch's global RNG,
    consumed in the reference's order) with one phoneme label per `phone_downsample_factor` samples and
    one word label per `word_downsample_factor` samples; -1 = no label (silence / out-of-vocabulary).
    Label lookup is a dictionary (first index wins, like list.index) and the per-sample label tracks are
    numpy repeats instead of Python lists."""

    def __init__(self, wav_paths, textgrid_paths, Sy_phoneme, Sy_word, config, shard=False):
        self.wav_paths = wav_paths
        self.textgrid_paths = textgrid_paths
        self.length_mean = config.pretraining_length_mean
        self.length_var = config.pretraining_length_var
        self.Sy_phoneme = Sy_phoneme
        self.Sy_word = Sy_word
        self.phone_downsample_factor = config.phone_downsample_factor
        self.word_downsample_factor = config.word_downsample_factor
        self._phone_idx, self._word_idx = {}, {}
        for i, v in enumerate(Sy_phoneme):
            self._phone_idx.setdefault(v, i)
        for i, v in enumerate(Sy_word):
            self._word_idx.setdefault(v, i)
        rank, world = _world() if shard else (0, 1)          # training split under data parallelism
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(self, num_replicas=world, rank=rank,
                                                                      shuffle=True, seed=getattr(config, "seed", 0))
            self.loader = ForkSafeDataLoader(
                self, batch_size=config.pretraining_batch_size, num_workers=_loader_workers(), sampler=sampler,
                collate_fn=CollateWavsASR(), pin_memory=torch.cuda.is_available())
        else:
            self.loader = ForkSafeDataLoader(
                self, batch_size=config.pretraining_batch_size, num_workers=_loader_workers(), shuffle=True,
                collate_fn=CollateWavsASR(), pin_memory=torch.cuda.is_available())

    def __len__(self):
        return len(self.wav_paths)

    @staticmethod
    def _track(intervals, index_of, fs):
        idx = np.array([index_of(mark) for _, _, mark in intervals], dtype=np.int64)
        reps = np.array([max(0, round((t1 - t0) * fs)) for t0, t1, _ in intervals], dtype=np.int64)
        return np.repeat(idx, reps)

    def __getitem__(self, idx):
        x32, fs = read_wav(self.wav_paths[idx])
        x = x32.astype(np.float64)                     # soundfile.read hands the reference float64
        tg = read_textgrid(self.textgrid_paths[idx])
        y_phoneme = self._track(tg["phones"], lambda m: -1 if m == "" else self._phone_idx.get(_strip_stress(m), -1), fs)
        y_word = self._track(tg["words"], lambda m: self._word_idx.get(m, -1), fs)
        random_length = round(fs * max(self.length_mean + self.length_var * torch.randn(1).item(), 0.5))
        if len(x) <= random_length:
            start = 0
        else:
            start = torch.randint(low=0, high=len(x) - random_length, size=(1,)).item()
        end = start + random_length
        return (x[start:end], y_phoneme[start:end:self.phone_downsample_factor].tolist(),
                y_word[start:end:self.word_downsample_factor].tolist())


class SyntheticASRDataset(ASRDataset):
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


def _select_rows(df, count):
    """`df.loc[np.random.choice(len(df), count, replace=False)]` of the reference (data.py:174,178):
    label-based, which equals positional selection on a freshly read csv.  After a speaker subset of the
    same csv the index has gaps and the reference raises KeyError; here the rows are then taken by
    position (documented superset)."""
    pick = np.random.choice(len(df), count, replace=False)
    if df.index.is_unique and set(pick).issubset(set(df.index)):
        return df.loc[pick]
    return df.iloc[pick]


def _speaker_subset(df, fraction):
    speakers = np.array(list(Counter(df.speakerId)))
    np.random.shuffle(speakers)
    selected = speakers[:round(fraction * len(speakers))]
    return df[df["speakerId"].isin(selected)]


def _read_phoneme_count(config):
    phonemes = os.path.join(config.folder, "pretraining", "phonemes.txt")
    if os.path.isfile(phonemes):
        with open(phonemes) as f:
            return len([ln for ln in f.read().split("\n") if ln != ""])
    print("No phoneme file found.")
    return None


def get_SLU_datasets(config):
    """(train, valid, test) SLU datasets; also sets config.values_per_slot / Sy_intent /
    num_phonemes (reference data.py:132-240).  `slu_path` is a Fluent-Speech-Commands tree
    (data/{synthetic,train,valid,test}_data.csv + wavs) or `synthetic[:NxBxT]`.  The numpy global RNG is
    consumed in the reference's order (speaker shuffles, then row choices), so a seeded run selects the
    same subsets."""
    import pandas as pd
    spec = _synthetic_spec(config.slu_path)
    if spec is not None:
        nb, bs, ns = spec
        bs = bs or config.training_batch_size
        config.values_per_slot = list(FSC_VALUES_PER_SLOT)
        config.Sy_intent = synthetic_Sy_intent(config.values_per_slot)
        n_ph = _read_phoneme_count(config)
        config.num_phonemes = 42 if n_ph is None else n_ph
        mk = lambda n, seed: SyntheticSLUDataset(n, bs, ns, config.values_per_slot, seed=seed,
                                                 Sy_intent=config.Sy_intent)
        rank = _world()[0]                                  # every rank its own training batches
        if config.seq2seq:
            config.Sy_intent = list(SYNTHETIC_SEQ2SEQ_LABELS)
            mk = lambda n, seed: SyntheticSeq2SeqDataset(n, bs, ns, seed=seed, Sy_intent=config.Sy_intent)
        return (mk(nb, config.seed + 1000003 * rank), mk(max(1, nb // 4), config.seed + 1),
                mk(max(1, nb // 4), config.seed + 2))
    base_path = config.slu_path
    sfx = "_seq2seq" if config.seq2seq else ""          # reference data.py:139-146, 182-187
    csv = lambda name: pd.read_csv(os.path.join(base_path, "data", name.replace(".csv", sfx + ".csv")))
    synthetic_train_df = csv("synthetic_data.csv")
    real_train_df = csv("train_data.csv")
    have_spk = "speakerId" in list(real_train_df) and "speakerId" in list(synthetic_train_df)
    if have_spk:
        if config.real_speaker_subset_percentage < 1:
            real_train_df = _speaker_subset(real_train_df, config.real_speaker_subset_percentage)
        if config.synthetic_speaker_subset_percentage < 1:
            synthetic_train_df = _speaker_subset(synthetic_train_df, config.synthetic_speaker_subset_percentage)
    else:
        if "speakerId" in list(real_train_df):
            real_train_df = real_train_df.drop(columns="speakerId")
        if "speakerId" in list(synthetic_train_df):
            synthetic_train_df = synthetic_train_df.drop(columns="speakerId")
        for frac in (config.real_speaker_subset_percentage, config.synthetic_speaker_subset_percentage):
            if frac < 1:
                print("no speaker id listed in dataset .csv; ignoring speaker subset selection")
    if config.real_dataset_subset_percentage < 1:
        real_train_df = _select_rows(real_train_df, round(config.real_dataset_subset_percentage * len(real_train_df)))
    if config.synthetic_dataset_subset_percentage < 1:
        synthetic_train_df = _select_rows(
            synthetic_train_df, round(config.synthetic_dataset_subset_percentage * len(synthetic_train_df)))
    train_df = pd.concat([synthetic_train_df, real_train_df]).reset_index()
    valid_df = csv("valid_data.csv")
    test_df = csv("test_data.csv")

    if not config.seq2seq:
        Sy_intent = {"action": {}, "object": {}, "location": {}}
        values_per_slot = []
        for slot in ["action", "object", "location"]:
            slot_values = Counter(train_df[slot])          # first-appearance order, like the reference
            for idx, value in enumerate(slot_values):
                Sy_intent[slot][value] = idx
            values_per_slot.append(len(slot_values))
        config.values_per_slot = values_per_slot
    else:
        # output alphabet: <sos>, every character of the training semantics + string.printable, <eos> (reference
        # data.py:201-208).  The reference orders the characters by iterating a Python set — an order that changes
        # from process to process (string hashing is salted), so its checkpoints only decode within one process;
        # here the set is SORTED: the same alphabet, a reproducible index assignment.
        import string
        chars = set("".join(str(v) for v in train_df["semantics"].tolist()) + string.printable)
        Sy_intent = ["<sos>"] + sorted(chars) + ["<eos>"]
    config.Sy_intent = Sy_intent

    def wordings(path):
        with open(path, "r") as f:
            return [line.strip() for line in f.readlines()]

    if config.train_wording_path is not None:
        train_df = train_df.loc[train_df.transcription.isin(wordings(config.train_wording_path))]
        train_df = train_df.set_index(np.arange(len(train_df)))
    if config.test_wording_path is not None:
        keep = wordings(config.test_wording_path)
        valid_df = valid_df.loc[valid_df.transcription.isin(keep)]
        valid_df = valid_df.set_index(np.arange(len(valid_df)))
        test_df = test_df.loc[test_df.transcription.isin(keep)]
        test_df = test_df.set_index(np.arange(len(test_df)))

    n_ph = _read_phoneme_count(config)
    if n_ph is not None:
        config.num_phonemes = n_ph
    train_dataset = SLUDataset(train_df, base_path, Sy_intent, config, upsample_factor=config.dataset_upsample_factor,
                               shard=True)
    valid_dataset = SLUDataset(valid_df, base_path, Sy_intent, config)
    test_dataset = SLUDataset(test_df, base_path, Sy_intent, config)
    return train_dataset, valid_dataset, test_dataset


def get_ASR_datasets(config):
    """(train, valid, test) ASR pre-training datasets (reference data.py:393-451).  `asr_path` holds
    `text/<split>*/<speaker>/<chapter>/<utt>.TextGrid` alignments and the matching `audio/...wav` files
    (splits train* / dev* / test*), or is `synthetic[:NxBxT]`.  The phoneme / word vocabularies are read
    from <folder>/pretraining/{phonemes,words}.txt or built from the dev split (phonemes in first-seen
    order with stress digits stripped, the `vocabulary_size` most common words) and written there."""
    spec = _synthetic_spec(config.asr_path)
    if spec is not None:
        nb, bs, ns = spec
        bs = bs or config.pretraining_batch_size
        config.num_phonemes = 42
        mk = lambda n, seed: SyntheticASRDataset(n, bs, ns, config, seed=seed)
        rank = _world()[0]
        return (mk(nb, config.seed + 1000003 * rank), mk(max(1, nb // 4), config.seed + 1),
                mk(max(1, nb // 4), config.seed + 2))
    import glob
    base_path = config.asr_path
    wavs = lambda tgs: [p.replace("text", "audio").replace(".TextGrid", ".wav") for p in tgs]
    train_tg = glob.glob(base_path + "/text/train*/*/*/*.TextGrid")
    valid_tg = glob.glob(base_path + "/text/dev*/*/*/*.TextGrid")
    test_tg = glob.glob(base_path + "/text/test*/*/*/*.TextGrid")
    ph_file = os.path.join(config.folder, "pretraining", "phonemes.txt")
    wd_file = os.path.join(config.folder, "pretraining", "words.txt")
    rank, world = _world()
    if world > 1:
        # identical file order on every rank (DistributedSampler indexes into it); one rank builds the
        # vocabulary files, the others read them
        train_tg, valid_tg, test_tg = sorted(train_tg), sorted(valid_tg), sorted(test_tg)
        import torch.distributed as dist
        if rank != 0:
            dist.barrier()
    if os.path.isfile(ph_file) and os.path.isfile(wd_file):
        with open(ph_file, "r") as f:
            Sy_phoneme = [ln.rstrip("\n") for ln in f.readlines() if ln.rstrip("\n") != ""]
        with open(wd_file, "r") as f:
            Sy_word = [ln.rstrip("\n") for ln in f.readlines()]
        config.num_phonemes = len(Sy_phoneme)
    else:
        print("Getting vocabulary...")
        phoneme_counter, word_counter = Counter(), Counter()
        for path in valid_tg:
            tg = read_textgrid(path)
            phoneme_counter.update([_strip_stress(m) for _, _, m in tg["phones"] if m != ""])
            word_counter.update([m for _, _, m in tg["words"]])
        Sy_phoneme = list(phoneme_counter)
        Sy_word = [w[0] for w in word_counter.most_common(config.vocabulary_size)]
        config.num_phonemes = len(Sy_phoneme)
        with open(ph_file, "w") as f:
            for phoneme in Sy_phoneme:
                f.write(phoneme + "\n")
        with open(wd_file, "w") as f:
            for word in Sy_word:
                f.write(word + "\n")
    if world > 1 and rank == 0:
        dist.barrier()
    print("Done.")
    mk = lambda tgs, shard=False: ASRDataset(wavs(tgs), tgs, Sy_phoneme, Sy_word, config, shard=shard)
    return mk(train_tg, True), mk(valid_tg), mk(test_tg)
