"""Builds a tiny Fluent-Speech-Commands-shaped directory (CSV splits + PCM16 wavs) deterministically.
Shared by tests/golden/make_goldens.py (which runs the REFERENCE's data.py on it) and
tests/test_data_real.py (which runs this package's data.py on an identical copy)."""
import os

import numpy as np
import pandas as pd
from scipy.io import wavfile

ACTIONS = ["activate", "deactivate", "increase", "decrease", "change language", "bring"]
OBJECTS = ["lights", "music", "volume", "heat", "lamp", "none", "newspaper", "shoes"]
LOCATIONS = ["kitchen", "bedroom", "washroom", "none"]
PHRASES = ["turn on the lights", "switch off the lamp", "louder please", "bring me my shoes", "make it hotter",
           "lights off in the kitchen", "play some music"]


def _rows(rs, split, n, speakers, with_speaker):
    rows = []
    for k in range(n):
        spk = speakers[rs.randint(len(speakers))]
        row = {"path": "wavs/speakers/%s/%s_%03d.wav" % (spk, split, k)}
        if with_speaker:
            row["speakerId"] = spk
        row["transcription"] = PHRASES[rs.randint(len(PHRASES))]
        row["action"] = ACTIONS[rs.randint(len(ACTIONS))]
        row["object"] = OBJECTS[rs.randint(len(OBJECTS))]
        row["location"] = LOCATIONS[rs.randint(len(LOCATIONS))]
        rows.append(row)
    return pd.DataFrame(rows)


def semantics_of(row):
    """The seq2seq label string of a row (the *_seq2seq.csv splits of the reference carry a `semantics` column)."""
    return "{'action': '%s', 'object': '%s', 'location': '%s'}" % (row["action"], row["object"], row["location"])


def make_fsc_tree(root, seed=0, with_speaker=True, sizes=(23, 11, 7, 5), seq2seq=False):
    """root/data/{train,synthetic,valid,test}_data.csv + root/wavs/... ; returns {split: DataFrame}.
    seq2seq: also write the *_data_seq2seq.csv splits (same rows + a `semantics` column)."""
    rs = np.random.RandomState(seed)
    speakers = ["spk%02d" % i for i in range(6)]
    os.makedirs(os.path.join(root, "data"), exist_ok=True)
    out = {}
    for split, n in zip(("train", "synthetic", "valid", "test"), sizes):
        df = _rows(rs, split, n, speakers, with_speaker)
        df.to_csv(os.path.join(root, "data", "%s_data.csv" % split))        # FSC csvs carry an unnamed index column
        if seq2seq:
            d2 = df.copy()
            d2["semantics"] = [semantics_of(r) for _, r in df.iterrows()]
            d2.to_csv(os.path.join(root, "data", "%s_data_seq2seq.csv" % split))
        for p in df.path:
            full = os.path.join(root, p)
            os.makedirs(os.path.dirname(full), exist_ok=True)
            n_samp = int(rs.randint(900, 2400))
            pcm = (rs.randn(n_samp) * 3000).clip(-32768, 32767).astype(np.int16)
            wavfile.write(full, 16000, pcm)
        out[split] = df
    with open(os.path.join(root, "train_wordings.txt"), "w") as f:
        f.write("\n".join(PHRASES[:4]) + "\n")
    with open(os.path.join(root, "test_wordings.txt"), "w") as f:
        f.write("\n".join(PHRASES[2:6]) + "\n")
    return out


VARIANTS = {
    # name: (config overrides, numpy seed set right before get_SLU_datasets, tree kwargs)
    "default": ({}, 0, {}),
    # (the reference raises KeyError when a speaker subset and a dataset subset hit the SAME csv: the
    #  second selection is label-based on an index that has gaps by then)
    "subsets": ({"real_speaker_subset_percentage": 0.5, "synthetic_dataset_subset_percentage": 0.5}, 7, {}),
    "real_subset": ({"real_dataset_subset_percentage": 0.6}, 9, {}),
    "wordings": ({"train_wording_path": "train_wordings.txt", "test_wording_path": "test_wordings.txt",
                  "dataset_upsample_factor": 3}, 0, {}),
    "no_speaker_column": ({"real_speaker_subset_percentage": 0.5}, 3, {"with_speaker": False}),
}


# ------------------------------------------------------------------------------------------------
# LibriSpeech-shaped tree with Montreal-Forced-Aligner style TextGrids (long format)
# ------------------------------------------------------------------------------------------------
WORDS = ["the", "cat", "sat", "on", "mat", "a", "dog", "ran", "fast", "slow", "", "blue", "sky"]
PHONES = ["DH", "AH0", "K", "AE1", "T", "S", "AA1", "N", "M", "EY1", "D", "AO1", "G", "R", "F", "L", "OW1", "", "sil"]


def _tier(name, total, marks_and_ends):
    out = ['    item [%d]:' % (1 if name == "words" else 2), '        class = "IntervalTier" ',
           '        name = "%s" ' % name, '        xmin = 0 ', '        xmax = %s ' % repr(total),
           '        intervals: size = %d ' % len(marks_and_ends)]
    t0 = 0.0
    for k, (mark, t1) in enumerate(marks_and_ends):
        out += ['        intervals [%d]:' % (k + 1), '            xmin = %s ' % repr(t0),
                '            xmax = %s ' % repr(t1), '            text = "%s" ' % mark]
        t0 = t1
    return out


def write_textgrid(path, total, words, phones):
    lines = ['File type = "ooTextFile"', 'Object class = "TextGrid"', '', 'xmin = 0 ', 'xmax = %s ' % repr(total),
             'tiers? <exists> ', 'size = 2 ', 'item []: '] + _tier("words", total, words) + _tier("phones", total, phones)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def _segments(rs, total, marks, n):
    cuts = np.sort(rs.uniform(0.02, total - 0.02, size=n - 1)).round(2).tolist() + [total]
    return [(marks[rs.randint(len(marks))], float(c)) for c in cuts]


def make_asr_tree(root, seed=0, counts=(5, 4, 3)):
    """root/asr/{text,audio}/{train-clean,dev-clean,test-clean}/<spk>/<chapter>/<utt>.{TextGrid,wav}"""
    rs = np.random.RandomState(seed)
    base = os.path.join(root, "asr")
    for split, n in zip(("train-clean", "dev-clean", "test-clean"), counts):
        for k in range(n):
            spk, chap = "%d" % (100 + rs.randint(3)), "%d" % (7 + rs.randint(2))
            utt = "%s-%s-%04d" % (spk, chap, k)
            tdir = os.path.join(base, "text", split, spk, chap)
            adir = os.path.join(base, "audio", split, spk, chap)
            os.makedirs(tdir, exist_ok=True)
            os.makedirs(adir, exist_ok=True)
            total = float(np.round(rs.uniform(0.9, 2.2), 2))
            n_samp = int(round(total * 16000))
            pcm = (rs.randn(n_samp) * 2500).clip(-32768, 32767).astype(np.int16)
            wavfile.write(os.path.join(adir, utt + ".wav"), 16000, pcm)
            write_textgrid(os.path.join(tdir, utt + ".TextGrid"), total,
                           _segments(rs, total, WORDS, 2 + rs.randint(4)), _segments(rs, total, PHONES, 5 + rs.randint(8)))
    return base
