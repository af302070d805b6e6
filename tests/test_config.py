"""CPU: data.read_config against the reference's own parser on all 29 experiment cfgs (fixture
g8_config.json = cfg text in, vars(Config) or the reference's exception out)."""
import configparser
import json
import os

import pytest

G8 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_config.json")))
LEGACY = sorted(k for k, v in G8.items() if "error" in v)
PARSEABLE = sorted(k for k, v in G8.items() if "expected" in v)


def _write(tmp_path, name):
    os.makedirs(tmp_path / "experiments", exist_ok=True)
    p = tmp_path / "experiments" / name
    p.write_text(G8[name]["text"])
    return p


def test_fixture_census():
    assert len(G8) == 29 and len(LEGACY) == 9 and len(PARSEABLE) == 20
    for k in LEGACY:
        assert G8[k]["error_type"] == "NoOptionError" and "real_dataset_subset_percentage" in G8[k]["error"]


@pytest.mark.parametrize("name", PARSEABLE)
def test_read_config_matches_reference(name, tmp_path, monkeypatch, capsys):
    import data
    monkeypatch.chdir(tmp_path)
    _write(tmp_path, name)
    cfg = data.read_config("experiments/" + name)
    assert vars(cfg) == G8[name]["expected"]
    assert capsys.readouterr().out == G8[name]["stdout"]
    # side effects of the reference: experiment folder, sub-folders and a copy of the cfg
    assert os.path.isdir(os.path.join(cfg.folder, "pretraining")) and os.path.isdir(os.path.join(cfg.folder, "training"))
    assert open(os.path.join(cfg.folder, "experiment.cfg")).read() == G8[name]["text"]


@pytest.mark.parametrize("name", LEGACY)
def test_legacy_cfgs_parse_with_documented_mapping(name, tmp_path, monkeypatch):
    """The reference raises NoOptionError on these nine; this package accepts them
    (dataset_subset_percentage -> real_dataset_subset_percentage, the other three default 1.0)."""
    import data
    monkeypatch.chdir(tmp_path)
    _write(tmp_path, name)
    cfg = data.read_config("experiments/" + name)
    parser = configparser.ConfigParser()
    parser.read_string(G8[name]["text"])
    assert cfg.real_dataset_subset_percentage == float(parser.get("training", "dataset_subset_percentage"))
    assert cfg.synthetic_dataset_subset_percentage == 1.0 and cfg.real_speaker_subset_percentage == 1.0
    assert cfg.cnn_N_filt == [80, 60, 60] and cfg.phone_rnn_num_hidden == [128, 128]
    assert cfg.phone_downsample_factor == 640 and cfg.word_downsample_factor == 2560


def test_missing_required_key_raises_like_the_reference(tmp_path, monkeypatch):
    import data
    monkeypatch.chdir(tmp_path)
    name = PARSEABLE[0]
    text = G8[name]["text"].replace("training_lr", "training_lrx")
    os.makedirs(tmp_path / "experiments")
    (tmp_path / "experiments" / "bad.cfg").write_text(text)
    with pytest.raises(configparser.NoOptionError):
        data.read_config("experiments/bad.cfg")
    text = G8[name]["text"].replace("real_dataset_subset_percentage", "xreal_dataset_subset_percentage")
    (tmp_path / "experiments" / "bad2.cfg").write_text(text)
    with pytest.raises(configparser.NoOptionError):
        data.read_config("experiments/bad2.cfg")
