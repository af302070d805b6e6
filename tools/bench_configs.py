#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations that are not the headline bench line:
  configs[1]  SincNet + Conv1d front end (phoneme module up to conv2), forward only, B = 64 x 3 s:
              utterances/s and the HBM GB/s the algorithmic bytes correspond to
  configs[2]  full PretrainedModel (Sinc + conv + 4-layer biGRU + both ASR heads) forward/backward + Adam,
              B = 64 x 3 s, fp32 (ASR pre-training step through training.Trainer)
Synthetic inputs resident in HBM; one JSON line per configuration."""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-slu_amd")
sys.path.insert(0, PKG)
import torch
import data, models, training

B, S = 64, 48000
work = tempfile.mkdtemp(prefix="slu_cfgs_")
os.makedirs(os.path.join(work, "experiments"))
name = "unfreeze_all_layers_synthetic.cfg"
shutil.copy(os.path.join(PKG, "experiments", name), os.path.join(work, "experiments", name))
cwd = os.getcwd(); os.chdir(work)
config = data.read_config(os.path.join("experiments", name))
config.folder = os.path.join(work, config.folder)
config.asr_path = "synthetic:4x%dx%d" % (B, S)
train_ds, _, _ = data.get_ASR_datasets(config)
os.chdir(cwd)
torch.manual_seed(1)
pm = models.PretrainedModel(config)
dev = next(pm.parameters()).device
batches = [tuple(t.to(dev) for t in b) for b in train_ds.loader]

def timed(fn, n, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# configs[1]: front end forward only (sinc -> abs -> pool -> LeakyReLU -> conv1 -> conv2), eval mode
pm.eval()
x = batches[0][0]
stages = pm._stages()
n_cnn = sum(1 for st in stages if type(st).__name__ == "_ConvStage")
with torch.no_grad():
    dt = timed(lambda: pm.run_stages(x, 0, n_cnn), 200, 20)
    out = pm.run_stages(x, 0, n_cnn)
alg_bytes = x.numel() * 4 + out.numel() * 4                  # waveform in, conv2 activations out
flops = B * (38.496e6 + 14.4e6 + 10.8e6)
print(json.dumps({"config": "configs[1] SincNet+Conv1d front end, forward only, B=64 x 3 s", "ms": round(dt * 1e3, 4),
                  "utterances_per_s": round(B / dt, 1), "algorithmic_GB_per_s": round(alg_bytes / dt / 1e9, 1),
                  "tflops_fp32_mfma": round(flops / dt / 1e12, 2),
                  "note": "compute-bound stage (134 / 86 FLOP per byte): the HBM figure is far below 8 TB/s by construction"}))

# configs[2]: ASR pre-training step (forward, both CE heads, backward, Adam)
pm.train()
models.set_dropout_seed(3)
trainer = training.Trainer(pm, config)
def epoch():
    for _ in trainer._iterate(batches, True, True):
        pass
dt = timed(epoch, 8, 2) / len(batches)
print(json.dumps({"config": "configs[2] full PretrainedModel fwd/bwd + Adam (ASR pre-training step, phoneme + word heads, "
                            "vocabulary %d), B=64 x 3 s, fp32; GRU hidden 128 as in the reference's cfgs" % config.vocabulary_size,
                  "ms_per_step": round(dt * 1e3, 3), "utterances_per_s": round(B / dt, 1)}))
shutil.rmtree(work, ignore_errors=True)
