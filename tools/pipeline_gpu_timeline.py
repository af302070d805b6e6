#!/usr/bin/env python3
"""GPU-side timeline of the look-ahead loop from HIP events (no profiler): start/end of every
StepGraph.run on the train stream and of every PrefixSlot.run on its side stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
import bench
from slu_hip import pipeline

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
bench.run_steps(model, trainer, batches, 64)
torch.cuda.synchronize()
log = []
orig_sg = pipeline.StepGraph.run
def sg_run(self, *a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig_sg(self, *a, **k); e1.record(); log.append(("step", e0, e1)); return r
pipeline.StepGraph.run = sg_run
orig_ps = pipeline.PrefixSlot.run
def ps_run(self, *a, **k):
    e0 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.stream):
        e0.record()
    feats, done = orig_ps(self, *a, **k)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(self.stream):
        e1.record()
    log.append(("prefix", e0, e1)); return feats, done
pipeline.PrefixSlot.run = ps_run
ref = torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(trainer._train_stream):
    ref.record()
bench.run_steps(model, trainer, batches, 42)
torch.cuda.synchronize()
for n, a, b in log:
    print("%-7s start %9.1f  end %9.1f  dur %8.1f us" % (n, ref.elapsed_time(a) * 1e3, ref.elapsed_time(b) * 1e3, a.elapsed_time(b) * 1e3))
