#!/usr/bin/env python3
"""CPU-side timing of the look-ahead loop: how long do PrefixSlot.run / StepGraph.run calls take on the
host inside the real pipelined loop (a blocked call shows up as an outlier)?"""
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
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        t0 = time.perf_counter(); r = orig(self, *a, **k); log.append((name + ":" + cls.__name__, t0, time.perf_counter())); return r
    setattr(cls, name, f)
wrap(pipeline.PrefixSlot, "run"); wrap(pipeline.StepGraph, "run")
t0 = time.perf_counter(); bench.run_steps(model, trainer, batches, 56); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue loop %.1f us/step, drain after loop %.1f us" % ((t1 - t0) / 56 * 1e6, (t2 - t1) * 1e6))
for n, a, b in log[:40]:
    print("%10.1f  %-22s %8.1f us" % ((a - t0) * 1e6, n, (b - a) * 1e6))
