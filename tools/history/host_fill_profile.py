#!/usr/bin/env python3
"""cProfile of the HOST side of a short pipelined run (what happens before the first super-batch is launched).
python tools/host_fill_profile.py"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
import bench

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
for _ in range(4):
    bench.run_steps(model, trainer, batches, 20)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    bench.run_steps(model, trainer, batches, 20)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("run_steps(20): host returns after %.0f us, device done after %.0f us" % (1e6 * (t1 - t0), 1e6 * (t2 - t0)))
pr = cProfile.Profile()
pr.enable()
bench.run_steps(model, trainer, batches, 20)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
