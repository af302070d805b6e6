#!/usr/bin/env python3
"""Host time of the pieces Trainer._iterate runs before it launches the first look-ahead super-batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch
import bench
import training

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
for _ in range(4):
    bench.run_steps(model, trainer, batches, 20)
torch.cuda.synchronize()
pm = model.pretrained_model

def t(label, fn, n=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print("%-48s %7.1f us" % (label, 1e6 * (time.perf_counter() - t0) / n))

t("trainer._sums_buffer().zero_()", lambda: trainer._sums_buffer().zero_())
t("trainer.lookahead_depth(True, False)", lambda: trainer.lookahead_depth(True, False))
t("next(model.parameters()).device", lambda: next(model.parameters()).device)
t("main.wait_stream(outer)", lambda: trainer._train_stream.wait_stream(torch.cuda.current_stream()))
t("pm.warm_weight_caches()", lambda: pm.warm_weight_caches())
n_prefix = model.frozen_prefix_len()
t("frozen param list + signature", lambda: tuple(p._version for st in pm._stages()[:n_prefix] for p in st.parameters()))
t("slot.stream.wait_stream(main) x2", lambda: [s.stream.wait_stream(trainer._train_stream) for s in trainer._slots])
t("trainer._graphable()", lambda: trainer._graphable())
t("trainer._fused_sums()", lambda: trainer._fused_sums())
t("training._param_signature(model)", lambda: training._param_signature(model))
t("trainer._slu_forward(n_prefix, None)", lambda: trainer._slu_forward(n_prefix, None))
t("len(loader) + iter", lambda: (len(batches), iter(batches)))
from models import next_rng_step
t("12 x next_rng_step()", lambda: [next_rng_step() for _ in range(12)])
from slu_hip import ops
words = trainer._slots[0].words
t("ops.store_u64 (table + rng)", lambda: ops.store_u64(words, [b[0].data_ptr() for b in batches[:4]] * 3 + [0] * 19 + [16]))
torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_steps(model, trainer, batches, 20); t1 = time.perf_counter(); torch.cuda.synchronize()
print("run_steps(20) host time %.0f us" % (1e6 * (t1 - t0)))
