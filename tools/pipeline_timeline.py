#!/usr/bin/env python3
"""GPU-side timeline of the pipelined loop from HIP events (no profiler attached): when each super-batch's
frozen prefix starts / ends on its side stream and when each group's first / last trainable step starts /
ends on the main stream.    python tools/pipeline_timeline.py [--lookahead 24] [--groups 6]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lookahead", type=int, default=20)
ap.add_argument("--groups", type=int, default=6)
ap.add_argument("--steps", type=int, default=0, help="run exactly this many steps (default: groups x lookahead)")
a = ap.parse_args()
os.environ["SLU_LOOKAHEAD"] = str(a.lookahead)
import bench
from slu_hip import pipeline

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, a.batch, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
P = a.lookahead
for _ in range(3):
    bench.run_steps(model, trainer, batches, a.steps or 4 * P)
torch.cuda.synchronize()

ev = lambda: torch.cuda.Event(enable_timing=True)
prefix_marks, step_marks, group_sizes = [], [], []
orig_slot_run = pipeline.PrefixSlot.run
def slot_run(self, model_, xs, n_prefix, step0, use_graph, after=None, **kw):
    e0, e1, e2 = ev(), ev(), ev()
    with torch.cuda.stream(self.stream):
        if self.consumed is not None:
            self.stream.wait_event(self.consumed)
        if after is not None:
            self.stream.wait_event(after)
        e0.record(self.stream)                 # dependencies satisfied: the copies start here
    orig_fill = pipeline.PrefixSlot._fill
    filled = []
    def fill(x_cat, xs_):
        orig_fill(x_cat, xs_)
        e1.record(torch.cuda.current_stream())
        filled.append(1)
    pipeline.PrefixSlot._fill = staticmethod(fill)
    try:
        out = orig_slot_run(self, model_, xs, n_prefix, step0, use_graph, after, **kw)
    finally:
        pipeline.PrefixSlot._fill = staticmethod(orig_fill)
    (self.last_done and self.stream.wait_event(self.last_done))
    e2.record(self.stream)
    prefix_marks.append((e0, e1 if filled else e0, e2))       # batches read in place (row-pointer table): no copies
    group_sizes.append(list(xs))
    return out
pipeline.PrefixSlot.run = slot_run
orig_sg_run = pipeline.StepGraph.run
def sg_run(self, inputs, step):
    e0, e1 = ev(), ev()
    e0.record(torch.cuda.current_stream())
    out = orig_sg_run(self, inputs, step)
    e1.record(torch.cuda.current_stream())
    step_marks.append((e0, e1))
    return out
pipeline.StepGraph.run = sg_run

base = ev()
with torch.cuda.stream(trainer._train_stream):
    base.record(trainer._train_stream)
n_steps = a.steps or a.groups * P
bench.run_steps(model, trainer, batches, n_steps)
torch.cuda.synchronize()
t = lambda e: base.elapsed_time(e) * 1e3
print("all times in us since the loop start; super-batch = %d batches" % P)
for g, (e0, e1, e2) in enumerate(prefix_marks):
    print("prefix %2d: deps ok %9.0f  copies done %9.0f (+%5.0f)  graph done %9.0f (+%6.0f)" %
          (g, t(e0), t(e1), t(e1) - t(e0), t(e2), t(e2) - t(e1)))
widths = [len(xs) for xs in group_sizes]
pos = 0
for g, w in enumerate(widths):
    s = step_marks[pos:pos + w]
    pos += w
    if not s:
        break
    durs = [t(b) - t(a_) for a_, b in s]
    gaps = [t(s[i + 1][0]) - t(s[i][1]) for i in range(len(s) - 1)]
    print("steps  %2d (%2d): first starts %9.0f  last ends %9.0f  (%.0f us / step; step dur min %.0f med %.0f max %.0f; "
          "first %.0f; gap max %.0f)" % (g, w, t(s[0][0]), t(s[-1][1]), (t(s[-1][1]) - t(s[0][0])) / w,
                                          min(durs), sorted(durs)[len(durs) // 2], max(durs), durs[0],
                                          max(gaps) if gaps else 0))
