#!/usr/bin/env python3
"""Round 6: do the two halves of the pipelined loop slow each other down?  The captured frozen prefix (one super-batch graph,
look-ahead partition) and the captured trainable suffix (step graph, training partition) timed (a) alone, (b) side by side with
no dependency between them — each stream between its own pair of HIP events — and (c) the real pipelined loop.
    python tools/interference_probe.py [--lookahead 20]"""
import argparse, os, sys, time, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lookahead", type=int, default=20)
a = ap.parse_args()
os.environ["SLU_LOOKAHEAD"] = str(a.lookahead)
import bench

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, a.batch, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
for _ in range(3):
    bench.run_steps(model, trainer, batches, 4 * a.lookahead)
torch.cuda.synchronize()
P = a.lookahead
sg = next(iter(trainer._step_graphs.values()))
main = trainer._train_stream
slot = trainer._slots[0]
graph, x_static, feats = next(v for v in slot.graphs.values() if v is not None)


def clocks():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l)[:200]
    except Exception as e:
        return "rocm-smi: %s" % e


def run(n_prefix, n_suffix, sample_clocks=False):
    """n_prefix super-batch replays and n_suffix step replays, enqueued together; -> (ms per super-batch, ms per step, wall ms)"""
    torch.cuda.synchronize()
    ep0, ep1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    es0, es1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if n_prefix:
        with torch.cuda.stream(slot.stream):
            ep0.record(slot.stream)
            for _ in range(n_prefix):
                graph.replay()
            ep1.record(slot.stream)
    if n_suffix:
        with torch.cuda.stream(main):
            es0.record(main)
            for i in range(n_suffix):
                sg.run(sg.inputs, 5000 + i)
            es1.record(main)
    ck = clocks() if sample_clocks else ""
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0)
    return (ep0.elapsed_time(ep1) / n_prefix if n_prefix else 0.0, es0.elapsed_time(es1) / n_suffix if n_suffix else 0.0, wall, ck)


run(2, 40)
pa = run(40, 0, True)
sa = run(0, 40 * P, True)
both = run(40, 40 * P, True)
print("prefix alone        : %.3f ms per super-batch of %d = %.4f ms / step   [%s]" % (pa[0], P, pa[0] / P, pa[3]))
print("suffix alone        : %.4f ms / step   [%s]" % (sa[1], sa[3]))
print("side by side (no dependencies, %d super-batches + %d steps): prefix %.3f ms per super-batch (x %.2f), suffix %.4f ms / step (x %.2f), "
      "wall %.1f ms   [%s]" % (40, 40 * P, both[0], both[0] / pa[0], both[1], both[1] / sa[1], both[2], both[3]))
# the suffix partition kept busy by something cheap: is it the suffix's WORK or the mere presence of a second queue?
idle = torch.zeros(1 << 20, device=dev)
def light(n):
    with torch.cuda.stream(main):
        for _ in range(n):
            idle.add_(1.0)
torch.cuda.synchronize()
ep0, ep1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(slot.stream):
    ep0.record(slot.stream)
    for _ in range(20):
        graph.replay()
    ep1.record(slot.stream)
light(20000)
torch.cuda.synchronize()
print("prefix beside a stream of tiny elementwise launches on the training partition: %.3f ms per super-batch (x %.2f)" % (ep0.elapsed_time(ep1) / 20, ep0.elapsed_time(ep1) / 20 / pa[0]))
t0 = time.perf_counter(); bench.run_steps(model, trainer, batches, 40 * P); torch.cuda.synchronize()
print("real pipelined loop : %.4f ms / step" % (1e3 * (time.perf_counter() - t0) / (40 * P)))
