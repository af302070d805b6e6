#!/usr/bin/env python3
"""Where does the pipelined step time go?  Times, on one GPU, (a) the captured trainable suffix of the
step alone (StepGraph replays back to back), (b) the captured frozen prefix alone (super-batch graph
replays back to back) and (c) the pipelined loop, so that overlap quality = (a + b/P) vs (c) is visible.
    python tools/pipeline_split.py [--batch 64] [--lookahead 8]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--lookahead", type=int, default=20)
ap.add_argument("--steps", type=int, default=128)
a = ap.parse_args()
os.environ["SLU_LOOKAHEAD"] = str(a.lookahead)
import bench

config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, a.batch, 48000, 4)
dev = next(model.parameters()).device
batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
for _ in range(3):
    bench.run_steps(model, trainer, batches, 4 * a.lookahead)          # warm-up: captures the graphs
torch.cuda.synchronize()

def wall(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

t_pipe = wall(lambda n: bench.run_steps(model, trainer, batches, n), a.steps)
sg = next(iter(trainer._step_graphs.values()))
main = trainer._train_stream
def suffix(n):
    with torch.cuda.stream(main):
        for i in range(n):
            sg.run(sg.inputs, 1000 + i)
t_suffix = wall(suffix, a.steps)
slot = trainer._slots[0]
(graph, x_static, feats) = next(v for v in slot.graphs.values() if v is not None)
def prefix(n):
    with torch.cuda.stream(slot.stream):
        for i in range(n):
            graph.replay()
t_prefix = wall(prefix, 16)
P = a.lookahead
print("pipelined step        : %8.1f us" % t_pipe)
print("suffix alone (graphs) : %8.1f us / step" % t_suffix)
print("prefix alone (graph)  : %8.1f us / super-batch of %d = %8.1f us / step" % (t_prefix, P, t_prefix / P))
print("sum                   : %8.1f us   (perfect overlap would approach max = %.1f us)" %
      (t_suffix + t_prefix / P, max(t_suffix, t_prefix / P)))

# (d) both at once, no dependencies between them: is the hardware running the two streams concurrently?
def both(n):
    for _ in range(n):
        with torch.cuda.stream(slot.stream):
            graph.replay()
        with torch.cuda.stream(main):
            for i in range(P):
                sg.run(sg.inputs, 2000 + i)
t_both = wall(both, 8)
print("prefix graph + %d suffix steps enqueued together: %8.1f us per group = %6.1f us / step "
      "(serial would be %.1f, ideal overlap %.1f)" % (P, t_both, t_both / P, t_suffix + t_prefix / P,
                                                       max(t_suffix, t_prefix / P)))

# (e) CPU cost of enqueueing (no synchronisation inside the timed region; queues drained before)
def cpu_cost(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("CPU enqueue cost: suffix step %.1f us, prefix super-batch graph %.1f us" %
      (cpu_cost(suffix, 12), cpu_cost(prefix, 4)))
def g1_only(n):
    with torch.cuda.stream(main):
        for i in range(n):
            sg.g1.replay()
def g2_only(n):
    with torch.cuda.stream(main):
        for i in range(n):
            if sg.g2 is not None:
                sg.g2.replay()
print("CPU enqueue cost: G1 replay %.1f us, G2 replay %.1f us" % (cpu_cost(g1_only, 12), cpu_cost(g2_only, 12)))

# (f) is the real loop host-bound?  wall time until the step loop RETURNS (everything enqueued) vs until the GPU is done
def loop_host_vs_gpu(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(model, trainer, batches, n)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
h, g = loop_host_vs_gpu(8 * P)
print("real loop, %d steps: host returns after %.1f us / step, GPU done after %.1f us / step" % (8 * P, h, g))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
bench.run_steps(model, trainer, batches, 4 * P)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

# (g) which ingredient of the real loop makes its steps slower than (d)?  add them one at a time
y_in = sg.inputs[1]
B = a.batch
def variant(consumer, slices, lockstep):
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    slots = trainer._slots
    gs = [next(v for v in s.graphs.values() if v is not None) for s in slots]
    def fn(n):
        done = [None, None]
        consumed = [None, None]
        for g in range(n):
            s = slots[g % 2]
            with torch.cuda.stream(s.stream):
                if lockstep and consumed[g % 2] is not None:
                    s.stream.wait_event(consumed[g % 2])
                if lockstep and done[(g + 1) % 2] is not None:
                    s.stream.wait_event(done[(g + 1) % 2])
                gs[g % 2][0].replay()
                e = torch.cuda.Event(); e.record(s.stream); done[g % 2] = e
            if lockstep and g == 0:
                continue                                  # the suffix runs one group behind the prefix
            gg = g - 1 if lockstep else g
            with torch.cuda.stream(main):
                if lockstep:
                    main.wait_event(done[gg % 2])
                feats_cat = gs[gg % 2][2]
                for k in range(P):
                    ins = [feats_cat[:, k * B:(k + 1) * B], y_in] if slices else sg.inputs
                    m = sg.run(ins, 3000 + k)
                    if consumer:
                        sums.add_(m.double())
                if lockstep:
                    e = torch.cuda.Event(); e.record(main); consumed[gg % 2] = e
    return wall(fn, 8) / P
for name, args_ in (("(d) again", (0, 0, 0)), ("+ metric accumulation", (1, 0, 0)),
                    ("+ inputs = slices of the prefix output", (1, 1, 0)), ("+ lockstep events", (1, 1, 1))):
    print("%-42s %7.1f us / step" % (name, variant(*args_)))
