"""Host-side cost of starting a K-step run of the look-ahead loop (the driver's K = 20): cProfile of one run_steps() after
the graphs are captured, and wall-clock marks: loop entry -> first super-batch enqueued -> first step enqueued -> loop done.
usage: python tools/diag_fill_host.py [steps=20]"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 32)
model = model.to(dev) if not next(model.parameters()).is_cuda else model
batches = [tuple(t.to(dev) for t in b) for b in train_ds.loader]
model.train()
for _ in range(4):
    bench.run_steps(model, trainer, batches, n)
    bench.run_steps(model, trainer, batches, 5)
torch.cuda.synchronize()
for rep in range(3):
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ev0.record()
    bench.run_steps(model, trainer, batches, n, first_done=ev1)
    t1 = time.perf_counter(); ev2.record()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("run %d: host enqueue of the whole run %.3f ms; device: first step done at %.3f ms, all at %.3f ms; wall %.3f ms"
          % (rep, 1e3 * (t1 - t0), ev0.elapsed_time(ev1), ev0.elapsed_time(ev2), 1e3 * (t2 - t0)))
# wall-clock marks inside the start of one run (host side), relative to run_steps() entry
from slu_hip import pipeline as _pl, ops as _ops
marks = []
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        marks.append((label + " >", time.perf_counter()))
        r = f(*a, **k)
        marks.append((label + " <", time.perf_counter()))
        return r
    setattr(obj, name, g)
import training as _tr, models as _mo
wrap(type(trainer), "_sums_buffer", "sums_buffer")
wrap(type(trainer), "lookahead_depth", "lookahead_depth")
wrap(_tr, "_lookahead_width", "width")
wrap(_tr, "_ramp_plan", "ramp_plan")
wrap(_pl.PrefixSlot, "_table_ok", "table_ok")
wrap(_pl, "graphs_enabled", "graphs_enabled")
wrap(_pl.PrefixSlot, "_run", "slot._run")
wrap(_ops, "store_u64", "store_u64")
wrap(torch.cuda.CUDAGraph, "replay", "graph.replay")
wrap(type(model.pretrained_model), "warm_weight_caches", "warm_weight_caches")
for rep in range(2):
    del marks[:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.run_steps(model, trainer, batches, n)
    torch.cuda.synchronize()
    print("marks (us after run_steps entry):", ", ".join("%s %.0f" % (l, 1e6 * (t - t0)) for l, t in marks[:40]))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
bench.run_steps(model, trainer, batches, n)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
