"""Start-up of a short run of the look-ahead loop (the driver's `bench.py --steps 20`): where the first super-batch (13
batches) runs and what the host does meanwhile.  No profiler: HIP events on the replaying streams.
  1. whole 20-step runs under the four combinations of SLU_RAMP_WHOLE_CHIP (first super-batch on the unmasked stream) and
     SLU_HOST_WAIT (the host waits for it before enqueuing the steps), with device-side marks around the first replays
  2. the captured first-super-batch graphs replayed alone on their streams
  3. the same with dependent work queued behind them on the other streams
  4. the same after an idle gap (clock ramp)
usage: python tools/diag_whole_chip.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

n = 20
dev = torch.device("cuda:0")
config, model, trainer, train_ds, work = bench.setup("no_unfreezing", 0, 64, 48000, 32)
batches = [tuple(t.to(dev) for t in b) for b in train_ds.loader]
model.train()
COMBOS = (("0", "0", "round 5: look-ahead partition, steps queued at once"), ("0", "1", "look-ahead partition, host waits"),
          ("1", "0", "whole chip, steps queued at once"), ("1", "1", "DEFAULT: whole chip, host waits"))
for w, h, _ in COMBOS:
    os.environ["SLU_RAMP_WHOLE_CHIP"], os.environ["SLU_HOST_WAIT"] = w, ("all" if h == "1" else "0")
    for _ in range(4):
        bench.run_steps(model, trainer, batches, n)
        torch.cuda.synchronize()

_replay = torch.cuda.CUDAGraph.replay
marks = []
def replay(self):
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); _replay(self); b.record(st)
    marks.append((a, b, st.cuda_stream))
torch.cuda.CUDAGraph.replay = replay
print("1. 20-step runs (ms after the run was entered; s0 = first super-batch, s1 = second, s2 = optimisation steps)")
for w, h, label in COMBOS:
    os.environ["SLU_RAMP_WHOLE_CHIP"], os.environ["SLU_HOST_WAIT"] = w, ("all" if h == "1" else "0")
    for rep in range(3):
        del marks[:]
        ev0, ev2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        bench.run_steps(model, trainer, batches, n)
        ev2.record()
        torch.cuda.synchronize()
        names = {}
        print("   %-52s all 20 steps %.3f ms | %s" % (label, ev0.elapsed_time(ev2), "  ".join(
            "s%d %.3f-%.3f" % (names.setdefault(sid, len(names)), ev0.elapsed_time(a), ev0.elapsed_time(b)) for a, b, sid in marks[:4])))
torch.cuda.CUDAGraph.replay = _replay
os.environ.pop("SLU_RAMP_WHOLE_CHIP", None); os.environ.pop("SLU_HOST_WAIT", None)

from slu_hip import ops as _ops
import training as _training
A = _training._ramp_plan(n, 20, 2)[0][0]                 # batches of the first super-batch
sl = trainer._slots[0]
firsts = sorted(((k, v) for k, v in sl.graphs.items() if v is not None and k[0] == A), key=lambda kv: kv[0][-1])
xs = [batches[i % len(batches)][0] for i in range(A)]
_ops.store_u64(sl.words, [t.data_ptr() for t in xs] + [0] * (sl.MAX_TABLE - A) + [16])
torch.cuda.synchronize()
sg = next(iter(trainer._step_graphs.values()))
main = trainer._train_stream
other = trainer._slots[1]
g2 = [v for k, v in other.graphs.items() if v is not None and k[0] == n - A][0][0]

def timed(graph, st, pend="nothing", gap=0.0, reps=5):
    ts = []
    for rep in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        while time.perf_counter() - t < gap:
            pass
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st); graph.replay(); e1.record(st)
        if pend in ("the second super-batch", "both"):
            with torch.cuda.stream(other.stream):
                other.stream.wait_event(e1)
                g2.replay()
        if pend in ("the group's optimisation steps", "both"):
            with torch.cuda.stream(main):
                main.wait_event(e1)
                for i in range(A):
                    sg.run(sg.inputs, 200000 + i)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return " ".join("%.3f" % t for t in ts)

for key, (graph, x, f) in firsts:
    st = sl.whole if key[-1] else sl.stream
    where = "whole chip (256 CUs)" if key[-1] else "look-ahead partition (160 CUs)"
    print("2. first super-batch (%d batches) on the %s, replayed alone: %s ms" % (A, where, timed(graph, st)))
    for pend in ("the second super-batch", "the group's optimisation steps", "both"):
        print("3.    with %s queued behind it on the other stream(s): %s ms" % (pend, timed(graph, st, pend)))
    for gap in (0.0003, 0.001, 0.005, 0.05):
        print("4.    after %.1f ms of idle device: %s ms" % (1e3 * gap, timed(graph, st, gap=gap, reps=4)))
