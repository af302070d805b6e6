"""Worker of tests/test_hip_ipc_allreduce.py: one rank of the hand-written IPC all-reduce (slu_comm_allreduce_ipc),
checked word for word against the sum the control plane computes.
    python ipc_allreduce_worker.py <out.json> <world_size>
Every case: each rank draws its own fp32 (+ float64) bucket, the inputs are gathered over gloo (host tensors) and added
IN RANK ORDER on the CPU in the bucket's own precision — the order the kernel uses, so the comparison is bit for bit for
any number of ranks —, then the kernel reduces the device buckets in place."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from slu_hip import dp, lib, pipeline  # noqa: E402

out, world = sys.argv[1], int(sys.argv[2])
rank, ws, local = dp.init_from_env()
assert ws == world and (ws == 1 or dist.get_backend() == "gloo")
torch.cuda.set_device(local)
lib.require_gfx950()
dev = torch.device("cuda", local)
comm = dp.IpcComm(rank, ws, dev)
res = {"cases": [], "backend": dist.get_backend() if ws > 1 else "none", "first_touch_ms": round(comm.first_touch_ms, 3)}


def ordered_sum(t):
    """sum over ranks of the host tensor t, added in rank order 0 .. N-1 in t's dtype (what the kernel computes)."""
    if ws == 1:
        return t.clone()
    parts = [torch.empty_like(t) for _ in range(ws)]
    dist.all_gather(parts, t)
    s = parts[0].clone()
    for q in range(1, ws):
        s += parts[q]
    return s


def case(name, n32, n64, seed, graph_replays=0, straggler=False):
    g = torch.Generator().manual_seed(1000 * seed + rank)
    ok = True
    stream = torch.cuda.Stream(dev)
    f32 = torch.empty(max(n32, 1), dtype=torch.float32, device=dev)[:n32]
    f64 = torch.empty(max(n64, 1), dtype=torch.float64, device=dev)[:n64]
    flats = {}
    if n32:
        flats[torch.float32] = f32
    if n64:
        flats[torch.float64] = f64
    graph = None
    rounds = max(1, graph_replays)
    for it in range(rounds):
        h32 = torch.randn(n32, generator=g) * (1.0 + it)
        h64 = torch.randn(n64, generator=g, dtype=torch.float64) * (1.0 + it)
        want32, want64 = ordered_sum(h32), ordered_sum(h64)
        with torch.cuda.stream(stream):
            f32.copy_(h32)
            f64.copy_(h64)
            if straggler and rank == (it % ws):
                # uneven arrival: this rank is busy for a few milliseconds before it reaches the collective
                junk = torch.randn(4096, 4096, device=dev)
                for _ in range(3):
                    junk = junk @ junk * 1e-3
            if graph_replays:
                if graph is None:
                    stream.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    if ws > 1:
                        dist.barrier()
                    with pipeline.capture(graph, stream):
                        comm.allreduce_flats(flats)
                    # a capture enqueues nothing: the first replay is the first execution on every rank
                graph.replay()
            else:
                comm.allreduce_flats(flats)
        stream.synchronize()
        got32, got64 = f32.cpu(), f64.cpu()
        good = torch.equal(got32, want32) and torch.equal(got64, want64)
        if not good and "diag" not in res:
            bad = (got32 != want32).nonzero().flatten()
            res["diag"] = {"case": name, "round": it, "rank": rank, "bad32": int(bad.numel()), "bad64": int((got64 != want64).sum()),
                           "first": int(bad[0]) if bad.numel() else -1, "last": int(bad[-1]) if bad.numel() else -1,
                           "got": [float(v) for v in got32[bad[:4]]], "want": [float(v) for v in want32[bad[:4]]],
                           "own_input": [float(v) for v in h32[bad[:4]]]}
        ok = ok and good
    res["cases"].append({"name": name, "ok": bool(ok), "rounds": rounds})


case("one element", 1, 0, 1)
case("three elements + one double (padded tails)", 3, 1, 2)
case("fewer units than ranks", 5, 0, 3)
case("float64 only", 0, 160, 4)
case("intent bucket 1.21 MB", 302616, 0, 5)
case("everything trainable: 1 380 000 fp32 + 160 float64", 1380000, 160, 6)
case("odd length, uneven arrival", 70001, 33, 7, straggler=True)
case("captured as a hipGraph node, 12 replays with fresh data", 302616, 160, 8, graph_replays=12)
case("captured, uneven arrival", 9999, 0, 9, graph_replays=6, straggler=True)
# timing of the step's bucket: 50 back-to-back calls between two events
f = torch.zeros(302616, dtype=torch.float32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    comm.allreduce(f)
e0.record()
for _ in range(50):
    comm.allreduce(f)
e1.record()
torch.cuda.synchronize()
res["us_per_call_1p21MB"] = round(1e3 * e0.elapsed_time(e1) / 50, 2)
res["status"] = comm.status()
res["max_wait_polls"] = comm.max_wait_polls()
# a payload beyond the window's staging capacity (8 MB) goes in several launches; the C entry point itself refuses it
case("20 MB bucket + 160 float64: three launches", 5000000, 160, 10)
import ctypes  # noqa: E402
big = torch.zeros((9 << 20) // 4, dtype=torch.float32, device=dev)
rc = comm._L.slu_comm_allreduce_ipc(comm._windows, rank, ws, comm.window_bytes, big.data_ptr(), big.numel(), None, 0,
                                    torch.cuda.current_stream().cuda_stream)
res["oversize_refused"] = rc != 0 and b"capacity" in comm._L.slu_last_error()
comm.close()
json.dump(res, open(out, "w"))
if ws > 1:
    dist.barrier()
    dist.destroy_process_group()
