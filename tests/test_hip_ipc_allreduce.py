"""GPU: the hand-written all-reduce over peer-mapped device memory (slu_comm_allreduce_ipc, csrc/slu_comm_ipc.hip;
SURVEY 8(e): the xGMI-native collective of the data-parallel step) against the control plane's sum, BIT FOR BIT: the
kernel adds in rank order, one rank per element, and so does the check (tests/ipc_allreduce_worker.py).
On a one-GPU box the ranks share GPU 0 — same-device IPC windows, N = 2 / 4 processes whose kernels spin on each
other's flags side by side — which exercises the whole protocol (typed fp32 + float64 segments, padded tails, epochs
across calls and graph replays, uneven arrival, the bounded waits) except the links themselves; on a box with N GPUs the
same cases run one rank per GPU (they skip themselves otherwise)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, world, shared):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("ipc_w%d_r%d.json" % (world, r)))
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for k in ("SLU_DIST_BACKEND", "SLU_LOCAL_DEVICE", "SLU_DP_SINGLE", "SLU_COMM"):
            env.pop(k, None)
        if shared:
            env["SLU_LOCAL_DEVICE"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ipc_allreduce_worker.py"), out, str(world)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs.append(out)
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(log)
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    return [json.load(open(o)) for o in outs]


def _check(results, world, where):
    assert [r["status"] for r in results] == [0] * len(results), [(r["status"], r.get("diag")) for r in results]   # no wait timed out
    for r in results:
        assert r["oversize_refused"]
        bad = [c["name"] for c in r["cases"] if not c["ok"]]
        assert not bad, (bad, [q.get("diag") for q in results])
        assert len(r["cases"]) == 10
    print("%d ranks %s: 1.21 MB all-reduce %s us per call; first touch of the mapped windows %s ms; longest flag wait %s polls"
          % (world, where, [r["us_per_call_1p21MB"] for r in results], [r["first_touch_ms"] for r in results],
             [r["max_wait_polls"] for r in results]))


@pytest.mark.parametrize("world", [1, 2, 4])
def test_ipc_allreduce_ranks_sharing_one_gpu(tmp_path, world):
    """Up to FOUR ranks on one GPU: their kernels spin on each other's flags and must therefore run side by side.  Eight
    processes oversubscribe the device's hardware queues and the scheduler time-slices them (measured: 22.8 ms per call
    instead of 20 us, profiles/r05_e_ipc_shared_gpu.txt) — an artefact of the shared-GPU test set-up that one rank per GPU
    does not have (the N = 8 case below runs there).  The uneven-arrival cases make real stragglers: the rank whose first
    matmul initialises rocBLAS arrives 1.5 - 5 s late (longest flag wait 0.6 - 2.1 M polls, printed below)."""
    _check(_run(tmp_path, world, shared=True), world, "sharing GPU 0")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_ipc_allreduce_one_rank_per_gpu(tmp_path, world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs (this box has %d)" % (world, torch.cuda.device_count()))
    _check(_run(tmp_path, world, shared=False), world, "over xGMI")
