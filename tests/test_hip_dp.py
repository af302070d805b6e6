"""GPU: the data-parallel training step of the HIP model, two processes (gloo; both ranks on the box's one GPU —
RCCL refuses duplicate devices): flat gradient buckets (fp32 + the Sinc layer's float64), one collective per
dtype, the mean's 1/N folded into the Adam kernel.  Two half-batch ranks must end with IDENTICAL parameters,
equal (to summation-order round-off, amplified by Adam's normalisation) to the single-process run on the full batches."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / ("w%d_r%d.pt" % (world, r)))
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SLU_DIST_BACKEND="gloo", SLU_LOCAL_DEVICE="0", SLU_LOOKAHEAD="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_hip_worker.py"), out, str(world)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs.append(out)
    for p in procs:
        log, _ = p.communicate(timeout=300)
        assert p.returncode == 0, log[-3000:]
    return [torch.load(o) for o in outs]


def test_two_rank_hip_training_equals_single_process(tmp_path):
    (one,) = _run(tmp_path, 1)
    a, b = _run(tmp_path, 2)
    assert a["payload"] == one["payload"] > 0 and a["dtypes"] == ["torch.float32", "torch.float64"]
    for k, v in a["sd"].items():
        assert torch.equal(v, b["sd"][k]), k                      # replicas stay bit-identical
    worst = 0.0
    for k, v in one["sd"].items():
        scale = max(v.abs().max().item(), 1e-6)
        worst = max(worst, (v.double() - a["sd"][k].double()).abs().max().item() / scale)
    print("two half-batch ranks vs one full batch: worst relative parameter deviation %.2e" % worst)
    # Adam divides the step by sqrt(v) ~ |g|: where a gradient entry is ~0 the two summation orders can differ in
    # sign, which moves that parameter by up to lr = 3e-3 per step (absolute) — hence a bound of this size, not 1e-6
    assert worst <= 2e-3
    # the per-rank losses are means over the rank's half: their mean is the full-batch loss
    for l1, la, lb in zip(one["losses"], a["losses"], b["losses"]):
        assert abs(0.5 * (la + lb) - l1) <= 1e-5
