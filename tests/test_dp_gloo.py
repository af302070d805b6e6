"""CPU, world_size 2, gloo: the data-parallel pieces (flat gradient bucket, metric reduction,
Trainer step loop) give every rank the parameters a single process gets on the un-sharded batch."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "end-to-end-slu_amd")


class TinySLU(torch.nn.Module):
    """Duck-types what Trainer needs from models.Model (loss, acc) without any GPU kernel."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.enc = torch.nn.Linear(20, 16)
        self.f64 = torch.nn.Parameter(torch.ones(3, dtype=torch.float64))     # like the Sinc params
        self.unused = torch.nn.Linear(4, 4)                                   # never receives a gradient
        self.head = torch.nn.Linear(16, 5)
        self.seq2seq = False
        self.unfreeze_calls = 0

    def print_frozen(self):
        pass

    def unfreeze_one_layer(self):
        self.unfreeze_calls += 1
        for p in self.enc.parameters():
            p.requires_grad = True

    def forward(self, x, y):
        h = torch.tanh(self.enc(x)) * self.f64.sum().float()
        logits = self.head(h)
        loss = torch.nn.functional.cross_entropy(logits, y[:, 0])
        acc = (logits.max(1)[1] == y[:, 0]).float().mean()
        return loss, acc


class Cfg:
    training_lr = 0.01
    pretraining_lr = 0.01
    pretraining_type = 2


class OneShotDataset:
    def __init__(self, batches):
        self.loader = batches


def _batches(n, bs):
    g = torch.Generator().manual_seed(5)
    return [(torch.randn(bs, 20, generator=g), torch.randint(0, 5, (bs, 1), generator=g)) for _ in range(n)]


def _worker(rank, world, port, tmp, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from slu_hip import dp
    import training
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cfg = Cfg()
    cfg.folder = tmp
    model = TinySLU()
    for p in model.enc.parameters():
        p.requires_grad = False                                   # frozen at first, unfrozen after epoch 1
    trainer = training.Trainer(model, cfg)
    full = _batches(3, 8)
    shard = [(x[rank::world], y[rank::world]) for x, y in full]
    res = []
    for _ in range(2):
        res.append(trainer.train(OneShotDataset(shard), print_interval=1000))
    res.append(trainer.test(OneShotDataset(shard)))
    payload = trainer.bucket.nbytes()
    torch.save({"sd": model.state_dict(), "res": res, "payload": payload,
                "log_exists": os.path.isfile(os.path.join(tmp, "training", "log.csv"))},
               os.path.join(out, "rank%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_trainer_dp2_equals_single_process(tmp_path):
    sys.path.insert(0, PKG)
    import training
    os.makedirs(tmp_path / "training")
    os.makedirs(tmp_path / "single" / "training")
    port = 29000 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # single process on the full batches
    cfg = Cfg()
    cfg.folder = str(tmp_path / "single")
    model = TinySLU()
    for p in model.enc.parameters():
        p.requires_grad = False
    trainer = training.Trainer(model, cfg)
    full = _batches(3, 8)
    res = [trainer.train(OneShotDataset(full), print_interval=1000) for _ in range(2)]
    res.append(trainer.test(OneShotDataset(full)))
    for k, v in model.state_dict().items():
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k                  # ranks stay in lock-step
        assert torch.allclose(r0["sd"][k], v, rtol=1e-5, atol=1e-6), k   # == un-sharded training
    for a, b, c in zip(r0["res"], r1["res"], res):
        assert a == b                                                     # reduced metrics identical
        assert all(abs(x - y) < 1e-5 for x, y in zip(a, c))
    assert r0["log_exists"]
    # second-epoch bucket: enc (20*16+16) + head (16*5+5) fp32 + 3 fp64; `unused` never joins
    assert r0["payload"] == (20 * 16 + 16 + 16 * 5 + 5) * 4 + 3 * 8
    log = open(tmp_path / "training" / "log.csv").read().splitlines()
    assert log[0] == ",intent_loss,intent_acc,set" and len(log) == 4 and log[-1].endswith("valid")


def _bucket_worker(rank, world, port, out):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from slu_hip import dp
    dp.init_from_env(backend="gloo")
    a = torch.nn.Parameter(torch.zeros(4))
    b = torch.nn.Parameter(torch.zeros(2, 3, dtype=torch.float64))
    c = torch.nn.Parameter(torch.zeros(5))
    bucket = dp.GradBucket([a, b, c])
    a.grad = torch.full((4,), float(rank + 1))
    b.grad = torch.full((2, 3), float(10 * (rank + 1)), dtype=torch.float64)
    bucket.allreduce_mean()
    assert c.grad is None and bucket.active
    ok = torch.allclose(a.grad, torch.full((4,), 1.5)) and torch.allclose(b.grad, torch.full((2, 3), 15.0, dtype=torch.float64))
    # after the collective the gradients ARE slices of the flat buckets (no copy back)
    ok = ok and a.grad.data_ptr() == bucket.flats[torch.float32].data_ptr() and bucket.nbytes() == 4 * 4 + 6 * 8
    bucket.release_grads()
    ok = ok and a.grad is None and b.grad is None
    sums = dp.allreduce_sums([1.0 + rank, 2.0], torch.device("cpu"))
    ok = ok and sums == [3.0, 4.0]
    bucket.reset()
    ok = ok and a.grad is None and not bucket.active
    # a step of this process contains a collective; over gloo (host-staged) it is never a node of the step's hipGraph
    ok = ok and dp.data_parallel() and not bucket.collective_in_graph(torch.device("cpu"))
    # bench.py's measurement aid: with the collective stubbed out the gradients stay the local ones
    a.grad = torch.full((4,), float(rank + 1))
    b.grad = None
    bucket.stub = True
    bucket.allreduce_mean()
    ok = ok and torch.equal(a.grad, torch.full((4,), float(rank + 1)))
    bucket.stub = False
    open(os.path.join(out, "ok%d" % rank), "w").write(str(ok))
    torch.distributed.destroy_process_group()


def test_grad_bucket_and_metric_reduction(tmp_path):
    port = 30000 + os.getpid() % 1000
    mp.spawn(_bucket_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok0").read() == "True" and open(tmp_path / "ok1").read() == "True"


# ---- forced failures of the data plane's set-up: every rank must leave on the SAME path (round-5 advisor findings) ----------
class _FakeLib:
    """Stands in for libslu_hip.so's slu_comm_* entry points on a CPU box: each call can be made to fail on chosen ranks."""

    def __init__(self, rank, fail):
        self.rank, self.fail, self.destroyed = rank, fail, 0

    def _rc(self, name):
        return -3 if self.rank in self.fail.get(name, ()) else 0

    def slu_comm_unique_id(self, buf):
        return self._rc("unique_id")

    def slu_comm_init(self, handle_ref, buf, world, rank):
        return self._rc("init")

    def slu_comm_destroy(self, handle):
        self.destroyed += 1
        return 0

    def slu_last_error(self):
        return b"forced failure"


class _FakePlane:
    """A data plane for _race: all-reduce over gloo on host tensors; can be told to fail while it is being timed."""
    kind = "ipc"

    def __init__(self, fail_timing=False, delay=None):
        self.fail_timing, self.closed, self.delay, self.calls = fail_timing, False, delay, 0

    def allreduce(self, flat):
        import time
        torch.distributed.all_reduce(flat)
        self.calls += 1
        if self.delay is not None:
            time.sleep(self.delay(flat.numel()))
        if self.fail_timing:
            flat.add_(1.0)            # wrong sums on this rank only: _time_plane's verification raises after the last call

    def allreduce_flats(self, flats):
        for f in flats.values():
            self.allreduce(f)

    def close(self, collective=True):
        if collective:
            torch.distributed.barrier()
        self.closed = True


def _fallback_worker(rank, world, port, out):
    import contextlib
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from slu_hip import dp, lib
    dp.init_from_env(backend="gloo")
    cpu = torch.device("cpu")
    torch.cuda.device = lambda d: contextlib.nullcontext()            # DirectComm selects its device; none here
    torch.cuda.synchronize = lambda *a, **k: None
    log = []

    def direct(fail):
        fake = _FakeLib(rank, fail)
        lib.load = lambda: fake
        try:
            c = dp.DirectComm(rank, world, cpu)
            return "built", fake, c
        except lib.SluHipError as e:
            return "raised: " + str(e)[:60], fake, None

    # 1. rank 0 cannot draw a unique id: BOTH ranks raise (rank 1 used to wait in the broadcast)
    r, fake, _ = direct({"unique_id": (0,)})
    log.append(("unique_id@0", r.startswith("raised")))
    torch.distributed.barrier()
    # 2. slu_comm_init fails on rank 1 only: both raise, rank 0 destroys the communicator it had built
    r, fake, _ = direct({"init": (1,)})
    log.append(("init@1", r.startswith("raised") and fake.destroyed == (1 if rank == 0 else 0)))
    torch.distributed.barrier()
    # 3. nothing fails: both build
    r, fake, c = direct({})
    log.append(("clean", r == "built" and c is not None))
    c._handle = None                                                  # (nothing to destroy behind the fake)
    torch.distributed.barrier()

    # 4. the start-up race: timing fails on rank 1 only -> both ranks keep the IPC plane, the RCCL one is closed
    class FakeDirect(_FakePlane):
        kind = "rccl"

        def __init__(self, *a):
            super().__init__(fail_timing=False)

        def close(self):
            self.closed = True
    made = []
    real_direct = dp.DirectComm
    dp.DirectComm = lambda *a: made.append(FakeDirect()) or made[-1]
    ipc = _FakePlane(fail_timing=(rank == 1))
    try:
        kept = dp._race(ipc, rank, world, cpu, payloads=(1000,))
    except Exception as e:                                            # noqa: BLE001
        kept = e
    log.append(("race, timing fails@1", kept is ipc and made[-1].closed and ipc.race_us["reason"].startswith("timing failed")))
    torch.distributed.barrier()
    # 5. the race proper (host tensors over gloo, verified sums): both ranks agree on one plane and close the other
    ipc = _FakePlane()
    kept = dp._race(ipc, rank, world, cpu, payloads=(1000,))
    same = torch.tensor([1.0 if kept is ipc else 0.0])
    both = [torch.zeros(1), torch.zeros(1)]
    torch.distributed.all_gather(both, same)
    log.append(("race agrees", both[0].item() == both[1].item() and (made[-1].closed if kept is ipc else ipc.closed)
                and isinstance(kept.race_us["ipc"], float)))
    # 6. RCCL unavailable on every rank (DirectComm raises everywhere): the proven plane stays
    def no_rccl(*a):
        raise lib.SluHipError("no RCCL here")
    dp.DirectComm = no_rccl
    ipc = _FakePlane()
    log.append(("race, no rccl", dp._race(ipc, rank, world, cpu) is ipc and ipc.race_us["reason"] == "rccl unavailable"))
    # 6b. each plane faster for one payload: BOTH stay, a RacedComm picks per payload (the same choice on every rank)
    class SlowSmall(FakeDirect):
        def __init__(self, *a):
            _FakePlane.__init__(self, delay=lambda n: 0.004 if n < 10000 else 0.0)
    dp.DirectComm = lambda *a: made.append(SlowSmall()) or made[-1]
    ipc = _FakePlane(delay=lambda n: 0.0 if n < 10000 else 0.004)
    kept = dp._race(ipc, rank, world, cpu, payloads=(1000, 100000))
    ok = isinstance(kept, dp.RacedComm) and kept._pick(4 * 1000)[0] == "ipc" and kept._pick(4 * 100000)[0] == "rccl" \
        and kept._pick(4 * 50000)[0] == "rccl" and not ipc.closed and not made[-1].closed
    before = (ipc.calls, made[-1].calls)
    kept.allreduce_flats({torch.float32: torch.ones(1000)})
    kept.allreduce_flats({torch.float32: torch.ones(100000)})
    ok = ok and (ipc.calls, made[-1].calls) == (before[0] + 1, before[1] + 1) and kept.kind == "rccl"
    kept.close()
    log.append(("race, mixed winners", ok and ipc.closed and made[-1].closed))
    dp.DirectComm = real_direct
    # 7. a bounded wait expired on rank 1 only: every rank sees the worst status
    log.append(("agreed status", dp.agreed_status(3 if rank == 1 else 0) == 3 and dp.agreed_status(0) == 0))
    open(os.path.join(out, "fb%d" % rank), "w").write(repr(log))
    torch.distributed.destroy_process_group()


def test_data_plane_fallbacks_leave_on_the_same_path(tmp_path):
    """SURVEY 8(e) readiness: each fallback edge of slu_hip/dp.py with a forced failure on ONE rank (world size 2, gloo) —
    DirectComm's unique id and init, the start-up race's timing, RCCL missing, a timed-out wait's status: no rank hangs in
    a control-plane collective its peer never enters, both take the same branch."""
    port = 31000 + os.getpid() % 1000
    mp.spawn(_fallback_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        log = eval(open(tmp_path / ("fb%d" % r)).read())
        assert len(log) == 8 and all(ok for _, ok in log), (r, log)


class _StatusComm:
    """bucket.comm stand-in whose bounded wait 'expired' on rank 1."""
    kind = "ipc"

    def __init__(self, rank):
        self.rank = rank

    def allreduce_flats(self, flats):
        for f in flats.values():
            torch.distributed.all_reduce(f)

    def status(self):
        return 1 if self.rank == 1 else 0          # rank 1 gave up waiting for rank 0


def _poison_worker(rank, world, port, tmp):
    sys.path.insert(0, PKG)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from slu_hip import dp
    import training
    dp.init_from_env(backend="gloo")
    cfg = Cfg()
    cfg.folder = tmp
    trainer = training.Trainer(TinySLU(), cfg)
    trainer.bucket.comm = _StatusComm(rank)
    shard = [(x[rank::world], y[rank::world]) for x, y in _batches(2, 8)]
    try:
        trainer.train(OneShotDataset(shard), print_interval=1000)
        verdict = "no exception"
    except RuntimeError as e:
        verdict = str(e)
    trainer.save_checkpoint()
    open(os.path.join(tmp, "poison%d" % rank), "w").write(verdict)
    torch.distributed.destroy_process_group()


def test_timed_out_wait_stops_every_rank_and_blocks_the_checkpoint(tmp_path):
    """training.Trainer._run: comm.status() != 0 on ONE rank raises on EVERY rank (agreed over the control plane) instead of
    leaving the healthy rank in the epoch-metric all-reduce, and the poisoned parameters are not written over a checkpoint."""
    os.makedirs(tmp_path / "training")
    port = 32000 + os.getpid() % 1000
    mp.spawn(_poison_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        v = open(tmp_path / ("poison%d" % r)).read()
        assert "out of step" in v and "rank 0 inside" in v.replace("for rank 0", "rank 0"), v
    assert not os.path.isfile(tmp_path / "training" / "model_state.pth")
