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
