"""Data parallelism for the training loop: one process per GPU, gradients averaged with ONE RCCL
all-reduce per step over a flat bucket (torch.distributed backend "nccl" is RCCL on ROCm; xGMI
between the 8 GPUs of a node).  The reference has no distributed code at all — this is new design
(SURVEY.md §8e).

Per step the payload is small (1.2 MB with a frozen encoder, 5.5 MB fully unfrozen), i.e. the
collective is latency-bound: a single collective over one contiguous buffer is the shape that
matters, not bandwidth tuning.  Parameter gradients are VIEWS into the bucket (no pack/unpack
copies); the bucket is rebuilt when the set of parameters that receive gradients changes (the
gradual-unfreezing schedule changes it once per epoch).
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialises torch.distributed from torchrun's environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).  No-op for 1 process."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if ws > 1 and not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=ws, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=ws)
    return rank, ws, local


class GradBucket:
    """Flat gradient buckets (one per dtype: the Sinc parameters are float64) whose slices ARE the
    parameters' .grad tensors."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.live = []
        self.flats = {}
        self.signature = None

    def reset(self):
        """Forget the bucket (call after the trainable set changed, e.g. unfreeze_one_layer)."""
        for p in self.live:
            p.grad = None
        self.live, self.flats, self.signature = [], {}, None

    @property
    def active(self):
        return self.signature is not None

    def zero(self):
        for flat in self.flats.values():
            flat.zero_()

    def nbytes(self):
        return sum(f.numel() * f.element_size() for f in self.flats.values())

    def _rebuild(self, live):
        by_dtype = {}
        for p in live:
            by_dtype.setdefault(p.grad.dtype, []).append(p)
        self.flats = {}
        for dtype, ps in by_dtype.items():
            n = sum(p.numel() for p in ps)
            flat = torch.empty(n, dtype=dtype, device=ps[0].device)
            off = 0
            for p in ps:
                view = flat[off:off + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view
                off += p.numel()
            self.flats[dtype] = flat
        self.live = live
        self.signature = tuple(id(p) for p in live)

    def allreduce_mean(self):
        """Average the gradients of all ranks (call after backward)."""
        rank, ws = world()
        live = [p for p in self.params if p.grad is not None]
        sig = tuple(id(p) for p in live)
        if sig != self.signature:
            self._rebuild(live)
        if ws == 1:
            return
        for flat in self.flats.values():
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(ws)


def allreduce_sums(values, device):
    """Sum a short list of Python floats over ranks (epoch metrics); identity for one process."""
    rank, ws = world()
    if ws == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()
