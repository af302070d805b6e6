"""Adam on the HIP kernels: the update rule, defaults and per-parameter step counting of
torch.optim.Adam (what the reference builds at training.py:19), one launch per (dtype, cohort) instead of
torch's capturable multi-tensor kernel.  A cohort = the parameters that received their first gradient at
the same optimisation step (gradual unfreezing adds cohorts); each has its own device-resident step
counter, so a hipGraph-captured optimiser step can be replayed."""
import ctypes

import torch

from . import lib as _lib


class HipAdam(torch.optim.Optimizer):
    MAX_COHORTS = 64

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._steps = None            # int64 device vector: updates done so far, per cohort
        self._tickets = None          # uint32 device vector: workgroup tickets of the launch that advances a counter
        self._n_cohorts = 0
        self._plan = None             # cached launch arguments, keyed by the identity of every (param, grad)
        self._plan_key = None
        self._plan_mask = 0           # bit c set: cohort c has a parameter with a gradient in this plan
        self.grad_div = 1.0           # gradients are divided by it inside the kernel (DP: the world size)

    def _build(self):
        L = _lib.load()
        vmax = L.slu_adam_max_tensors()
        new_cohort = None
        lists = {}
        key = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.grad.layout != torch.strided:
                    raise TypeError("HipAdam needs dense CUDA parameters")
                st = self.state[p]
                if not st:
                    if new_cohort is None:
                        if self._n_cohorts >= self.MAX_COHORTS:
                            raise RuntimeError("HipAdam: more than %d parameter cohorts" % self.MAX_COHORTS)
                        new_cohort = self._n_cohorts
                        self._n_cohorts += 1
                    st["cohort"] = new_cohort
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if p.dtype not in (torch.float32, torch.float64) or p.grad.dtype != p.dtype:
                    raise TypeError("HipAdam supports float32 / float64 parameters")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise TypeError("HipAdam needs contiguous parameters and gradients")
                lists.setdefault((gi, p.dtype, st["cohort"]), []).append((p, p.grad, st["exp_avg"], st["exp_avg_sq"]))
                key.append((p.data_ptr(), p.grad.data_ptr()))
        if self._steps is None and lists:
            dev = next(iter(lists.values()))[0][0].device
            self._steps = torch.zeros(self.MAX_COHORTS, dtype=torch.int64, device=dev)
        plan = []
        for (gi, dtype, cohort), items in lists.items():
            for i in range(0, len(items), vmax):
                part = items[i:i + vmax]
                n = len(part)
                arr = lambda j: (ctypes.c_void_p * n)(*[t[j].data_ptr() for t in part])
                numel = (ctypes.c_int64 * n)(*[t[0].numel() for t in part])
                plan.append([gi, arr(0), arr(1), arr(2), arr(3), numel, n, 4 if dtype == torch.float32 else 8, cohort, 0])
        # the LAST launch that reads a cohort's step counter also advances it (ticket: its last workgroup writes
        # the counter after every workgroup has read it) — no separate increment launch per step
        seen = set()
        for entry in reversed(plan):
            if entry[8] not in seen:
                seen.add(entry[8])
                entry[9] = 1
        self._plan_mask = 0
        for (_, _, cohort) in lists:
            self._plan_mask |= 1 << cohort
        if self._tickets is None and lists:
            self._tickets = torch.zeros(self.MAX_COHORTS, dtype=torch.int32, device=self._steps.device)
        return plan, tuple(key)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        L = _lib.load()
        # the launch arguments are rebuilt only when a parameter / gradient address changed
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for g in self.param_groups for p in g["params"]
                    if p.grad is not None)
        if key != self._plan_key:
            self._plan, self._plan_key = self._build()
        if not self._plan:
            return loss
        stream = torch.cuda.current_stream().cuda_stream
        base, tickets = self._steps.data_ptr(), self._tickets.data_ptr()
        # only the cohorts updated in this step advance (torch.optim.Adam counts steps per parameter and
        # skips parameters without a gradient)
        for gi, p, g, m, v, numel, n, eb, cohort, advance in self._plan:
            grp = self.param_groups[gi]
            b1, b2 = grp["betas"]
            _lib.check(L.slu_adam_multi(p, g, m, v, numel, n, eb, base + 8 * cohort, float(grp["lr"]), float(b1),
                                        float(b2), float(grp["eps"]), float(self.grad_div),
                                        tickets + 4 * cohort if advance else None, stream),
                       "slu_adam_multi")
        return loss

    # -- checkpointing: the per-parameter `step` of torch.optim.Adam's state_dict ----------------------
    def state_dict(self):
        if self._steps is not None:
            steps = self._steps.tolist()
            for st in self.state.values():
                if "cohort" in st:
                    st["step"] = torch.tensor(float(steps[st["cohort"]]))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """Accepts a state_dict of this class or of torch.optim.Adam (exp_avg, exp_avg_sq, step): cohorts
        are rebuilt from the distinct step counts."""
        super().load_state_dict(state_dict)
        values = sorted({int(st["step"]) for st in self.state.values() if "step" in st})
        if len(values) > self.MAX_COHORTS:
            raise RuntimeError("HipAdam: more than %d distinct step counts" % self.MAX_COHORTS)
        dev = None
        for p, st in self.state.items():
            if "step" in st:
                st["cohort"] = values.index(int(st["step"]))
                dev = p.device
        self._n_cohorts = len(values)
        if dev is not None:
            self._steps = torch.zeros(self.MAX_COHORTS, dtype=torch.int64, device=dev)
            if values:
                self._steps[:len(values)] = torch.tensor(values, dtype=torch.int64)
        self._plan = self._plan_key = None
