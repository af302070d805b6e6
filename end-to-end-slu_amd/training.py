"""`training.Trainer` with the reference's interface (reference training.py:9-171) on top of the
MI355X-native model, plus what the reference lacks: data-parallel training over the GPUs of a node
(one RCCL all-reduce of a flat gradient bucket per step) and a step loop without a host
synchronisation per batch.

Kept from the reference: constructor signature and attributes (`lr`, `checkpoint_path`,
`optimizer` = Adam over ALL model parameters, `epoch`, `df`), `train(dataset, print_interval)` /
`test(dataset)` return tuples and their order, the per-epoch `unfreeze_one_layer()` call, the
`log.csv` format (test rows are labelled "valid" as well), `model_state.pth` checkpoints, and the
printed messages.  Deliberate differences: metrics are accumulated on the device (float64) and read
back once per print interval / epoch instead of four `.cpu()` round trips per step
(training.py:99-100 — the values logged are the same); `test()` stays on the GPU (the reference
moves the model to the CPU for its seq2seq beam search, training.py:150,166, and crashes on a
CPU-only host); under torch.distributed only rank 0 prints, logs and writes checkpoints.
"""
import collections
import contextlib
import math
import os
import warnings

import pandas as pd
import torch
from tqdm import tqdm

from data import SLUDataset, ASRDataset
from models import PretrainedModel, Model
from slu_hip import dp


def models_masks_injected():
    import models
    return models._DropoutState.masks is not None


def _lookahead_env():
    """SLU_LOOKAHEAD: number of batches per look-ahead super-batch; unset / "auto" -> -1 (by batch size)."""
    v = os.environ.get("SLU_LOOKAHEAD", "auto")
    return -1 if v == "auto" else int(v)


def _max_step_graphs():
    """Distinct step shapes kept as captured hipGraphs (each owns its activations: ~0.2 GB at B = 64)."""
    return int(os.environ.get("SLU_MAX_STEP_GRAPHS", "8"))


def _param_signature(model):
    return tuple(True if p.requires_grad else p._version for p in model.parameters())


def _lookahead_width(depth, batch_size):
    """Batches per look-ahead super-batch: explicit, or eight sequences per CU the side streams may use — one
    16-sequence workgroup of the split-precision recurrence per direction and CU: 1280 sequences = 20 batches of 64 on
    the 160 CUs of the default partition (slu_hip/pipeline.cu_split has the sweeps).  One batch more and the recurrences
    need a second round of workgroups (round 2, 96 + 160 CUs: 18 / 19 / 20 / 21 / 22 batches -> 272 / 265 / 272-282 / 232 /
    238 k utt/s).  Wider super-batches make the frozen stages more efficient, but the two partitions share the power
    budget (with all 256 CUs busy the clock drops to ~1.8 GHz and the latency-bound trainable step slows by 20-30 %), and
    a wider first super-batch is a longer pipeline fill."""
    if depth > 0:
        return depth
    from slu_hip import pipeline
    cus = 256
    if torch.cuda.is_available():
        cus = pipeline.n_compute_units(torch.cuda.current_device()) - pipeline.cu_split()
    return max(2, min(32, (8 * cus) // max(1, batch_size)))


def _ramp_plan(n_run, width, n_slots, latency_steps=5.0, slope=0.4):
    """Sizes of the FIRST super-batches of a run of n_run batches (the rest are `width` batches each), and how many of
    them are started side by side instead of one behind the other.  Default: ONE capped first super-batch — a run that fits
    in two super-batches (n_run < 2 width) is split a : n_run - a so that the second super-batch's encoder, which runs
    beside the first one's steps, is ready when they end.  A frozen prefix of k batches costs L + r k on the look-ahead
    partition (L = the latency of ~560 dependent recurrence steps, r = the throughput term), a trainable step costs s; the
    first super-batch's steps (a s) hide the second's encoder (L + r (n - a)) when
        a >= L / (s + r) + n r / (s + r).
    With L = 1.0 - 1.4 ms, r = 0.08 - 0.12 ms, s = 0.15 - 0.17 ms (f16x2 ... exact fp32 frozen stages) the
    two terms are 4.3 - 4.8 batches and 0.35 - 0.41 n: ONE rule for every arithmetic, a = ceil(5 + 0.4 n), capped by the
    width and the run — no constant fitted to a mode.  One batch too few stalls the steps for s + r = 0.27 ms, one too many
    costs 0.085 ms.  (Rounds 4 - 6 used a share of n — 0.6 / 0.7 per arithmetic, then 2/3 — which is this line at n = 20 only.
    Measured, bf16x3, first super-batch on the whole chip, k utt/s, 3 runs each, profiles/r06_y_ramp_sweep.txt:
    n = 12: 8+4 175, 9+3 187, 10+2 190;  n = 20: 12+8 221, 13+7 222, 14+6 220;  n = 30: 16+14 242, 17+13 246, 20+10 238.)
    A run shorter than the rule's a is ONE super-batch.
    SLU_RAMP=a,b,c: explicit sizes, one behind the other (SLU_RAMP_SIDE=1 with as many look-ahead slots: side by side — measured
    slower, profiles/r05_a_sweep.txt: CU-masked streams have no priorities, super-batches side by side share the partition and
    ALL finish late)."""
    env = os.environ.get("SLU_RAMP", "0")
    if env not in ("auto", "0"):
        sizes = [max(1, int(v)) for v in env.split(",") if v.strip()]
        k = int(os.environ.get("SLU_RAMP_SIDE", "0"))           # 1: all of them side by side; k >= 2: the first k
        side = min(len(sizes), n_slots) if k == 1 and n_slots >= len(sizes) else (min(k, len(sizes), n_slots) if k >= 2 else 0)
        return sizes, side
    T = min(n_run, width)
    if env == "0" or n_slots < 3 or T < 9:
        if n_run >= 2 * width:
            return [], 0
        return [max(1, min(T, int(math.ceil(latency_steps + slope * n_run - 1e-9))))], 0
    a = max(2, int(T / 7.0 + 0.5))
    b = max(a, int(2 * T / 7.0 + 0.5))
    return [a, b, T - a - b], 3


class Trainer:
    def __init__(self, model, config):
        self.model = model
        self.config = config
        if isinstance(self.model, PretrainedModel):
            self.lr = config.pretraining_lr
            self.checkpoint_path = os.path.join(self.config.folder, "pretraining")
        else:
            self.lr = config.training_lr
            self.checkpoint_path = os.path.join(self.config.folder, "training")
        # same optimiser and defaults as the reference (training.py:19); on a GPU the update runs as one
        # launch per dtype on the HIP kernel (slu_hip/optim.py: same rule and per-parameter step counts)
        on_gpu = all(p.is_cuda for p in model.parameters())
        if on_gpu:
            from slu_hip import ops as _ops
            self.wgrad = _ops.resolve_wgrad()      # (mode, workgroup budget) of the weight-gradient launches, fixed for this trainer
            from slu_hip.optim import HipAdam
            self.optimizer = HipAdam(model.parameters(), lr=self.lr)
        else:
            self.optimizer = torch.optim.Adam(model.parameters(), lr=self.lr)
        self.epoch = 0
        self.df = None
        self.rank, self.world_size = dp.world()
        # flat gradient bucket: the unit of the per-step all-reduce under data parallelism, and (on a
        # GPU) what gives the gradients fixed addresses for hipGraph-captured steps
        self.data_parallel = dp.data_parallel()      # world size > 1 (or the one-rank test mode SLU_DP_SINGLE=1)
        self.bucket = dp.GradBucket(model.parameters()) if (self.data_parallel or on_gpu) else None
        self._step_graphs, self._eager_steps = {}, {}
        self.capture_failures = 0
        self._hip_adam = on_gpu
        if on_gpu and self.data_parallel:
            # data-parallel mean: the all-reduce delivers the SUM, Adam divides by the world size in-kernel
            self.optimizer.grad_div = float(self.world_size)
            self.bucket.divide = False
            # the data plane of the gradient collective (slu_hip/dp.make_comm): the hand-written IPC all-reduce after its
            # self-test, else RCCL through the C ABI, else (None) torch.distributed's collective on the buckets
            self.bucket.comm = dp.make_comm(self.rank, self.world_size, next(model.parameters()).device)

    # -- checkpoints / log (reference training.py:23-45) -------------------------------------------
    def load_checkpoint(self):
        path = os.path.join(self.checkpoint_path, "model_state.pth")
        if os.path.isfile(path):
            try:
                dev = next(self.model.parameters()).device
                self.model.load_state_dict(torch.load(path, map_location=dev))
            except Exception:
                print("Could not load previous model; starting from scratch")
        else:
            print("No previous model; starting from scratch")

    def save_checkpoint(self):
        if self.rank != 0:
            return
        if getattr(self, "poisoned", False):
            # a bounded wait of the gradient all-reduce expired during the last epoch (_run raised on every rank): the
            # parameters have absorbed unreduced gradients since — never overwrite a good checkpoint with them
            print("Could not save model (the data-parallel replicas went out of step)")
            return
        try:
            torch.save(self.model.state_dict(), os.path.join(self.checkpoint_path, "model_state.pth"))
        except Exception:
            print("Could not save model")

    def log(self, results):
        if self.rank != 0:
            return
        if self.df is None:
            self.df = pd.DataFrame(columns=[field for field in results])
        self.df.loc[len(self.df)] = results
        self.df.to_csv(os.path.join(self.checkpoint_path, "log.csv"))

    # -- one optimisation step ---------------------------------------------------------------------
    def _step(self, loss):
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if loss.is_cuda:
            from slu_hip import ops
            ops._Fork.join(loss.device)          # branches left open by the backward pass (ops._Fork.defer): one join
        if self.bucket is not None:
            self.bucket.allreduce_mean()
        self.optimizer.step()

    def _is_asr(self, dataset):
        return isinstance(dataset, ASRDataset)

    def _say(self, text):
        if self.rank == 0:
            print(text)

    def _epoch_means(self, sums, num_examples, device):
        comm = self.bucket.comm if self.bucket is not None else None
        tot = dp.allreduce_sums([float(s) for s in sums] + [float(num_examples)], device, comm)
        return [v / tot[-1] for v in tot[:-1]]

    def close(self):
        """Release what the trainer holds outside torch's allocator: the direct RCCL communicator (SLU_COMM=rccl).
        Called by main.py / bench.py at the end of a run; harmless to call twice."""
        if self.bucket is not None and self.bucket.comm is not None:
            self.bucket.comm.close()
            self.bucket.comm = None

    def _forward_losses(self, batch, asr, rng_step=None):
        """-> ([metric tensors in log order], loss to back-propagate).  rng_step: None = the model draws the
        next dropout-stream index itself (the plain reference-style call), else the index to use."""
        if asr:
            x, y_phoneme, y_word = batch
            if rng_step is None:
                phoneme_loss, word_loss, phoneme_acc, word_acc = self.model(x, y_phoneme, y_word)
            else:
                phoneme_loss, word_loss, phoneme_acc, word_acc = self.model(x, y_phoneme, y_word, rng_step=rng_step)
            ptype = self.config.pretraining_type
            loss = {1: phoneme_loss, 3: word_loss}.get(ptype)
            if ptype == 2:
                loss = phoneme_loss + word_loss
            return [phoneme_loss, word_loss, phoneme_acc, word_acc], loss
        x, y_intent = batch
        # through model.__call__ (forward hooks / wrappers keep working); rng_step is a keyword-only extra of this
        # package's Model.forward
        if rng_step is None:
            intent_loss, intent_acc = self.model(x, y_intent)
        else:
            intent_loss, intent_acc = self.model(x, y_intent, rng_step=rng_step)
        return [intent_loss, intent_acc], intent_loss

    def _parameters(self):
        """The model's parameter objects (the set is fixed; device / requires_grad / version are read live): walked once
        instead of at the start of every run."""
        if getattr(self, "_plist", None) is None or self._plist[0] is not self.model:
            self._plist = (self.model, list(self.model.parameters()))
        return self._plist[1]

    def lookahead_depth(self, train, asr):
        """How many batches ahead the FROZEN prefix of the encoder is evaluated on side HIP streams
        (0 = plain sequential steps).  Only SLU training with a frozen prefix qualifies: a frozen
        stage's output does not depend on earlier optimisation steps.  At 64 utterances per step a
        recurrence occupies a fraction of the 256 CUs, so several batches' encoders run as one
        super-batch; the per-batch dropout streams are step-indexed, so the result is the sequential
        one.  depth -1 = automatic: see _lookahead_width (one 16-sequence recurrence workgroup per
        look-ahead CU and direction)."""
        depth = _lookahead_env()
        if not train or asr or depth in (0, 1) or not hasattr(self.model, "prefix_features"):
            return 0, 0
        if not all(p.is_cuda for p in self._parameters()) or models_masks_injected():
            return 0, 0
        n = self.model.frozen_prefix_len()
        # a CNN-block dropout inside the frozen prefix has no per-sub-batch stream (all shipped cfgs have
        # cnn_drop = 0): stay sequential there, so that the step-indexed masks remain those of the plain loop
        pm = self.model.pretrained_model
        if any(st.drop > 0.0 for st in pm._cnn_stages[:n]):
            return 0, 0
        return (depth, n) if n > 0 else (0, 0)

    # -- hipGraph-captured optimisation steps -------------------------------------------------------------
    def _graphable(self):
        """Captured steps need the HIP optimiser (device-resident step counters), fixed gradient addresses
        (the bucket) and the in-kernel Philox dropout (injected masks change per call)."""
        from slu_hip import pipeline
        return (pipeline.graphs_enabled() and self.bucket is not None and self._hip_adam
                and not models_masks_injected() and all(p.is_cuda for p in self.model.parameters()))

    def graph_stats(self):
        """{"step_graphs": captured optimisation steps, "prefix_graphs": captured look-ahead super-batch
        shapes, "capture_failures": shapes that fell back to eager launches} — reported by bench.py."""
        slots = getattr(self, "_slots", None) or []
        out = {"step_graphs": len(self._step_graphs),
               "prefix_graphs": sum(1 for sl in slots for g in sl.graphs.values() if g is not None),
               "capture_failures": self.capture_failures + sum(sl.capture_failures for sl in slots)}
        if self.data_parallel:
            # where the gradient all-reduce of a captured step sits (slu_hip/pipeline.StepGraph)
            sgs = list(self._step_graphs.values())
            out["collective"] = ("none captured yet" if not sgs else
                                 "a node of the step's hipGraph" if all(g.collective_in_graph for g in sgs) else
                                 "eager call between two hipGraphs")
        return out

    def _graph_step(self, key, inputs, step, forward, stream, forks=False, guard=None):
        """One optimisation step on `inputs` (device tensors): replay of the hipGraph captured for `key`
        (captured after three eager steps of that key), else eagerly.  -> metrics (tensor or list).
        guard: the step evaluates FROZEN stages on guarded f16x2 (pipeline.StepGraph): a violation repeats the step
        eagerly — where the model's own guard re-runs the frozen stages on bf16x3 — before the optimiser has run."""
        from slu_hip import pipeline
        if guard is not None:
            key = key + ("guarded",)
        sg = self._step_graphs.get(key)
        if sg is not None:
            self._step_graphs[key] = self._step_graphs.pop(key)          # most recently used last
        if sg is not None and sg.signature != self.bucket.signature:
            sg = None                                   # trainable set changed since capture
        if sg is None and self._eager_steps.get(key, 0) >= 3 and self.bucket.active:
            if key not in self._step_graphs and len(self._step_graphs) >= _max_step_graphs():
                self._step_graphs.pop(next(iter(self._step_graphs)))     # evict the LEAST RECENTLY USED capture
            try:
                sg = pipeline.StepGraph(self, inputs, forward, stream, forks, guard)
                self._step_graphs[key] = sg
            except RuntimeError as e:                   # keep training eagerly if capture fails
                print("hipGraph capture of the training step failed (%s); staying eager" % (e,))
                self._eager_steps[key] = -(1 << 30)
                self._step_graphs.pop(key, None)
                self.capture_failures += 1
        if sg is not None:
            self._last_key = key
            try:
                return sg.run(inputs, step)
            except pipeline.RangeTrip as trip:
                pm = getattr(self.model, "pretrained_model", self.model)
                if trip.overflow:
                    pm.pin_bf16x3("a split-precision stage of a captured step saw |value| = %.3g (limit 65504)"
                                  % max(trip.seen))
                self._step_graphs.pop(key, None)
                self._eager_steps[key] = 0
                self.bucket.release_grads()
                metrics, loss = forward(inputs, step)          # eager: run_stages' own guard / pin decides the arithmetic
                self._step(loss)
                return metrics
        # only runs of equally-shaped steps are worth a capture (~50 ms, a private activation pool): with ragged
        # batches (real-data loaders without length bucketing) the count restarts at every shape change
        if getattr(self, "_last_key", None) != key and self._eager_steps.get(key, 0) >= 0:
            self._eager_steps[key] = 0
        self._last_key = key
        self._eager_steps[key] = self._eager_steps.get(key, 0) + 1
        if len(self._eager_steps) > 256:                # ragged real-data shapes: keep the table bounded
            keep = {k: v for k, v in self._eager_steps.items() if v < 0 or k == key or k in self._step_graphs}
            self._eager_steps = keep
        metrics, loss = forward(inputs, step)
        self._step(loss)
        return metrics

    def _slu_forward(self, n_prefix, sums=None):
        """sums: float64 device tensor the intent head's own launch adds B * (loss, acc) to (the epoch statistics
        of reference training.py:100-104), so that the loop needs no accumulation kernel per step."""
        from slu_hip import ops

        def forward(ins, rng):
            ops.IntentHeadFn.epoch_sums = sums
            try:
                loss, _ = self.model(ins[0], ins[1], rng_step=rng, n_prefix=n_prefix)
            finally:
                ops.IntentHeadFn.epoch_sums = None
            return self.model.last_loss_acc, loss                # (2,) float32 [loss, acc]
        return forward

    def _fused_sums(self):
        """Can the epoch statistics be accumulated inside the step's own kernels?  (the intent head's launch does it;
        the seq2seq decoder's loss is accumulated by the loop instead)"""
        return hasattr(self.model, "pretrained_model") and not getattr(self.model, "seq2seq", False)

    def _sums_buffer(self):
        """Persistent float64 (4) device tensor: the running B-weighted sums of the step metrics of one epoch
        (a fixed address, so that captured steps can accumulate into it)."""
        dev = next(self.model.parameters()).device
        if getattr(self, "epoch_sums", None) is None or self.epoch_sums.device != dev:
            self.epoch_sums = torch.zeros(4, dtype=torch.float64, device=dev)
        return self.epoch_sums

    @staticmethod
    def _accumulate(sums, vals, batch_size):
        if torch.is_tensor(vals):
            sums[:vals.numel()].add_(vals, alpha=batch_size)
        else:
            step_vals = torch.stack([v.detach().to(sums.device).double().reshape(()) for v in vals])
            sums[:len(vals)].add_(step_vals, alpha=batch_size)

    def _asr_forward(self, ins, rng):
        vals, loss = self._forward_losses(ins, True, rng)
        dev = ins[0].device
        # pretraining_type 1 returns host zeros for the word head (reference models.py:317-319)
        vals = torch.stack([v.detach().float().reshape(()) if v.is_cuda else torch.zeros((), device=dev)
                            for v in vals])
        return vals, loss

    def _iterate_full_steps(self, loader, asr, sums=None):
        """Training with nothing to look ahead to (ASR pre-training, SLU with an unfrozen first layer or
        SLU_LOOKAHEAD=0): each step is ~100 short launches, so fixed-shape steps are captured as hipGraphs
        (StepGraph) on a dedicated stream; the losses and parameters are those of the eager loop."""
        from models import next_rng_step
        dev = next(self.model.parameters()).device
        outer = torch.cuda.current_stream()
        if getattr(self, "_full_stream", None) is None:
            self._full_stream = torch.cuda.Stream(dev)
        main = self._full_stream
        main.wait_stream(outer)
        fused = sums is not None and self._fused_sums()
        if hasattr(self.model, "pretrained_model"):
            pm, forward = self.model.pretrained_model, self._slu_forward(0, sums if fused else None)
            forward_plain = self._slu_forward(0, None)
        else:
            pm, forward = self.model, self._asr_forward
            forward_plain = forward
        # a captured step is specific to the set of trainable parameters (gradual unfreezing changes it) and to
        # the frozen weights' contents (their packed bf16 planes are baked into the graph)
        trainable = _param_signature(self.model)
        # frozen stages inside a captured step run on guarded f16x2 like everywhere else (same arithmetic as the look-ahead
        # loop, so the two stay bit-identical): the step is then two graphs with the range check between them, and the
        # epoch statistics are accumulated after the check instead of inside the head's launch
        def step_guard():
            if pm is None or not any(not any(q.requires_grad for q in st.parameters()) for st in pm._stages()):
                return None
            return pm.range_guard() if pm.f16x2_allowed() else None
        from slu_hip import ops as _ops
        was_defer = _ops._Fork.defer
        # nothing runs beside these steps: the weight-gradient launches of long GRU layers go to a graph branch
        # (ops.GRULayerFn.backward, ops.wgrad_branch); the flag is raised per step, never across a yield
        defer = os.environ.get("SLU_GRAPH_FORKS", "1") != "0"
        try:
            with torch.cuda.stream(main):
                pm.warm_weight_caches()
                for batch in loader:
                    ins = [t.to(dev, non_blocking=True) for t in batch]
                    if ins[0].dtype != torch.int16:           # PCM16 batches stay int16 (the model's first block scales them)
                        ins[0] = ins[0].float()
                    guard = step_guard()
                    fused_now = fused and guard is None
                    fwd = forward if (fused_now or not fused) else forward_plain
                    key = ("full", asr, trainable, fused_now) + tuple((tuple(t.shape), t.dtype) for t in ins)
                    _ops._Fork.defer = defer
                    try:
                        vals = self._graph_step(key, ins, next_rng_step(), fwd, main, forks=defer, guard=guard)
                    finally:
                        _ops._Fork.defer = was_defer
                    if sums is not None and not fused_now:
                        self._accumulate(sums, vals, len(batch[0]))
                    yield vals, len(batch[0])
        finally:
            _ops._Fork.defer = was_defer
            outer.wait_stream(main)

    def _iterate(self, loader, train, asr, accumulate=False):
        """Yields ([metric tensors], batch_size) per batch, doing the optimisation step when `train`.
        NOTE for consumers: the metrics of a hipGraph-captured step are ONE static device buffer that the next replay
        overwrites — read (or .clone()) them before advancing the generator, as _run does at print intervals.
        accumulate: also keep the epoch statistics on the device — self.epoch_sums[:n] (float64, zeroed here)
        receives batch_size * metrics of every batch, inside the step's own kernels where they are captured
        (no per-step accumulation launch); the consumer reads it when the generator is exhausted."""
        sums = None
        if accumulate:
            sums = self._sums_buffer()
            sums.zero_()
        depth, n_prefix = self.lookahead_depth(train, asr)
        group_eval = (not train and not asr and hasattr(self.model, "eval_group") and not models_masks_injected()
                      and not getattr(self.model, "seq2seq", False)
                      and all(p.is_cuda for p in self.model.parameters())
                      and _lookahead_env() not in (0, 1))
        if group_eval:
            # evaluation has no step-to-step dependency at all: whole batches are grouped
            group = []

            def flush():
                res = self.model.eval_group([b[0] for b in group], [b[1] for b in group])
                out = [([l, a], len(b[0])) for (l, a), b in zip(res, group)]
                if sums is not None:
                    for vals, bs in out:
                        self._accumulate(sums, vals, bs)
                group.clear()
                return out

            for batch in loader:
                if group and (tuple(batch[0].shape) != tuple(group[0][0].shape) or batch[0].dtype != group[0][0].dtype
                              or len(group) == _lookahead_width(_lookahead_env(), len(group[0][0]))):
                    yield from flush()
                group.append(batch)
            if group:
                yield from flush()
            return
        if depth == 0:
            if train and self._graphable():
                yield from self._iterate_full_steps(loader, asr, sums)
                return
            # the same arithmetic as the captured loop (the weight-gradient branch of long GRU layers and its workgroup
            # budget, ops.wgrad_branch): SLU_GRAPHS=0 and =1 stay bit-identical
            from slu_hip import ops as _ops
            was_defer = _ops._Fork.defer
            on_gpu = train and all(p.is_cuda for p in self.model.parameters())
            try:
                for batch in loader:
                    _ops._Fork.defer = on_gpu and os.environ.get("SLU_GRAPH_FORKS", "1") != "0"
                    try:
                        with torch.set_grad_enabled(train):
                            vals, loss = self._forward_losses(batch, asr)
                            if train:
                                self._step(loss)
                    finally:
                        _ops._Fork.defer = was_defer
                    if sums is not None:
                        self._accumulate(sums, vals, len(batch[0]))
                    yield vals, len(batch[0])
            finally:
                _ops._Fork.defer = was_defer
            return
        # ---- encoder look-ahead pipeline (slu_hip/pipeline.py) --------------------------------------
        import collections
        from models import next_rng_step
        from slu_hip import pipeline
        # The trainable part runs on a dedicated non-default stream: autograd pins each parameter's
        # gradient-accumulation node to the stream of its first use, and hipGraph capture (which cannot
        # happen on the default stream) needs the eager warm-up steps and the capture to agree on it.
        outer = torch.cuda.current_stream()
        dev = next(self.model.parameters()).device
        if getattr(self, "_train_stream", None) is None:
            from slu_hip import pipeline as _pl
            n_cu = _pl.cu_split()
            self._train_stream = (_pl.cu_range_stream(dev, 0, n_cu, priority=-1) if n_cu > 0
                                  else torch.cuda.Stream(dev, priority=-1))     # ahead of the look-ahead streams
        main = self._train_stream
        main.wait_stream(outer)
        if getattr(self, "_slots", None) is None:
            # two slots: one super-batch is consumed while the next is computed (a third slot only reads further ahead:
            # 342 -> 323 k utt/s steady state, profiles/r05_a_sweep.txt)
            n_slots = max(2, int(os.environ.get("SLU_LOOKAHEAD_SLOTS", "2")))
            self._slots = [pipeline.PrefixSlot(dev) for _ in range(n_slots)]   # in-flight super-batches
        pm = self.model.pretrained_model
        with torch.cuda.stream(main):
            pm.warm_weight_caches()
        signature = tuple(p._version for ps in pm.stage_parameters()[:n_prefix] for p in ps)
        for slot in self._slots:
            if slot.signature != signature:          # frozen weights were reloaded: re-capture
                slot.invalidate()
                slot.signature = signature
            slot.stream.wait_stream(main)
        use_graph = pipeline.graphs_enabled()
        # (what the optimisation steps need — step graphs, the forward closure, the parameter signature — is set up AFTER
        # the first super-batches are on their way, below: the device is idle until the first of them is enqueued, and
        # ~0.15 ms of host work in front of it was ~2 % of a 20-step run)
        # the first super-batches of a run are sized and started by _ramp_plan (pipeline fill)
        try:
            n_run = len(loader)
        except TypeError:
            n_run = 1 << 30
        pending = collections.deque()
        it = iter(loader)
        carry = []                                    # a batch read ahead that did not fit its group
        launched = 0
        last_done = [None]
        ramp = [[], 0]                                # [sizes of the first super-batches, how many start side by side]
        # SLU_PREFIX_CHAIN=0 (experiment): super-batches of different slots never wait for each other
        chain = os.environ.get("SLU_PREFIX_CHAIN", "1") != "0"
        # several ranks on ONE GPU (the --share-gpu / SLU_LOCAL_DEVICE test set-up): a whole-chip super-batch of one process
        # would run over the other processes' training partitions (measured: 4 ranks 185 -> 83 k utt/s)
        own_gpu = not dp._shared_device()

        def launch_next():
            """Read up to `depth` equally-shaped batches and start their frozen prefix as one super-batch."""
            nonlocal launched
            group = [carry.pop()] if carry else []
            ver = lambda b: b[0]._version if b[0].is_cuda else None
            versions = [ver(b) for b in group]                 # tensor version of every batch WHEN IT WAS READ
            wcache = {}

            def width():
                bs = len(group[0][0])
                if bs not in wcache:
                    w = _lookahead_width(depth, bs)
                    if launched == 0:
                        ramp[0], ramp[1] = _ramp_plan(n_run, w, len(self._slots))
                    wcache[bs] = min(ramp[0][launched], w) if launched < len(ramp[0]) else w
                return wcache[bs]
            while not group or len(group) < width():
                try:
                    batch = next(it)
                except StopIteration:
                    break
                if group and (tuple(batch[0].shape) != tuple(group[0][0].shape) or batch[0].dtype != group[0][0].dtype):
                    carry.append(batch)          # a super-batch holds ONE shape and ONE sample format (float32 or PCM16)
                    break
                group.append(batch)
                versions.append(ver(batch))
            if not group:
                return False
            slot = self._slots[launched % len(self._slots)]
            launched += 1
            steps = [next_rng_step() for _ in group]                        # consecutive by construction
            # whole_chip: the capped first super-batch of a SHORT run (one that _ramp_plan splits: fewer than two full
            # super-batches, e.g. the driver's 20 steps) is replayed on the whole chip — the training partition has
            # nothing to do until it is through, and in a short run that wait is a large share of the run (14 batches:
            # 2.6 -> 2.3 ms).  Decided by the run's length alone, not by stream.query(): which captured graph a run uses
            # must not be a race.  Long runs keep every super-batch on the look-ahead partition (one key per slot).
            # the first ramp[1] super-batches of the run start side by side; from then on each waits for its predecessor
            # (two full-width super-batches side by side would only delay the one the training stream is waiting for)
            feats, done, guard = slot.run(self.model, [b[0] for b in group], n_prefix, steps[0], use_graph,
                                          after=None if (launched <= ramp[1] or not chain) else last_done[0],
                                          whole_chip=(own_gpu and bool(ramp[0])
                                                      and launched <= (1 if ramp[1] == 0 else int(os.environ.get("SLU_RAMP_WHOLE_N", "0")))))
            last_done[0] = done
            # device-resident batches are read IN PLACE by the (asynchronous) super-batch: remember their tensor
            # versions, so that a loader that recycles its device buffers is caught instead of silently training on
            # whatever the buffer holds by then (batches must stay unchanged until consumed)
            pending.append((group, feats, done, steps, slot, versions, guard))
            return True

        # The consumer's per-step work (metric accumulation in _run) runs with `main` as the current
        # stream: ordered after the step without touching the default stream, whose legacy
        # synchronisation with blocking streams (the CU-masked ones) would serialise the pipeline.
        try:
            with torch.cuda.stream(main):
                launch_next()
                step_graphs = use_graph and self._graphable()
                fused = step_graphs and self._fused_sums()
                forward = self._slu_forward(n_prefix, sums if fused else None)
                trainable = _param_signature(self.model)
                for _ in self._slots[1:]:
                    launch_next()
                host_wait = os.environ.get("SLU_HOST_WAIT", "all")          # "all" | "first" | "0"
                while pending:
                    group, feats_cat, done, steps, slot, versions, guard = pending.popleft()
                    for b, v in zip(group, versions):
                        if v is not None and b[0]._version != v:
                            raise RuntimeError(
                                "a device-resident input batch was modified in place while its look-ahead super-batch was "
                                "still reading it (the loader recycles device buffers): hand over fresh tensors per batch or "
                                "host batches, or set SLU_LOOKAHEAD=0")
                    if host_wait != "0":
                        # The host WAITS for a super-batch before it enqueues that group's steps.  For the run's first one
                        # nothing can run before it anyway, and step graphs queued on the (high-priority) training stream
                        # behind its event slow the running super-batch down — measured, tools/diag_whole_chip.py,
                        # profiles/r06_y_first_super_batch.txt: 13 batches on the whole chip 2.7 - 3.0 ms with the steps
                        # queued, 2.3 ms with the host waiting.  In steady state the super-batch is normally through when the
                        # previous group's steps are (the prefix bounds the loop), so the wait is short; it keeps the host
                        # at most one group ahead and is worth 0.5 % there (363.2 - 365.0 -> 365.8 - 367.2 k utt/s,
                        # profiles/r06_y_host_wait_all.txt).  "first": only the run's first super-batch.  (Queuing the
                        # group's first 1 / 2 / 4 steps BEFORE the wait, to hide the host's wake-up: 227 / 228 / 226 k
                        # against 230 k utt/s for the 20-step command — not kept.)
                        done.synchronize()
                        if host_wait == "first":
                            host_wait = "0"
                    if guard is not None:
                        # f16x2 ran under the slot's range guard: read its words BEFORE the features are used (the
                        # super-batch normally finished while the previous group's steps were running: the wait is short
                        # and the training stream still has that group's tail to run)
                        done.synchronize()
                        overflow, quiet, seen = guard.verdict()
                        if overflow or quiet:
                            if overflow:
                                pm.pin_bf16x3("a split-precision stage of a look-ahead super-batch saw |value| = %.3g "
                                              "(limit 65504)" % max(seen))
                            feats_cat, done, _ = slot.run(self.model, [b[0] for b in group], n_prefix, steps[0],
                                                          use_graph, after=None, guarded=False)
                    B = group[0][0].shape[0]
                    for k, batch in enumerate(group):
                        if k == 0:
                            main.wait_event(done)
                            feats_cat.record_stream(main)
                        feats = feats_cat[:, k * B:(k + 1) * B] if len(group) > 1 else feats_cat
                        y = batch[1].to(dev, non_blocking=True)
                        if step_graphs:
                            key = (tuple(feats.shape), tuple(y.shape), n_prefix, trainable, sums is not None)
                            vals = self._graph_step(key, [feats, y], steps[k], forward, main)
                        else:
                            loss, acc = self.model(feats, y, rng_step=steps[k], n_prefix=n_prefix)
                            self._step(loss)
                            vals = [loss, acc]
                        if sums is not None and not fused:
                            self._accumulate(sums, vals, len(batch[0]))
                        if k == len(group) - 1:
                            slot.consumed = torch.cuda.Event()
                            slot.consumed.record(main)
                            launch_next()
                        yield vals, len(batch[0])
        finally:
            outer.wait_stream(main)

    def _run(self, dataset, train, print_interval):
        asr = self._is_asr(dataset)
        names = (["phoneme loss", "word loss", "phoneme acc", "word acc"] if asr
                 else ["intent loss", "intent acc"])
        dev = next(self.model.parameters()).device
        num_examples = 0
        self.model.train(train)
        if train and not asr:
            if self.rank == 0:
                self.model.print_frozen()
        it = dataset.loader
        if train:
            # data-parallel / bucketed samplers reshuffle per epoch from (seed, epoch)
            for smp in (getattr(it, "sampler", None), getattr(it, "batch_sampler", None)):
                if hasattr(smp, "set_epoch"):
                    smp.set_epoch(self.epoch)
        if train and self.rank == 0:
            it = tqdm(it)
        # closing(): if the loop is left early (exception, KeyboardInterrupt) the generator's finally clause
        # runs NOW — the current stream returns to the caller's and it waits for the training stream
        # the batch_size-weighted sums of the metrics (reference training.py:100-104) stay on the device:
        # self.epoch_sums, filled by the step loop itself (inside the captured step's kernels where it can)
        seq2seq = not asr and getattr(self.model, "seq2seq", False)
        batches_read = collections.deque()
        if seq2seq:                      # the decoded strings need the batch the step consumed
            # the step loop reads AHEAD of the step it yields (look-ahead super-batches, grouped evaluation) but
            # yields exactly once per batch, in reading order: the batch of the idx-th yield is the idx-th one read
            def tee(loader):
                for b in loader:
                    batches_read.append(b)
                    yield b
            src = it
            it = tee(src)
        string_acc = 0.0
        with contextlib.closing(self._iterate(it, train, asr, accumulate=True)) as steps:
            for idx, (vals, batch_size) in enumerate(steps):
                num_examples += batch_size
                current = batches_read.popleft() if seq2seq else None
                if train and idx % print_interval == 0 and self.rank == 0:
                    step_vals = vals if torch.is_tensor(vals) else [v.detach().reshape(()) for v in vals]
                    for n, v in zip(names, [float(v) for v in step_vals]):   # one host sync per print interval
                        print(n + ": " + str(v))
                    if seq2seq:          # reference training.py:103-112: show one decoded utterance
                        self._say_seq2seq_sample(current)
                if seq2seq and not train and self.epoch > 1:
                    # reference training.py:158-164: from the third epoch on the test accuracy is the fraction of
                    # utterances whose beam-search string equals the label string
                    x, y = current
                    guess = self.model.decode_intents(x)
                    truth = [self.model.one_hot_to_string(y[i], self.model.Sy_intent) for i in range(batch_size)]
                    hit = sum(g == t for g, t in zip(guess, truth)) / batch_size
                    string_acc += hit * batch_size
                    self._say("acc: " + str(hit))
                    self._say("guess: " + guess[0])
                    self._say("truth: " + truth[0])
        comm = self.bucket.comm if self.bucket is not None else None
        if train and comm is not None and hasattr(comm, "status"):
            # the epoch statistics are read here anyway: the device is idle.  The verdict is AGREED over the control plane
            # (MAX): a rank whose wait timed out must not raise alone — its peers saw its flags, finished with status 0 and
            # would sit in the next collective (_epoch_means) until the control plane's own timeout
            st = comm.status()
            worst = dp.agreed_status(st)
            if worst != 0:
                self.poisoned = True           # every step since the timeout applied unreduced gradients: no checkpoint
                raise RuntimeError("data parallel: %s gave up waiting for rank %d inside the gradient all-reduce (bounded "
                                   "wait, slu_comm_allreduce_ipc): the replicas are out of step — every rank stops here"
                                   % ("rank %d" % self.rank if st != 0 else "a peer of rank %d" % self.rank,
                                      (st if st != 0 else worst) - 1))
        means = self._epoch_means(self.epoch_sums[:len(names)].tolist() + [string_acc], num_examples, dev)
        if not all(math.isfinite(float(m)) for m in means[:len(names)]):
            import models
            if models.frozen_math_mode() == "f16x2":
                warnings.warn("non-finite epoch metrics with SLU_FROZEN_MATH=f16x2 (the UNGUARDED form of the scheme: "
                              "operands must stay below 65504 in magnitude) — the default mode guards the range and falls "
                              "back to bf16x3 by itself")
        if seq2seq and not train:
            means[1] = means[1] + means[-1]          # intent_acc += string accuracy (the model's own acc is 0)
        return means[:len(names)]

    def _say_seq2seq_sample(self, batch):
        """Reference training.py:103-112: decode the first utterance of the batch the step just consumed."""
        x, y = batch
        import models
        was_training = self.model.training
        step = models._DropoutState.step          # the sample must not shift the training run's dropout streams
        self.model.eval()
        try:
            print("seq2seq output")
            print("guess: " + self.model.decode_intents(x[:1])[0])
            print("truth: " + self.model.one_hot_to_string(y[0].cpu(), self.model.Sy_intent))
        finally:
            self.model.train(was_training)
            models._DropoutState.step = step

    # -- reference API -----------------------------------------------------------------------------
    def train(self, dataset, print_interval=100):
        if self._is_asr(dataset):
            phone_loss, word_loss, phone_acc, word_acc = self._run(dataset, True, print_interval)
            self.log({"phone_loss": phone_loss, "phone_acc": phone_acc, "word_loss": word_loss,
                      "word_acc": word_acc, "set": "train"})
            self.epoch += 1
            return phone_acc, phone_loss, word_acc, word_loss
        intent_loss, intent_acc = self._run(dataset, True, print_interval)
        before = [p.requires_grad for p in self.model.parameters()]
        self.model.unfreeze_one_layer()
        if self.bucket is not None and before != [p.requires_grad for p in self.model.parameters()]:
            self.bucket.reset()            # the set of parameters receiving gradients changed
        self.log({"intent_loss": intent_loss, "intent_acc": intent_acc, "set": "train"})
        self.epoch += 1
        return intent_acc, intent_loss

    def test(self, dataset):
        if self._is_asr(dataset):
            phone_loss, word_loss, phone_acc, word_acc = self._run(dataset, False, 0)
            self.log({"phone_loss": phone_loss, "phone_acc": phone_acc, "word_loss": word_loss,
                      "word_acc": word_acc, "set": "valid"})
            return phone_acc, phone_loss, word_acc, word_loss
        intent_loss, intent_acc = self._run(dataset, False, 0)
        self.log({"intent_loss": intent_loss, "intent_acc": intent_acc, "set": "valid"})
        return intent_acc, intent_loss
