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
import os

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


def _lookahead_width(depth, batch_size):
    """Batches per look-ahead super-batch: explicit, or as many as give every CU the side streams may use
    two 4-sequence recurrence workgroups (its register-limited occupancy) per direction, i.e. 4 sequences
    per CU: 768 on the 192 CUs of the default partition (best of a 4..14 sweep on MI355X)."""
    if depth > 0:
        return depth
    from slu_hip import pipeline
    cus = 256
    if torch.cuda.is_available():
        cus = pipeline.n_compute_units(torch.cuda.current_device()) - pipeline.cu_split()
    return max(2, min(32, (4 * cus) // max(1, batch_size)))


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
            from slu_hip.optim import HipAdam
            self.optimizer = HipAdam(model.parameters(), lr=self.lr)
        else:
            self.optimizer = torch.optim.Adam(model.parameters(), lr=self.lr)
        self.epoch = 0
        self.df = None
        self.rank, self.world_size = dp.world()
        # flat gradient bucket: the unit of the per-step all-reduce under data parallelism, and (on a
        # GPU) what gives the gradients fixed addresses for hipGraph-captured steps
        self.bucket = dp.GradBucket(model.parameters()) if (self.world_size > 1 or on_gpu) else None
        self._step_graphs, self._eager_steps = {}, {}

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
        if self.bucket is not None:
            self.bucket.allreduce_mean()
        self.optimizer.step()

    def _is_asr(self, dataset):
        return isinstance(dataset, ASRDataset)

    def _say(self, text):
        if self.rank == 0:
            print(text)

    def _epoch_means(self, sums, num_examples, device):
        tot = dp.allreduce_sums([float(s) for s in sums] + [float(num_examples)], device)
        return [v / tot[-1] for v in tot[:-1]]

    def _forward_losses(self, batch, asr):
        if asr:
            x, y_phoneme, y_word = batch
            phoneme_loss, word_loss, phoneme_acc, word_acc = self.model(x, y_phoneme, y_word)
            ptype = self.config.pretraining_type
            loss = {1: phoneme_loss, 3: word_loss}.get(ptype)
            if ptype == 2:
                loss = phoneme_loss + word_loss
            return [phoneme_loss, word_loss, phoneme_acc, word_acc], loss
        x, y_intent = batch
        intent_loss, intent_acc = self.model(x, y_intent)
        return [intent_loss, intent_acc], intent_loss

    def lookahead_depth(self, train, asr):
        """How many batches ahead the FROZEN prefix of the encoder is evaluated on side HIP streams
        (0 = plain sequential steps).  Only SLU training with a frozen prefix qualifies: a frozen
        stage's output does not depend on earlier optimisation steps.  At 64 utterances per step a
        recurrence occupies a fraction of the 256 CUs, so several batches' encoders run as one
        super-batch; the per-batch dropout streams are step-indexed, so the result is the sequential
        one.  depth -1 = automatic: as many batches as make a super-batch of ~512 utterances (one
        4-sequence recurrence workgroup per CU and direction)."""
        depth = _lookahead_env()
        if not train or asr or depth in (0, 1) or not hasattr(self.model, "prefix_features"):
            return 0, 0
        if not all(p.is_cuda for p in self.model.parameters()) or models_masks_injected():
            return 0, 0
        n = self.model.frozen_prefix_len()
        return (depth, n) if n > 0 else (0, 0)

    def _iterate(self, loader, train, asr):
        """Yields ([metric tensors], batch_size) per batch, doing the optimisation step when `train`."""
        depth, n_prefix = self.lookahead_depth(train, asr)
        group_eval = (not train and not asr and hasattr(self.model, "eval_group") and not models_masks_injected()
                      and all(p.is_cuda for p in self.model.parameters())
                      and _lookahead_env() not in (0, 1))
        if group_eval:
            # evaluation has no step-to-step dependency at all: whole batches are grouped
            group = []

            def flush():
                res = self.model.eval_group([b[0] for b in group], [b[1] for b in group])
                out = [([l, a], len(b[0])) for (l, a), b in zip(res, group)]
                group.clear()
                return out

            for batch in loader:
                if group and (tuple(batch[0].shape) != tuple(group[0][0].shape)
                              or len(group) == _lookahead_width(_lookahead_env(), len(group[0][0]))):
                    yield from flush()
                group.append(batch)
            if group:
                yield from flush()
            return
        if depth == 0:
            for batch in loader:
                with torch.set_grad_enabled(train):
                    vals, loss = self._forward_losses(batch, asr)
                    if train:
                        self._step(loss)
                yield vals, len(batch[0])
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
            n_slots = max(2, int(os.environ.get("SLU_LOOKAHEAD_SLOTS", "2")))
            self._slots = [pipeline.PrefixSlot(dev) for _ in range(n_slots)]   # in-flight super-batches
        pm = self.model.pretrained_model
        with torch.cuda.stream(main):
            pm.warm_weight_caches()
        frozen = [p for st in pm._stages()[:n_prefix] for p in st.parameters()]
        signature = tuple(p._version for p in frozen)
        for slot in self._slots:
            if slot.signature != signature:          # frozen weights were reloaded: re-capture
                slot.invalidate()
                slot.signature = signature
            slot.stream.wait_stream(main)
        use_graph = pipeline.graphs_enabled()
        pending = collections.deque()
        it = iter(loader)
        carry = []                                    # a batch read ahead that did not fit its group
        launched = 0
        last_done = [None]

        def launch_next():
            """Read up to `depth` equally-shaped batches and start their frozen prefix as one super-batch."""
            nonlocal launched
            group = [carry.pop()] if carry else []
            while not group or len(group) < _lookahead_width(depth, len(group[0][0])):
                try:
                    batch = next(it)
                except StopIteration:
                    break
                if group and tuple(batch[0].shape) != tuple(group[0][0].shape):
                    carry.append(batch)
                    break
                group.append(batch)
            if not group:
                return False
            slot = self._slots[launched % len(self._slots)]
            launched += 1
            steps = [next_rng_step() for _ in group]                        # consecutive by construction
            feats, done = slot.run(self.model, [b[0] for b in group], n_prefix, steps[0], use_graph,
                                   after=last_done[0])
            last_done[0] = done
            pending.append((group, feats, done, steps, slot))
            return True

        # The consumer's per-step work (metric accumulation in _run) runs with `main` as the current
        # stream: ordered after the step without touching the default stream, whose legacy
        # synchronisation with blocking streams (the CU-masked ones) would serialise the pipeline.
        try:
            with torch.cuda.stream(main):
                for _ in self._slots:
                    launch_next()
                while pending:
                    group, feats_cat, done, steps, slot = pending.popleft()
                    B = group[0][0].shape[0]
                    for k, batch in enumerate(group):
                        if k == 0:
                            main.wait_event(done)
                            feats_cat.record_stream(main)
                        feats = feats_cat[:, k * B:(k + 1) * B] if len(group) > 1 else feats_cat
                        y = batch[1]
                        key = (tuple(feats.shape), tuple(y.shape), n_prefix)
                        sg = self._step_graphs.get(key) if use_graph else None
                        if sg is not None and sg.signature != self.bucket.signature:
                            sg = None                                   # trainable set changed since capture
                        if sg is None and use_graph and self._eager_steps.get(key, 0) >= 3 and self.bucket.active:
                            try:
                                sg = pipeline.StepGraph(self, feats, y, n_prefix, main)
                                self._step_graphs[key] = sg
                            except Exception as e:                      # keep training eagerly if capture fails
                                print("hipGraph capture of the training step failed (%s); staying eager" % (e,))
                                self._eager_steps[key] = -(1 << 30)
                        if sg is not None:
                            vals = sg.run(feats, y, steps[k])          # (2,) device tensor [loss, acc]
                        else:
                            self._eager_steps[key] = self._eager_steps.get(key, 0) + 1
                            loss, acc = self.model.forward_from(feats, n_prefix, y, steps[k])
                            self._step(loss)
                            vals = [loss, acc]
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
        sums = torch.zeros(len(names), dtype=torch.float64, device=dev)
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
        for idx, (vals, batch_size) in enumerate(self._iterate(it, train, asr)):
            num_examples += batch_size
            if torch.is_tensor(vals):                              # captured step: one fused accumulate
                step_vals = vals
                sums.add_(vals, alpha=batch_size)
            else:
                step_vals = torch.stack([v.detach().to(dev).double().reshape(()) for v in vals])
                sums += step_vals * batch_size
            if train and idx % print_interval == 0 and self.rank == 0:
                for n, v in zip(names, step_vals.tolist()):       # one host sync per print interval
                    print(n + ": " + str(v))
        return self._epoch_means(sums.tolist(), num_examples, dev)

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
