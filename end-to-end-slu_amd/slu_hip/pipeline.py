"""Encoder look-ahead: the FROZEN prefix of the encoder is evaluated for the next few batches in one
go (a "super-batch") on a side HIP stream, optionally replayed from a captured hipGraph, while the
trainable remainder of each step runs on the training stream.

Why: with 64 utterances per step a GRU recurrence is a few dozen persistent workgroups on a 256-CU
chip and the step is a chain of ~580 dependent recurrence steps — latency-bound, a few per cent of the
machine.  The outputs of frozen stages do not depend on earlier optimisation steps, so P upcoming
batches can share one pass: the recurrences launch P times as many workgroups at the same latency, the
GEMMs and convolutions see P times larger (more efficient) problems, and two slots alternate so that
the next super-batch is computed while the current one is consumed step by step.  The two halves run
on CU-partitioned streams (cu_split / cu_range_stream): without the partition the super-batch kernels
occupy every CU and each small kernel of the training step queues behind them.  A prefix forward is
~60 kernel launches with fixed shapes and no autograd, i.e. an ideal hipGraph: replaying it costs the
host a few microseconds instead of ~1 ms of Python dispatch.  Dropout inside a captured prefix reads
its Philox offset (step*16) from device memory, so every replay draws the masks of its own step.
"""
import contextlib
import os

import torch

from . import ops



@contextlib.contextmanager
def capture(graph, stream):
    """torch.cuda.graph(...) in thread-local capture mode WITH THE GARBAGE COLLECTOR OFF.  A captured step runs its
    backward pass in autograd's worker thread; a collection that starts there (or anywhere) during the capture may
    finalise objects whose destructors talk to the runtime — a pinned host buffer returning to torch's host allocator
    (event record / query), a graph of an earlier trainer — which is illegal while a capture is in progress and ends in
    an abort of the process (seen once: "Fatal Python error: Aborted ... Garbage-collecting" inside a captured backward).
    torch.cuda.graph collects once before the capture starts; nothing may be collected until it has ended."""
    import gc
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            yield
    finally:
        if was:
            gc.enable()

def cu_split(device=None):
    """CUs [0, n) of the device are reserved for the trainable part of the step, the look-ahead
    super-batches get the rest.  Without the partition the big frozen-prefix kernels keep every CU
    busy and each of the ~25 small, latency-bound kernels of the trainable part waits for workgroup
    slots (measured: the two streams then take almost the SUM of their times).  Default: three eighths of the device
    (96 of 256 CUs = 12 per XCD; the mask bits interleave over the 8 XCDs, odd counts per XCD split CU pairs between the
    partitions and are markedly slower) with 20-batch super-batches (one recurrence workgroup per look-ahead CU).
    Round 3 ran 128 + 128 with 16-batch super-batches:
    Round 3, with the frozen stages on the f16x2 scheme (their share of a step fell from 0.20 to 0.145 ms, the trainable
    part's ten launches became the longer side; tools/cu_split_sweep.sh, 512 steps, same box, CUs + batches -> k utt/s):
    96 + 20: 296-299, 112 + 16: 311, 128 + 16: 316-321, 128 + 12: 303, 144 + 14: 226, 160 + 12: 257, 80 + 22: 272,
    64 + 24: 253 (the driver's 20-step command: 200 / 192 / 197 / 199 / 188 / 185).  Round 2 (bf16x3 frozen stages):
    96 + 20 was the optimum (278-281; 128: 223).  SLU_CU_SPLIT=n overrides, 0 = off."""
    v = os.environ.get("SLU_CU_SPLIT", "auto")
    if v != "auto":
        return int(v)
    if not torch.cuda.is_available():
        return 0
    n = n_compute_units(torch.cuda.current_device() if device is None else device)
    # three eighths, rounded to whole CU PAIRS per XCD (a multiple of 16: the mask bits interleave over 8 XCDs and an
    # odd count per XCD splits a pair between the partitions); 256 CUs -> 96 (+ 160 for the look-ahead super-batches of
    # 20 batches).  Round 4 (tools/cu_split_sweep.sh, same box; suffix CUs + batches -> steady-state k utt/s | the driver's
    # 20-step command): 128 + 16: 386 | 227; 112 + 16: 385 | 228; 96 + 20: 381-384 | 234-236; 96 + 18: 385 | 232;
    # 96 + 22: 354 | 233; 112 + 18 / 20: 310 / 316 | 230 / 223; 128 + 18: 307 | 228; 144 + 14 / 16: 247 / 262 | 212; 80 + 22:
    # 315 | 205 — with the faster frozen stages the steady state is flat from 96 to 128 and short runs prefer the wider
    # look-ahead partition (their first super-batch is pure pipeline fill).
    return max(16, (n * 3 // 8) // 16 * 16) if n >= 32 else 0


_CU_MASK_BROKEN = [False]


def cu_range_stream(device, first, count, **stream_kw):
    """A HIP stream confined to CUs [first, first + count) (slu_stream_create_cu_range), wrapped for torch.
    The partition is a scheduling aid, not a correctness requirement: if the runtime refuses the mask
    (e.g. a driver without CU-mask support) an ordinary stream is used and a warning printed once."""
    import ctypes
    from . import lib as _lib
    if not _CU_MASK_BROKEN[0]:
        L = _lib.load()
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = L.slu_stream_create_cu_range(first, count, ctypes.byref(h))
        if rc == 0 and h.value:
            ops.STREAM_CUS[h.value] = int(count)
            return torch.cuda.ExternalStream(h.value, device=device)
        _CU_MASK_BROKEN[0] = True
        print("warning: CU-masked streams unavailable (%s); the look-ahead pipeline shares all CUs"
              % L.slu_last_error().decode(errors="replace"))
    return torch.cuda.Stream(device, **stream_kw)


_N_CU = {}


def n_compute_units(device):
    key = str(device)
    if key not in _N_CU:                       # (asked once per batch by the look-ahead loop)
        _N_CU[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return _N_CU[key]


class PrefixSlot:
    """One in-flight SUPER-BATCH: the frozen prefix of the encoder evaluated for several upcoming
    batches at once (concatenated along the batch axis) on this slot's side stream.  The recurrence
    kernels then launch #batches times as many workgroups (one per CU and direction) and every other kernel sees a
    proportionally larger problem, at (nearly) the latency of a single batch."""
    MAX_GRAPHS = 4          # distinct (super-batch shape, prefix length, mode) keys kept per slot
    # batches a super-batch may read through the row-pointer table (slu_store_u64 carries 64 words since ABI 9); wider
    # super-batches copy their batches into one input.  SLU_MAX_TABLE=63 admits 32 - 63 batches (measured in round 6 with
    # the two-tile recurrence: 40-batch super-batches 347 - 349 k utt/s against 347 - 356 k for the default 20 — no gain, so
    # the default width, and with it the tested envelope of the table, stay where they were)
    MAX_TABLE = max(1, min(63, int(os.environ.get("SLU_MAX_TABLE", "31"))))

    def __init__(self, device):
        self.device = device
        n = cu_split()
        self.stream = (cu_range_stream(device, n, n_compute_units(device) - n) if n > 0
                       else torch.cuda.Stream(device))
        # unmasked: the FIRST super-batch of a run is replayed here (SLU_RAMP_WHOLE_CHIP, _run) — the training partition
        # has nothing to do until that super-batch is through
        self.whole = torch.cuda.Stream(device) if n > 0 else None
        # [row-pointer table of a super-batch (MAX_TABLE entries) | step0*16, read by the dropout kernels]
        self.words = torch.zeros(self.MAX_TABLE + 1, dtype=torch.int64, device=device)
        self.rng = self.words[self.MAX_TABLE:]
        self.graphs = {}
        self.seen = {}             # how often each super-batch shape was requested
        self.consumed = None       # event: the main stream is done with this slot's last output
        self.signature = None      # versions of the frozen parameters the graphs were captured with
        self.capture_failures = 0
        # host batches: optionally their H2D copies run on a stream of their own, AHEAD of the compute (which is
        # serialised behind the previous super-batch): the copy of super-batch n + 1 overlaps the kernels of super-batch n
        # Measured (tools/host_inputs_probe.py, round 3): OFF by default.  With the copies on a stream of their own the
        # H2D rate drops to ~17 GB/s while the other slot's kernels run (33 GB/s on an idle device), and the loop gets
        # SLOWER (0.73 vs 0.56 ms/step; any CU mask for the copy stream: 16 / 32 / 160 CUs or none) than with the copies
        # serialised on the slot's own stream — the host-input path is PCIe-bound either way.  SLU_COPY_CUS=n > 0 enables
        # the separate copy stream on the top n CUs of the look-ahead partition (>= its size: unmasked).
        n_copy = int(os.environ.get("SLU_COPY_CUS", "0"))
        total = n_compute_units(device)
        if n_copy <= 0:
            self.copy_stream = None
        elif n > 0 and n_copy < total - n:
            self.copy_stream = cu_range_stream(device, total - n_copy, n_copy)
        else:
            self.copy_stream = torch.cuda.Stream(device)
        self.last_done = None      # event: this slot's previous super-batch has finished (its static input is free)
        # range guard of the f16x2 scheme (slu_hip/guard.py): armed at the start and copied to pinned host memory at the
        # end of every guarded super-batch (both inside the captured graph); the training loop reads it when it consumes
        # the slot and re-runs the super-batch on bf16x3 after a violation
        from . import guard as _guard
        self.guard = _guard.RangeGuard(device)

    def invalidate(self):
        self.graphs = {}
        self.seen = {}

    def run(self, model, xs, n_prefix, step0, use_graph, after=None, guarded=True, whole_chip=False):
        """-> (features, done event, guard or None).  guarded=False: the re-run of a super-batch whose range words
        reported a violation — no guard scope, so the default arithmetic resolves to bf16x3.  whole_chip: the caller
        knows the training partition to be idle until this super-batch is through (the first one of a run)."""
        import models as _models
        pm = getattr(model, "pretrained_model", model)
        guard = self.guard if (guarded and hasattr(pm, "f16x2_allowed") and pm.f16x2_allowed()) else None
        with _models.frozen_math_scope(guard):
            feats, done = self._run(model, xs, n_prefix, step0, use_graph, after, guard, whole_chip)
        return feats, done, guard

    def _run(self, model, xs, n_prefix, step0, use_graph, after, guard, whole_chip=False):
        """Enqueue stages [0, n_prefix) for the batches `xs` (equal shapes; consecutive dropout steps
        step0, step0+1, ...) on this slot's stream, after the event `after` (the previous super-batch:
        two super-batches side by side would only delay the one the training step is waiting for).
        whole_chip: replay an ALREADY CAPTURED super-batch on the unmasked stream (device-resident batches only).  Rounds
        2 and 3 tried this and saw nothing (round 3, f16x2, 12 batches on 256 instead of 128 CUs: 2.27 vs 2.48 ms for
        the super-batch, 199.1 k utt/s either way — inside the run-to-run spread of the 20-step figure); round 6 measured
        the bf16x3 timeline of the driver's command (profiles/r06_y_timeline_k20.txt): 14 batches spend 1.55 of 2.79 ms in
        throughput-bound convolution / projection kernels on 160 CUs while 96 CUs idle.
        Returns (features of the concatenated batch, event recorded when they are complete)."""
        B, T = xs[0].shape
        host = not all(x.is_cuda for x in xs)
        whole = (whole_chip and self.whole is not None and not host and use_graph
                 and os.environ.get("SLU_RAMP_WHOLE_CHIP", "1") != "0")
        stream = self.stream
        if whole:
            stream = self.whole
            stream.wait_stream(self.stream)         # this slot's earlier work (normally long finished)
        with torch.cuda.stream(stream):
            if self.consumed is not None:
                stream.wait_event(self.consumed)
            if after is not None and not host:
                stream.wait_event(after)
            feats = None
            if use_graph:
                import models as _models
                # (a graph is captured PER STREAM: replaying one on another stream than the last time costs ~12 ms on
                # ROCm 7 — measured, profiles/r06_y_timeline_k20.txt — so the whole-chip replay has a graph of its own)
                key = (len(xs), B, T, n_prefix, bool(model.training), _models.contraction_nsplit(True),
                       self._table_ok(model, xs), xs[0].dtype, whole)
                entry = self.graphs.get(key)
                # capture a shape on its second appearance in this slot: a one-off shape (the ragged last
                # group of an epoch, a short run) is cheaper launched eagerly than captured (~10 ms)
                self.seen[key] = self.seen.get(key, 0) + 1
                if (entry is None and key not in self.graphs and self.seen[key] >= 2
                        and len(self.graphs) < self.MAX_GRAPHS):
                    entry = self._capture(model, xs, n_prefix, step0, key, guard, stream)
                if entry is not None:
                    graph, x_static, feats = entry
                    if isinstance(x_static, ops.RowTable):
                        # the batches are read where they lie: refresh the address table and the dropout-stream
                        # offset (adjacent words of one buffer) in ONE launch instead of one copy per batch
                        ops.store_u64(self.words, [x.data_ptr() for x in xs] + [0] * (self.MAX_TABLE - len(xs)) + [step0 * 16])
                    else:
                        self._copy_in(x_static, xs, host, after)
                        self.rng.fill_(step0 * 16)
                    graph.replay()
            if feats is None:
                x_cat = torch.empty(len(xs) * B, T, dtype=xs[0].dtype, device=self.device)
                self._copy_in(x_cat, xs, host, after)
                if guard is not None:
                    guard.arm()
                feats = model.prefix_features(x_cat, n_prefix, step0, sub_batch=B if len(xs) > 1 else 0)
                if guard is not None:
                    guard.collect()
            done = torch.cuda.Event()
            done.record(stream)
            self.last_done = done
        if stream is not self.stream:
            self.stream.wait_event(done)            # the slot's own stream stays ordered behind what ran elsewhere
        return feats, done

    def _copy_in(self, dst, xs, host, after):
        """Fill the super-batch's input.  Device batches: plain copies on this slot's stream.  Host (pinned) batches: on
        the copy stream, as soon as this slot's previous super-batch has released the buffer — not behind `after`
        (the OTHER slot's super-batch, which is computing right now); the compute then waits for the copy and `after`."""
        if not host or self.copy_stream is None:
            if host and after is not None:
                self.stream.wait_event(after)
            self._fill(dst, xs)
            return
        cs = self.copy_stream
        cs.wait_stream(self.stream)                  # dst's allocation / the consumed-wait enqueued so far
        if self.last_done is not None:
            cs.wait_event(self.last_done)
        with torch.cuda.stream(cs):
            self._fill(dst, xs)
        dst.record_stream(cs)
        self.stream.wait_stream(cs)
        if after is not None:
            self.stream.wait_event(after)

    @staticmethod
    def _fill(x_cat, xs):
        B = xs[0].shape[0]
        for k, x in enumerate(xs):
            if x.dtype != x_cat.dtype:
                # copy_ would CONVERT: float [-1, 1) samples into an int16 buffer are all zeros, int16 PCM into a float
                # buffer is 32768 times too large — a super-batch holds one sample format (training.launch_next)
                raise TypeError("look-ahead super-batch of %s batches received a %s batch" % (x_cat.dtype, x.dtype))
            x_cat[k * B:(k + 1) * B].copy_(x, non_blocking=True)

    def _table_ok(self, model, xs):
        """Device-resident, contiguous fp32 batches and a first stage that can read through a row-pointer table: the
        captured graph then reads the batches in place (they must stay alive and unchanged until it has run: the
        training loop keeps them until their steps are done)."""
        pm = getattr(model, "pretrained_model", model)
        return (os.environ.get("SLU_ROW_TABLE", "1") != "0" and 1 < len(xs) <= self.MAX_TABLE
                and hasattr(pm, "accepts_row_table") and pm.accepts_row_table()
                and xs[0].dtype in (torch.float32, torch.int16)
                and all(x.is_cuda and x.device == self.device and x.dtype == xs[0].dtype and x.is_contiguous()
                        and x.data_ptr() % 16 == 0 for x in xs))

    def _capture(self, model, xs, n_prefix, step0, key, guard=None, stream=None):
        stream = self.stream if stream is None else stream
        B, T = xs[0].shape
        sub = B if len(xs) > 1 else 0
        if key[6]:
            x_static = ops.RowTable(self.words[:len(xs)], B, T, xs[0].dtype)
            ops.store_u64(self.words, [x.data_ptr() for x in xs] + [0] * (self.MAX_TABLE - len(xs)) + [step0 * 16])
        else:
            x_static = torch.empty(len(xs) * B, T, dtype=xs[0].dtype, device=self.device)
            self._fill(x_static, xs)
            self.rng.fill_(step0 * 16)
        model.prefix_features(x_static, n_prefix, self.rng, sub_batch=sub)      # warm-up (lazy initialisation)
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            # thread-local capture mode: other threads (e.g. the RCCL watchdog) may keep calling the
            # runtime while this thread captures
            with capture(graph, stream):
                if guard is not None:
                    guard.arm()                     # memset node
                feats = model.prefix_features(x_static, n_prefix, self.rng, sub_batch=sub)
                if guard is not None:
                    guard.collect()                 # device -> pinned host copy node
        except RuntimeError as e:                   # stay eager for this shape
            print("hipGraph capture of the frozen prefix failed (%s); staying eager" % (e,))
            self.graphs[key] = None
            self.capture_failures += 1
            return None
        entry = (graph, x_static, feats)
        self.graphs[key] = entry
        return entry


class RangeTrip(RuntimeError):
    """A captured step's range guard reported a violation (slu_hip/guard.py): the forward / backward graph has run on
    f16x2 operands that left the scheme's range, the optimiser has NOT run — the caller repeats the step."""

    def __init__(self, overflow, seen):
        super().__init__("f16x2 range guard tripped")
        self.overflow, self.seen = overflow, seen


class StepGraph:
    """One optimisation step captured as hipGraph(s).  Single process: ONE graph (forward, loss, backward, Adam).
    Data parallel: two graphs around an eager collective by default; ONE graph with the RCCL all-reduce as a node when
    SLU_DP_GRAPH=1 asks for it and the self-test passes (dp.GradBucket.collective_in_graph has the measurement and the
    reason it is opt-in):
    G1 = forward, loss, backward, packing;  [all-reduce, eager];  G2 = Adam.  Used for the trainable remainder of an SLU
    step (inputs = prefix features + labels), for a fully trainable SLU step (waveforms + labels) and
    for an ASR pre-training step (waveforms + phoneme / word labels): a B = 64 step is ~100 short
    kernels, i.e. host-bound when launched one by one.  Inputs and the dropout step are static device
    buffers refreshed before each replay.

    forward(static_inputs, rng_dev) -> (metrics, loss): `metrics` a 1-D device tensor (what the epoch
    statistics accumulate), `loss` the 0-d tensor to back-propagate."""

    def __init__(self, trainer, inputs, forward, stream, forks=False, guard=None):
        """forks: keep the backward pass' independent branches (ops._Fork) as parallel graph branches —
        for steps that have the device to themselves (no look-ahead streams beside them).
        guard: a RangeGuard when the step contains FROZEN stages that run on guarded f16x2 (a step without look-ahead:
        SLU_LOOKAHEAD=0, or a CNN-block dropout inside the frozen prefix): the guard is armed and collected inside G1, the
        optimiser always gets a graph of its own, and run() reads the guard between the two — a violation raises RangeTrip
        before any parameter has moved."""
        import models as _models
        self.guard = guard
        dev = next(trainer.model.parameters()).device
        self.trainer = trainer
        self.inputs = [torch.empty(tuple(t.shape), dtype=t.dtype, device=dev) for t in inputs]
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src)
        self.rng = torch.zeros(1, dtype=torch.int64, device=dev)
        from . import dp
        bucket = trainer.bucket
        assert bucket is not None and bucket.active
        self.dp = dp.data_parallel()
        # the all-reduce as a graph node (decided once per trainer, the same on every rank)
        self.collective_in_graph = self.dp and bucket.collective_in_graph(dev)
        torch.cuda.synchronize()
        bucket.release_grads()
        self.one = torch.ones((), dtype=torch.float32, device=dev)      # root gradient (no per-step fill)
        self.g1 = torch.cuda.CUDAGraph()
        aux = ops._aux_streams(dev, 3)              # created before the capture starts
        for st_ in [stream] + (aux[:1] if forks else []):
            with torch.cuda.stream(st_):
                ops.tn_tickets(dev)                 # likewise: the split-K weight-gradient launch's ticket words of this stream
        ops._Fork.capture_forks = bool(forks)
        try:
            with capture(self.g1, stream):
                if guard is not None:
                    guard.arm()
                with _models.frozen_math_scope(guard):
                    self.metrics, self.loss = forward(self.inputs, self.rng)
                self.loss.backward(self.one)
                ops._Fork.join(dev)                 # branches the backward pass left open (ops._Fork.defer): joined ONCE, here
                if guard is not None:
                    guard.collect()
                if self.dp:
                    bucket.pack()                   # one concatenation kernel per dtype; .grad -> slices
                    if self.collective_in_graph and guard is None:
                        bucket.allreduce_flats()    # one collective per gradient dtype, captured; the mean's 1 / N is
                        trainer.optimizer.step()    # folded into the Adam kernel (HipAdam.grad_div)
                elif _one_graph() and guard is None:
                    # single process: nothing sits between the backward pass and Adam, so they are ONE graph — a graph
                    # launch costs the stream ~8 us of ramp (SLU_ONE_STEP_GRAPH=0: two graphs as under data parallelism)
                    trainer.optimizer.step()
        finally:
            ops._Fork.capture_forks = False
        self.g2 = None
        one = (self.collective_in_graph if self.dp else _one_graph()) and guard is None
        if not one:
            self.g2 = torch.cuda.CUDAGraph()
            with capture(self.g2, stream):
                if self.collective_in_graph:        # guarded step: the range check sits between backward and the collective
                    bucket.allreduce_flats()
                trainer.optimizer.step()
        bucket.observe()
        self.signature = bucket.signature

    def run(self, inputs, step):
        # static inputs + dropout-stream offset in ONE launch (device-resident sources; others fall back to copy_)
        for dst, src in zip(self.inputs, inputs):
            if dst.dtype != src.dtype:          # the graph was captured for one sample format (float32 or PCM16)
                raise TypeError("captured step expects %s inputs, got %s" % (dst.dtype, src.dtype))
        for dst, src in ops.stage_inputs(list(zip(self.inputs, inputs)), self.rng, step * 16):
            dst.copy_(src, non_blocking=True)
        self.g1.replay()
        if self.guard is not None:
            torch.cuda.current_stream().synchronize()
            overflow, quiet, seen = self.guard.verdict()
            if overflow or quiet:
                raise RangeTrip(overflow, seen)
        if self.dp and not self.collective_in_graph:
            self.trainer.bucket.allreduce_flats()   # eager, between the two graphs
        if self.g2 is not None:
            self.g2.replay()
        return self.metrics


def _one_graph():
    return os.environ.get("SLU_ONE_STEP_GRAPH", "1") != "0"


def graphs_enabled():
    return os.environ.get("SLU_GRAPHS", "1") != "0"
