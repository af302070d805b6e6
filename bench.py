#!/usr/bin/env python3
"""Benchmark of the hot path: SLU training steps (utterances/s) on synthetic 3 s @16 kHz waveforms,
B = 64 per GPU — BASELINE.json `metric`, workload = configs[3] (no_unfreezing: frozen pre-trained
SincNet/conv/biGRU encoder in train mode + intent GRU trained; forward, loss, backward, gradient
all-reduce, Adam) — through models.Model / training.Trainer on the HIP kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload no_unfreezing|unfreeze_all|asr_pretrain]

`--gpus N` with N > 1 launches the N ranks itself (one process per GPU, RCCL over xGMI) unless the
process was already started by torch.distributed.run (WORLD_SIZE set), which works as well:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `value` = utterances all ranks processed / max-over-ranks wall time of the
K timed steps (inputs resident in HBM, barrier + synchronize on both sides).
`roofline` describes the kernel with the largest share of the step's GPU time and lists the top
three (`kernels`): algorithmic flops of each distinct launch shape of the timed steps / its average
duration, measured with HIP events on the CU-masked stream the kernel runs on, the launches enqueued
back to back (a hipGraph of 20) so that no host dispatch gap is counted.  `cpu_baseline` times the CPU
oracle (torch-CPU restatement of the reference path) on a bounded sample of the same workload on this
host's cores.  `parity` = the golden batch of BASELINE configs[0] (tests/golden/g6, generated from the
reference) through this very process' kernels: max-abs logit deviation and predicted-intent equality.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "end-to-end-slu_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz
BATCH = 64
SECONDS = 3
FS = 16000
CFG = {"no_unfreezing": "no_unfreezing_synthetic.cfg", "unfreeze_all": "unfreeze_all_layers_synthetic.cfg",
       "asr_pretrain": "unfreeze_all_layers_synthetic.cfg", "seq2seq": "seq2seq_synthetic.cfg"}
WORKLOAD_TEXT = {
    "no_unfreezing": "experiments/no_unfreezing.cfg SLU train step (frozen SincNet/conv/biGRU encoder in train "
                     "mode + intent biGRU trained): fwd + 3-slot CE + bwd + grad all-reduce + Adam",
    "unfreeze_all": "SLU train step with every encoder layer unfrozen (unfreeze_all_layers end state): fwd + CE "
                    "+ full bwd + grad all-reduce + Adam",
    "seq2seq": "seq2seq SLU train step (experiments/all_real_seq2seq.cfg sizes: frozen encoder + intent biGRU encoder 128 + "
               "2-layer attention decoder 256 over ~35 output characters, 24 teacher-forced steps): fwd + loss + bwd + Adam",
    "asr_pretrain": "ASR pre-training step of the full PretrainedModel (BASELINE configs[2]): fwd + phoneme/word "
                    "CE heads (vocabulary 10 000) + full bwd + grad all-reduce + Adam"}


# ------------------------------------------------------------------------------------------------------
# setup
# ------------------------------------------------------------------------------------------------------
def setup(workload, rank, batch, samples, n_batches, hidden=0):
    """read_config on the package's synthetic cfg, synthetic pre-training checkpoint, Model, Trainer.
    hidden > 0: every GRU layer gets that hidden size (a synthetic variant, not a reference cfg)."""
    import data
    import models
    import training

    work = tempfile.mkdtemp(prefix="slu_bench_")
    os.makedirs(os.path.join(work, "experiments"))
    name = CFG[workload]
    shutil.copy(os.path.join(PKG, "experiments", name), os.path.join(work, "experiments", name))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        config = data.read_config(os.path.join("experiments", name))
        config.folder = os.path.join(work, config.folder)
        for sub in ("", "pretraining", "training"):
            os.makedirs(os.path.join(config.folder, sub), exist_ok=True)
        config.seed = 1234 + rank                      # per-rank synthetic data
        if hidden:
            config.phone_rnn_num_hidden = [hidden] * len(config.phone_rnn_num_hidden)
            config.word_rnn_num_hidden = [hidden] * len(config.word_rnn_num_hidden)
            config.intent_rnn_num_hidden = [hidden] * len(config.intent_rnn_num_hidden)
        if workload == "asr_pretrain":
            config.asr_path = "synthetic:%dx%dx%d" % (n_batches, batch, samples)
            train_ds, _, _ = data.get_ASR_datasets(config)
            torch.manual_seed(1234)
            model = models.PretrainedModel(config)
        else:
            config.slu_path = "synthetic:%dx%dx%d" % (n_batches, batch, samples)
            train_ds, _, _ = data.get_SLU_datasets(config)
            torch.manual_seed(4321)                    # synthetic "pre-trained" encoder (no .pth published)
            torch.save({k: v.cpu() for k, v in models.PretrainedModel(config).state_dict().items()},
                       os.path.join(config.folder, "pretraining", "model_state.pth"))
            torch.manual_seed(1234)
            model = models.Model(config)
            if workload == "unfreeze_all":
                for p in model.pretrained_model.phoneme_layers.parameters():
                    p.requires_grad = True
                for p in model.pretrained_model.word_layers.parameters():
                    p.requires_grad = True
        models.set_dropout_seed(1234 + 7919 * rank)
        trainer = training.Trainer(model=model, config=config)
    finally:
        os.chdir(cwd)
    return config, model, trainer, train_ds, work


def run_steps(model, trainer, batches, n, asr=False, first_done=None):
    """n optimisation steps through Trainer's own step loop (forward, loss, backward, gradient
    all-reduce, Adam; with the frozen-encoder look-ahead pipeline when it applies).
    first_done: optional timing event recorded when the first step has been enqueued completely."""
    import contextlib
    loader = [batches[i % len(batches)] for i in range(n)]
    # accumulate=True: the epoch statistics are kept on the device by the loop itself (trainer.epoch_sums), as
    # Trainer.train() runs it — no accumulation launch per step here
    with contextlib.closing(trainer._iterate(loader, True, asr, accumulate=True)) as steps:
        for i, _ in enumerate(steps):
            if i == 0 and first_done is not None:
                first_done.record()
    return trainer.epoch_sums.clone()


# ------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = torch-CPU restatement of the reference path); the ONLY user of oracle/ here
# ------------------------------------------------------------------------------------------------------
def host_topology():
    """lscpu model / sockets / cores / threads of the box the baseline runs on."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        want = {"Model name": "model", "Socket(s)": "sockets", "Core(s) per socket": "cores_per_socket",
                "Thread(s) per core": "threads_per_core", "CPU(s)": "logical_cpus"}
        for line in out.splitlines():
            k, _, v = line.partition(":")
            if k.strip() in want:
                v = v.strip()
                info[want[k.strip()]] = int(v) if v.isdigit() else v
        if "sockets" in info and "cores_per_socket" in info:
            info["physical_cores"] = info["sockets"] * info["cores_per_socket"]
    except Exception as e:                           # lscpu missing: keep os.cpu_count()
        info["lscpu_error"] = str(e)
    return info


def cpu_baseline_seq2seq(config, batch, samples, max_len=24, budget_s=20.0):
    """The CPU oracle on the seq2seq workload: frozen encoder (train mode), seq2seq encoder + attention decoder
    trained — forward, loss, backward, Adam (torch-CPU restatement of reference models.py:381-557, 825-828)."""
    from oracle import slu_oracle as O
    import data
    threads = max(1, min(torch.get_num_threads(), 64))
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    labels = list(data.SYNTHETIC_SEQ2SEQ_LABELS)
    V = len(labels)
    sd = {"pretrained_model." + k: v for k, v in O.init_pretrained_state_dict(config).items()}
    head = O.init_seq2seq_state_dict(config, V)
    for v in head.values():
        v.requires_grad_()
    sd.update(head)
    opt = torch.optim.Adam(list(head.values()), lr=1e-3)
    ds = data.SyntheticSeq2SeqDataset(1, batch, samples, max_len=max_len, seed=1234, Sy_intent=labels)
    x, y = ds.batches[0]

    def step():
        masks = O.draw_seq2seq_masks(config, x, y.shape[1], seed=1)
        opt.zero_grad()
        loss, _ = O.seq2seq_forward(sd, x, y, config, masks, explicit_gru=False, SOS=labels.index("<sos>"))
        loss.backward()
        opt.step()

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < budget_s * 0.6 and n < 40):
        step()
        n += 1
    rate = n * batch / (time.perf_counter() - t0)
    return {"value": round(rate, 2), "unit": "utterances/s", "cores": threads, "kind": "port", "host": host_topology(),
            "sample": "%d seq2seq train steps (after 1 warm-up) of B=%d x %d s, %d teacher-forced steps, on %d torch-CPU threads "
                      "(oracle: ATen GRU encoder, explicit attention decoder)" % (n, batch, samples // FS, max_len, threads)}


def cpu_baseline(config, batch, samples, budget_s=20.0):
    """The CPU oracle (torch-CPU restatement of the reference path, ATen GRU like the reference) on
    the same workload: train-mode forward + backward of the trainable part + Adam, on the host's cores."""
    from oracle import slu_oracle as O
    # torch's default intra-op pool (what the reference would run with on this host), capped: with
    # one thread per SMT sibling (256 on the MI355X host) ATen's small per-step GRU matmuls thrash.
    threads = max(1, min(torch.get_num_threads(), 64))
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    sd = O.init_model_state_dict(config)
    trainable = [k for k in sd if k.startswith("intent_layers")]
    if config.unfreezing_type == 2:
        trainable = [k for k in sd if not k.startswith(("pretrained_model.phoneme_linear", "pretrained_model.word_linear"))]
    for k in trainable:
        sd[k].requires_grad_()
    opt = torch.optim.Adam([sd[k] for k in trainable], lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(batch, samples, generator=g)
    y = torch.stack([torch.randint(0, n, (batch,), generator=g) for n in config.values_per_slot], dim=1)

    def step(faithful):
        masks = O.draw_dropout_masks(config, x, seed=1)        # the reference draws them per step too
        opt.zero_grad()
        loss, _, _, _ = O.slu_forward(sd, x, y, config, masks, explicit_gru=False, faithful_sinc=faithful)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step(False)                                                # warm-up (thread pool, allocator, mkldnn primitives)
    first = time.perf_counter() - t0
    warm = 1
    while warm < 3 and first * (warm + 1) < budget_s * 0.25:
        step(False)
        warm += 1
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < budget_s * 0.5 and n < 50):
        step(False)
        n += 1
    dedup = n * batch / (time.perf_counter() - t0)
    step(True)                                                 # warm-up of the 80-convolution variant
    t0 = time.perf_counter()
    nf = 0
    while nf < 2 or (time.perf_counter() - t0 < budget_s * 0.2 and nf < 10):
        step(True)
        nf += 1
    faithful = nf * batch / (time.perf_counter() - t0)
    return {"value": round(dedup, 2), "unit": "utterances/s", "cores": threads, "kind": "port",
            "host": host_topology(),
            "sample": "%d train steps (after %d warm-up) of B=%d x %d s on %d torch-CPU threads (oracle, ATen GRU, "
                      "one conv per Sinc forward); reference-faithful variant with the 80 redundant in-loop "
                      "convolutions (models.py:98-108), %d steps after 1 warm-up: %.2f utterances/s"
                      % (n, warm, batch, samples // FS, threads, nf, faithful),
            "faithful_sinc_value": round(faithful, 2)}


# ------------------------------------------------------------------------------------------------------
# parity of this process' kernels on the reference's golden batch (BASELINE configs[0], fixture g6)
# ------------------------------------------------------------------------------------------------------
def parity_check(dev):
    """no_unfreezing architecture under the reference's seeds (state_dict SHA-256 pinned by the fixture),
    x = 0.1 randn(16, 16000) by seed: eval-mode logits vs the REFERENCE's (stored in
    tests/golden/g6_full_model.npz by tests/golden/make_goldens.py), predicted intents bit-identical."""
    import hashlib
    import numpy as np
    import data
    import models
    path = os.path.join(ROOT, "tests", "golden", "g6_full_model.npz")
    d = dict(np.load(path))
    meta = json.loads(bytes(d["meta_json"]).decode())
    work = tempfile.mkdtemp(prefix="slu_parity_")
    cwd = os.getcwd()
    try:
        os.makedirs(os.path.join(work, "experiments"))
        name = CFG["no_unfreezing"]
        shutil.copy(os.path.join(PKG, "experiments", name), os.path.join(work, "experiments", name))
        os.chdir(work)
        config = data.read_config(os.path.join("experiments", name))
        config.folder = os.path.join(work, config.folder)
        config.values_per_slot = [6, 14, 4]
        config.Sy_intent = data.synthetic_Sy_intent(config.values_per_slot)
        config.num_phonemes = 42                                   # no_unfreezing.cfg's phoneme inventory
        torch.manual_seed(meta["pretrain_seed"])
        torch.save({k: v.cpu() for k, v in models.PretrainedModel(config).state_dict().items()},
                   os.path.join(config.folder, "pretraining", "model_state.pth"))
        torch.manual_seed(meta["model_seed"])
        model = models.Model(config)
    finally:
        os.chdir(cwd)
    sha = {k: hashlib.sha256(v.cpu().contiguous().numpy().tobytes()).hexdigest() for k, v in model.state_dict().items()}
    weights_ok = sha == meta["model_sha256"]
    g = torch.Generator().manual_seed(1234)
    x = 0.1 * torch.randn(16, 16000, generator=g)
    model.eval()
    with torch.no_grad():
        logits, pred = model.predict_intents(x)
    dev_max = float((logits.cpu().double() - torch.from_numpy(d["eval.logits"]).double()).abs().max())
    same = bool(np.array_equal(pred.cpu().numpy(), d["eval.pred"]))
    shutil.rmtree(work, ignore_errors=True)
    del model
    return {"max_abs_logit_dev": dev_max, "intents_equal": same, "weights_sha256_match": weights_ok,
            "tolerance": 1e-4, "golden": "tests/golden/g6_full_model.npz (reference outputs, BASELINE configs[0])"}


# ------------------------------------------------------------------------------------------------------
# per-kernel roofline: each distinct launch shape of the step, back to back on the stream it runs on
# ------------------------------------------------------------------------------------------------------
def _timed_graph(fn, stream, reps=20):
    """Average duration (ms) of one fn() launch: `reps` launches captured as one hipGraph on `stream`
    (the CU-masked stream the kernel runs on in the timed steps), replayed between two HIP events on that
    stream — kernels run back to back, no host dispatch gap is counted."""
    with torch.cuda.stream(stream):
        fn()
        fn()
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        from slu_hip import pipeline as _pl
        with _pl.capture(graph, stream):
            for _ in range(reps):
                fn()
        graph.replay()                              # warm replay
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        graph.replay()
        e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del graph
    return ms


PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (fp16: the same rate)
MFMA_PRODUCTS = {1: 1.0, 2: 3.0, 3: 6.0}     # 16-bit MFMA products issued per algorithmic product, by split scheme
PEAK_HBM_TBS = 8.0                 # HBM3E spec (6.3 TB/s achievable per the same guide)


def kernel_table(model, trainer, batch, samples, width, asr=False):
    """Top kernels of the step by GPU time with the algorithmic flops and the measured average duration of
    every distinct launch shape (frozen stages at `width` x batch sequences on the look-ahead stream,
    trainable stages at `batch` sequences on the training stream).  Each stage is launched through the same
    wrappers, with the same arithmetic (exact fp32 MFMA, or the split-precision bf16 MFMA kernels for frozen
    GRU layers), as in the timed steps."""
    import models
    from slu_hip import ops
    dev = next(model.parameters()).device
    pm = model.pretrained_model if hasattr(model, "pretrained_model") else model
    n_prefix = pm.frozen_prefix_len() if width > 1 else 0
    slots = getattr(trainer, "_slots", None)
    side = slots[0].stream if slots else torch.cuda.Stream(dev)
    main = getattr(trainer, "_train_stream", None) if width > 1 else getattr(trainer, "_full_stream", None)
    main = main or torch.cuda.Stream(dev)
    rows = {}
    stages = pm._stages() + list(getattr(model, "_intent_stages", []))
    L, C = samples, 1
    for si, st in enumerate(stages):
        frozen = si < n_prefix
        B = batch * (width if frozen else 1)
        stream = side if frozen else main
        where = "%d sequences, look-ahead stream" % B if frozen else "%d sequences, training stream" % B
        if hasattr(st, "conv"):                                         # CNN block
            conv = st.conv
            if st.is_sinc:
                w = conv.filters().view(conv.N_filt, 1, conv.Filt_dim)
                bias, c_out, k, stride = None, conv.N_filt, conv.Filt_dim, conv.stride
            else:
                w, bias, c_out, k, stride = conv.weight.detach(), conv.bias.detach(), conv.out_channels, conv.kernel_size, conv.stride
            x = torch.randn(B, L, C, device=dev) * 0.1
            l_conv = ops.conv_out_len(L, k, stride)
            pool = st.pool if st.pool in (1, 2) else 1
            tm = si == len(pm._cnn_stages) - 1
            conv_frozen = not any(q.requires_grad for q in conv.parameters())
            ns = models.guarded_frozen_nsplit(model) if conv_frozen else (1 if ops.bf16_mode() else 0)
            if ns and ops.wconv_bf16_supported(C, stride, pool, k, ns):
                ms = _timed_graph(lambda: ops.wconv_fwd_bf16(x, w, bias, B, L, C, stride, st.do_abs, pool, st.slope, tm, ns), stream)
                name, mult, peak = "wconv_bf_fwd_kernel<%d>" % ns, MFMA_PRODUCTS[ns], PEAK_BF16_MFMA_TFLOPS
            else:
                ms = _timed_graph(lambda: ops.wconv_fwd(x, w, bias, B, L, C, stride, st.do_abs, pool, st.slope, tm, False), stream)
                name, mult, peak = "wconv_fwd_kernel", 1.0, PEAK_FP32_MFMA_TFLOPS
            rows.setdefault(name, []).append(
                {"shape": "B=%d L=%d Cin=%d Cout=%d k=%d stride=%d (%s)" % (B, L, C, c_out, k, stride, where),
                 "flops": 2.0 * B * l_conv * c_out * k * C, "ms": ms, "mfma_mult": mult, "peak": peak,
                 "bytes": 4.0 * (B * L * C + B * (-(-l_conv // pool)) * c_out)})
            L, C = -(-l_conv // pool), c_out
            del x
        else:                                                           # GRU layer
            gru = st.gru
            H, D, I = gru.hidden_size, 2 if gru.bidirectional else 1, gru.input_size
            T = L
            is_frozen = not any(q.requires_grad for q in gru.parameters())
            ns = models.guarded_frozen_nsplit(model) if is_frozen else models.contraction_nsplit(False)
            if not ops.split_path_supported(H, D):
                ns = 0
            w_ih, b_ih = gru._stacked_ih()
            w_ih, b_ih = w_ih.detach(), b_ih.detach()
            x = torch.randn(T * B, I, device=dev)
            N = D * 3 * H
            wr = gru.weight_hh_l0_reverse.detach() if D == 2 else None
            br = gru.bias_hh_l0_reverse.detach() if D == 2 else None
            gx = torch.randn(T, B, N, device=dev)
            fused_in = bool(ns) and is_frozen and ops.gru_fused_input_ok(I, H, D, ns)
            # Dropout + Downsample(avg, 2) in the recurrence's epilogue (round 4): the layer's launch writes the pooled output
            # (the next frozen layer's 16-bit planes, or fp32 for the trainable part) from a 1-bit mask drawn by a small launch
            nxt = stages[si + 1] if si + 1 < len(stages) else None
            nxt_frozen = (nxt is not None and hasattr(nxt, "gru") and si + 1 < n_prefix
                          and not any(q.requires_grad for q in nxt.gru.parameters()))
            pool_fused = bool(ns) and is_frozen and ops.gru_pool_fused_ok(H, D, T, st.p, None, st.method, st.factor)
            T_out = -(-T // st.factor)
            if pool_fused:
                keep = ops.dropout_bits(T, B, D * H, st.p, 1234, 16 + st.site, None, batch if B > batch else 0, dev) if st.p > 0 else None
                if keep is not None:
                    ms = _timed_graph(lambda: ops.dropout_bits(T, B, D * H, st.p, 1234, 16 + st.site, None, batch if B > batch else 0, dev), stream)
                    rows.setdefault("dropout_bits_kernel", []).append(
                        {"shape": "T=%d B=%d C=%d keep bits (%s)" % (T, B, D * H, where), "flops": 0.0, "ms": ms,
                         "mfma_mult": 1.0, "peak": PEAK_FP32_MFMA_TFLOPS, "bytes": T * B * D * H / 8.0})
                out_bytes = (2.0 * ns if nxt_frozen else 4.0) * T_out * B * D * H + (T * B * D * H / 8.0 if keep is not None else 0.0)
            if fused_in:
                # the first frozen GRU layer (K = 60): the recurrence computes x W_ih^T + b_ih itself (no projection
                # launch, no gx round trip): flops of both contractions, bytes = the input planes + the output
                planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
                mult = MFMA_PRODUCTS[ns]
                if pool_fused:
                    ms = _timed_graph(lambda: ops.gru_seq_fwd_pool_bf16(None, gru.weight_hh_l0.detach(), wr, gru.bias_hh_l0.detach(), br,
                                                                        T, B, H, D, ns, keep, st.p, nxt_frozen,
                                                                        fused=(planes, I, packed, b_ih)), stream)
                else:
                    ms = _timed_graph(lambda: ops.gru_seq_fwd_bf16(None, gru.weight_hh_l0.detach(), wr, gru.bias_hh_l0.detach(), br,
                                                                   T, B, H, D, ns, False, fused=(planes, I, packed, b_ih)), stream)
                rows.setdefault("gru_bf_fwd_kernel<%d,%d>" % (H, ns), []).append(
                    {"shape": "T=%d B=%d H=%d D=%d K=%d fused input projection%s (%s)" % (T, B, H, D, I, " + dropout/pool epilogue" if pool_fused else "", where),
                     "flops": 2.0 * B * H * 3 * H * D * T + 2.0 * T * B * N * I, "ms": ms,
                     "mfma_mult": mult, "peak": PEAK_BF16_MFMA_TFLOPS,
                     "bytes": 2.0 * ns * T * B * ops.round_up(I, 32) + (out_bytes if pool_fused else 4.0 * T * B * D * H) + 4.0 * D * 3 * H * H + 2.0 * ns * N * ops.round_up(I, 32)})
            elif ns:
                planes, packed = ops.split_bf16(x, ns), ops.gemm_bf16_pack(w_ih, ns)
                mult = MFMA_PRODUCTS[ns]
                ms = _timed_graph(lambda: ops.gemm_bf16(planes, packed, b_ih, N, I), stream)
                kc = ops.round_up(I, 32) // 32
                # K <= 64: the row-panel kernel (A resident in LDS, HBM-write bound); else the tiled kernel
                if kc <= 2:
                    gemm_name = "gemm_bf_panel_kernel<%d,%d>" % (ns, kc)
                elif kc in (4, 8) and T * B >= 16 * 1024 and os.environ.get("SLU_GEMM_PANEL96", "1") != "0":
                    gemm_name = "gemm_bf_panel96_kernel<%d,%d>" % (ns, kc)        # 96-row panels, A resident in LDS
                else:
                    gemm_name = "gemm_bf_kernel<%d>" % ns
                rows.setdefault(gemm_name, []).append(
                    {"shape": "M=%d N=%d K=%d input projection, %d 16-bit plane(s) (%s)" % (T * B, N, I, ns, where),
                     "flops": 2.0 * T * B * N * I, "ms": ms, "mfma_mult": mult, "peak": PEAK_BF16_MFMA_TFLOPS,
                     "bytes": 2.0 * ns * T * B * ops.round_up(I, 32) + 2.0 * ns * N * ops.round_up(I, 32) + 4.0 * T * B * N})
                if pool_fused:
                    ms = _timed_graph(lambda: ops.gru_seq_fwd_pool_bf16(gx, gru.weight_hh_l0.detach(), wr, gru.bias_hh_l0.detach(), br,
                                                                        T, B, H, D, ns, keep, st.p, nxt_frozen), stream)
                else:
                    ms = _timed_graph(lambda: ops.gru_seq_fwd_bf16(gx, gru.weight_hh_l0.detach(), wr, gru.bias_hh_l0.detach(), br,
                                                                   T, B, H, D, ns, not is_frozen), stream)
                rows.setdefault("gru_bf_fwd_kernel<%d,%d>" % (H, ns), []).append(
                    {"shape": "T=%d B=%d H=%d D=%d%s (%s)" % (T, B, H, D, " + dropout/pool epilogue" if pool_fused else "", where),
                     "flops": 2.0 * B * H * 3 * H * D * T, "ms": ms,
                     "mfma_mult": mult, "peak": PEAK_BF16_MFMA_TFLOPS,
                     "bytes": 4.0 * T * B * N + (out_bytes if pool_fused else 4.0 * T * B * D * H) + 4.0 * D * 3 * H * H})
            else:
                ms = _timed_graph(lambda: ops.gemm(x, w_ih.t(), b_ih), stream)
                rows.setdefault("gemm_f32_kernel<true,true,2>", []).append(
                    {"shape": "M=%d N=%d K=%d input projection (%s)" % (T * B, N, I, where), "flops": 2.0 * T * B * N * I, "ms": ms,
                     "mfma_mult": 1.0, "peak": PEAK_FP32_MFMA_TFLOPS, "bytes": 4.0 * (T * B * I + N * I + T * B * N)})
                ms = _timed_graph(lambda: ops.gru_seq_fwd(gx, gru.weight_hh_l0.detach(), wr, gru.bias_hh_l0.detach(), br,
                                                          T, B, H, D, not is_frozen), stream)
                rows.setdefault("gru_seq_fwd4_kernel<%d>" % H, []).append(
                    {"shape": "T=%d B=%d H=%d D=%d (%s)" % (T, B, H, D, where), "flops": 2.0 * B * H * 3 * H * D * T, "ms": ms,
                     "mfma_mult": 1.0, "peak": PEAK_FP32_MFMA_TFLOPS, "bytes": 4.0 * (T * B * N + T * B * D * H + D * 3 * H * H)})
            # Dropout + Downsample after the layer as launches of their own (trainable layers; pooling modes the epilogue
            # does not cover): frozen -> frozen hand-off writes bf16 planes
            if (st.p > 0.0 or st.factor > 1) and not (bool(ns) and is_frozen and ops.gru_pool_fused_ok(H, D, T, st.p, None, st.method, st.factor)):
                nxt = stages[si + 1] if si + 1 < len(stages) else None
                nxt_frozen = (nxt is not None and hasattr(nxt, "gru") and si + 1 < n_prefix
                              and not any(q.requires_grad for q in nxt.gru.parameters()))
                raw = torch.randn(T, B, D * H, device=dev)
                T_out = -(-T // st.factor)
                if ns and is_frozen and nxt_frozen and (D * H) % 32 == 0:
                    ms = _timed_graph(lambda: ops.dropout_pool_fwd_planes(raw, None, st.p, 1234, 16 + st.site, st.method, st.factor, ns), stream)
                    name, wbytes = "dropout_pool_fwd4_kernel<%d>" % ns, 2.0 * ns * T_out * B * D * H
                else:
                    ms = _timed_graph(lambda: ops.dropout_pool_fwd(raw, None, st.p, 1234, 16 + st.site, st.method, st.factor), stream)
                    name, wbytes = "dropout_pool_fwd4_kernel<0>", 4.0 * T_out * B * D * H
                rows.setdefault(name, []).append(
                    {"shape": "T=%d B=%d C=%d %s/%d (%s)" % (T, B, D * H, st.method, st.factor, where), "flops": 0.0, "ms": ms,
                     "mfma_mult": 1.0, "peak": PEAK_FP32_MFMA_TFLOPS, "bytes": 4.0 * T * B * D * H + wbytes})
                del raw
            L, C = -(-T // st.factor), D * H
            del x, gx
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    out = []
    for name, shapes in rows.items():
        # per optimisation step: a frozen-stage launch serves `width` steps
        t_step = sum(s["ms"] / (width if "look-ahead" in s["shape"] else 1) for s in shapes)
        flops = sum(s["flops"] for s in shapes)
        mfma = sum(s["flops"] * s["mfma_mult"] for s in shapes)
        byts = sum(s["bytes"] for s in shapes)
        ms = sum(s["ms"] for s in shapes)
        peak = shapes[0]["peak"]
        tf_mfma = mfma / (ms * 1e-3) / 1e12              # 16-bit MFMA products actually issued (x 3 / x 6 on the split schemes)
        tf_alg = flops / (ms * 1e-3) / 1e12              # algorithmic (fp32-equivalent) work rate
        tbs = byts / (ms * 1e-3) / 1e12
        # `frac` prices ALGORITHMIC work: fp32-equivalent flops against the peak of the MFMA unit the kernel issues to, and
        # algorithmic bytes against HBM; the issued-product utilisation of the unit is reported beside it
        f_mfma, f_issued, f_hbm = tf_alg / peak, tf_mfma / peak, tbs / PEAK_HBM_TBS
        # the kernel runs on its stream's CU partition only: its MFMA fraction of THAT partition's share of the peak
        # (HBM is shared by the whole chip: no such rescaling)
        n_cu, split = _n_cus(), _cu_split()
        cus = (n_cu - split if "look-ahead" in shapes[0]["shape"] else split) if 0 < split < n_cu else n_cu
        out.append({"kernel": name, "launches_per_cycle": len(shapes), "algorithmic_gflop": round(flops / 1e9, 2),
                    "cus": cus, "mfma_issued_frac_of_partition_peak": round(f_issued * n_cu / cus, 4),
                    "avg_us": round(1e3 * ms / len(shapes), 2),
                    "algorithmic_tflops": round(tf_alg, 2),                           # fp32-equivalent work rate
                    "mfma_tflops": round(tf_mfma, 2), "mfma_peak": peak, "mfma_frac": round(f_mfma, 4),
                    "mfma_issued_frac": round(f_issued, 4),
                    "frac_of_fp32_mfma_peak": round(tf_alg / PEAK_FP32_MFMA_TFLOPS, 4),
                    "hbm_tbs": round(tbs, 3), "hbm_frac": round(f_hbm, 4),
                    "bound": "mfma" if f_mfma >= f_hbm else "hbm", "frac": round(max(f_mfma, f_hbm), 4),
                    "gpu_ms_per_step": round(t_step, 4),
                    "shapes": [{"shape": s["shape"], "gflop": round(s["flops"] / 1e9, 3), "us": round(1e3 * s["ms"], 2),
                                "algorithmic_tflops": round(s["flops"] / (s["ms"] * 1e-3) / 1e12, 2),
                                "algorithmic_MB": round(s["bytes"] / 1e6, 2),
                                "hbm_tbs": round(s["bytes"] / (s["ms"] * 1e-3) / 1e12, 3)} for s in shapes]})
    out.sort(key=lambda r: -r["gpu_ms_per_step"])
    return out


def dtype_label():
    """The arithmetic the path computes in (not a precision claim)."""
    import models
    if models.contraction_nsplit(False) == 1:
        return ("bf16 (operands of every forward contraction - convolutions, input projections, recurrences - and of the "
                "data-gradient contractions on bf16 MFMA; fp32 accumulation, gate math, weight gradients, master weights, Adam)")
    if models.guarded_frozen_nsplit() == 2:
        return ("f32 with 22-bit frozen operands (trainable stages: exact fp32 MFMA; convolutions and GRU contractions of FROZEN "
                "stages: f16x2 = fp32 operands as 2 fp16 terms, 22-bit significand - NARROWER than the reference's fp32 -, 3 fp16 "
                "MFMA products, fp32 accumulation; opt-in, SLU_FROZEN_MATH=auto)")
    if models.guarded_frozen_nsplit() == 3:
        return ("f32 (trainable stages: exact fp32 MFMA; convolutions and GRU contractions of FROZEN stages: bf16x3 = every fp32 "
                "operand split EXACTLY into 3 bf16 terms (3 x 8 = 24 significand bits, fp32's exponent range), 6 bf16 MFMA "
                "products per fp32 product (only terms below 2^-24 |a b| dropped), fp32 accumulation)")
    return "f32"


def _cu_split():
    from slu_hip import pipeline
    return pipeline.cu_split()


def models_nsplit():
    """Split scheme of the frozen stages in this process (3 = bf16x3, the default; 2 = f16x2; 0 = exact fp32)."""
    import models
    return models.guarded_frozen_nsplit() or 3


def _n_cus():
    from slu_hip import pipeline
    return pipeline.n_compute_units(torch.cuda.current_device())


# Committed evidence of the same measurements (named as `source`, never read at run time).
PMC_SOURCE = "profiles/r06_z_pmc_gru_bf.txt"                     # rocprofv3 --pmc passes of the dominant kernel (HBM bytes, SQ activity)
INLOOP_SOURCE = "profiles/r06_z_default_kernel_stats_by_shape.txt"  # rocprofv3 --kernel-trace of the default command (in-loop durations)


def pmc_dominant_kernel(n_seq, nsplit, timeout=150):
    """HBM traffic and SQ activity of the dominant kernel (the frozen GRU layers' split-precision recurrence, product form,
    the four layer shapes of one look-ahead super-batch) from rocprofv3 PMC counters, collected LIVE by this run: three
    separate passes (FETCH_SIZE; WRITE_SIZE; SQ set — the TCC counters do not fit one pass) of tools/run_one.py
    gru_frozen_layers in child processes, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
    (FETCH_SIZE counts 64 B per 128-B request: x 2; counter unit KiB).  -> dict, or {"error": ...} when the profiler is not
    available (the line then carries traffic = null and says why)."""
    import csv
    import glob
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return {"error": "rocprofv3 not on PATH"}
    work = tempfile.mkdtemp(prefix="slu_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("SLU_BENCH_VERBOSE", None)
    passes = {"fetch": "FETCH_SIZE", "write": "WRITE_SIZE",
              "sq": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"}
    got, launches, us = {}, 0, []
    try:
        for tag, counters in passes.items():
            cmd = [rocprof, "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(work, tag),
                                                           "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "run_one.py"),
                                                           "gru_frozen_layers", str(n_seq), str(nsplit)]
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=timeout)
            f = glob.glob(os.path.join(work, tag, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not f:
                return {"error": "rocprofv3 pass %s failed (rc %d): %s" % (tag, r.returncode, (r.stderr or r.stdout)[-160:])}
            acc = {}
            for row in csv.DictReader(open(f[0])):
                if "gru_bf_fwd" in row["Kernel_Name"] or "gru_bf2_fwd" in row["Kernel_Name"]:
                    acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            if not acc:
                return {"error": "no gru_bf_fwd dispatch in pass %s" % tag}
            launches = max(len(v) for v in acc.values())
            got.update({k: sum(v) for k, v in acc.items()})
            if tag == "sq":
                t = glob.glob(os.path.join(work, tag, "**", "*kernel_trace.csv"), recursive=True)
                us = [(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3 for row in csv.DictReader(open(t[0]))
                      if "gru_bf_fwd" in row["Kernel_Name"] or "gru_bf2_fwd" in row["Kernel_Name"]]
    except Exception as e:                                       # noqa: BLE001 - a side measurement never takes the headline down
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    H, D = 128, 2
    waves = -(-n_seq // 16) * D * (H // 16)
    steps = 3 * (300 + 150 + 75 + 38)                           # wave-steps per wave over the 12 launches
    fetch_b, write_b = 2.0 * got["FETCH_SIZE"] * 1024.0, got["WRITE_SIZE"] * 1024.0
    q = 4.0 / (waves * steps)                                   # SQ_* cycle counters tick once per four cycles
    out = {"launches_profiled": launches, "hbm_bytes_per_launch": round((fetch_b + write_b) / launches),
           "fetch_bytes_per_launch": round(fetch_b / launches), "write_bytes_per_launch": round(write_b / launches),
           "avg_launch_us_under_profiler": round(sum(us) / max(1, len(us)), 1),
           "per_wave_step_cycles": {"wave": round(q * got["SQ_WAVE_CYCLES"]), "issuing": round(q * got["SQ_ACTIVE_INST_ANY"]),
                                    "valu_active": round(q * got["SQ_ACTIVE_INST_VALU"]), "waiting": round(q * got["SQ_WAIT_ANY"]),
                                    "issue_stalled": round(q * got["SQ_WAIT_INST_ANY"]),
                                    "mfma_busy": round(got["SQ_VALU_MFMA_BUSY_CYCLES"] / (waves * steps))}}
    c = out["per_wave_step_cycles"]
    # two waves share a SIMD: the share of a step in which the SIMD's matrix pipe or VALU works for this kernel's dependent chain
    out["issue_frac"] = round(2.0 * (c["mfma_busy"] + c["valu_active"]) / max(1, c["wave"]), 4)
    return out


def side_run(extra_args, env_extra=None, timeout=600):
    """Another bench.py measurement in a fresh process (its own model, graphs and environment), reduced to the
    numbers the parent attaches to its JSON line.  None if it failed."""
    env = dict(os.environ)
    env.update(env_extra or {})
    env.pop("SLU_BENCH_VERBOSE", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--sub", "--no-cpu-baseline", "--no-large-batch", "--no-kernel-table"] + extra_args
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:                                   # a side measurement never takes the headline down
        return {"error": str(e)[:200]}
    keep = {k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "dtype") if k in d}
    if d.get("steady_state"):
        keep["steady_state"] = d["steady_state"]
    if d.get("parity"):
        keep["parity_max_abs_logit_dev"] = d["parity"]["max_abs_logit_dev"]
        keep["parity_intents_equal"] = d["parity"]["intents_equal"]
    return keep


def host_inputs_point(model, trainer, batches, steps, asr):
    """The same loop fed from PINNED HOST batches (what Trainer.train does with a CPU DataLoader; reference
    models.py:351-352, 802-803 move every batch to the device inside the step): the H2D copies are inside the clock
    (they run on the look-ahead slots' streams, beside the previous super-batch's steps)."""
    host = [tuple(t.cpu().pin_memory() for t in b) for b in batches]
    run_steps(model, trainer, host, steps, asr)              # the slots capture their copy-in graph for this input kind
    run_steps(model, trainer, host, steps, asr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(model, trainer, host, steps, asr)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(t.numel() * t.element_size() for t in host[0])
    out = {"utterances_per_s": round(len(host[0][0]) * steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 4), "steps": steps,
           "h2d_bytes_per_step": nbytes, "h2d_gb_per_s": round(nbytes * steps / dt / 1e9, 2),
           "note": "pinned host batches, H2D inside the timed region; `value` is quoted with inputs resident in HBM"}
    if not asr:
        # the same waveforms as PCM16 (what SLU_PCM16_BATCHES=1 loaders hand over for 16-bit wavs): int16 across PCIe, scaled by
        # 1 / 32768 inside the first stage — half the bytes
        host16 = [((b[0] * 32768.0).round().clamp(-32768, 32767).to(torch.int16).pin_memory(),) + tuple(b[1:]) for b in host]
        run_steps(model, trainer, host16, steps, asr)
        run_steps(model, trainer, host16, steps, asr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(model, trainer, host16, steps, asr)
        torch.cuda.synchronize()
        dt16 = time.perf_counter() - t0
        nb16 = sum(t.numel() * t.element_size() for t in host16[0])
        out["pcm16"] = {"utterances_per_s": round(len(host16[0][0]) * steps / dt16, 2), "ms_per_step": round(1e3 * dt16 / steps, 4),
                        "h2d_bytes_per_step": nb16, "h2d_gb_per_s": round(nb16 * steps / dt16 / 1e9, 2),
                        "note": "int16 PCM batches (SLU_PCM16_BATCHES=1), scaled on the device by the first stage"}
    return out


def feature_parity(model, config, batch, samples, n_batches=16):
    """Encoder features of ONE look-ahead super-batch (n_batches x batch utterances: 1024 x 3 s by default) in eval mode
    against the CPU oracle, for each arithmetic of the frozen stages — the number that tells the three modes apart (the
    16 x 1 s golden logits agree to one ulp in all of them).  The oracle runs once (a few seconds of host time)."""
    import models
    from oracle import slu_oracle as O
    pm = model.pretrained_model
    g = torch.Generator().manual_seed(4321)
    x = 0.1 * torch.randn(n_batches * batch, samples, generator=g)
    sd = {k: v.detach().cpu() for k, v in pm.state_dict().items()}
    was_training = model.training
    model.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = O.encoder_stages(sd, x, config, None, explicit_gru=False)["features"]
    cpu_s = time.perf_counter() - t0
    out = {"utterances": n_batches * batch, "samples": samples, "oracle_cpu_s": round(cpu_s, 1), "max_abs_dev": {}}
    old = os.environ.get("SLU_FROZEN_MATH")
    xd = x.cuda()
    try:
        for mode in ("bf16x3", "fp32", "auto"):
            os.environ["SLU_FROZEN_MATH"] = mode
            with torch.no_grad():
                got = pm.compute_features(xd).float().cpu()
            label = {"bf16x3": "default (bf16x3)", "fp32": "exact fp32 MFMA",
                     "auto": "f16x2 under the range guard (opt-in, 22-bit operands)"}[mode]
            out["max_abs_dev"][label] = float((got - ref).abs().max())
        out["guard_trips"] = pm.range_guard().trips
    finally:
        if old is None:
            os.environ.pop("SLU_FROZEN_MATH", None)
        else:
            os.environ["SLU_FROZEN_MATH"] = old
        model.train(was_training)
    out["default_frozen_arithmetic"] = {2: "f16x2", 3: "bf16x3", 0: "fp32"}.get(models.guarded_frozen_nsplit(model))
    del xd
    return out


def frontend_fwd_point(model, config, batch, samples):
    """BASELINE.json configs[1]: the SincNet + Conv1d front end of the phoneme module alone (reference models.py:77-110,
    180-220: sinc -> abs -> pool -> LeakyReLU -> conv -> LeakyReLU -> conv -> LeakyReLU), B = 64 x 3 s, forward only, on
    the whole device: time per batch, utterances/s, algorithmic HBM GB/s (waveform in + conv2 output out = SURVEY 8(d)'s
    minimum for these stages) and MFMA rate, in the default arithmetic of frozen stages (bf16x3), on the exact fp32 kernels
    and on guarded f16x2, with the CPU oracle's front end timed beside it."""
    import copy
    import models
    from oracle import slu_oracle as O
    pm = model.pretrained_model
    n_cnn = len(pm._cnn_stages)
    g = torch.Generator().manual_seed(99)
    x = 0.1 * torch.randn(batch, samples, generator=g)
    xd = x.cuda()
    stream = torch.cuda.Stream()
    l_out = samples
    flops = 0.0
    c_prev = 1
    for k, st in enumerate(pm._cnn_stages):
        conv = st.conv
        c_out = conv.N_filt if st.is_sinc else conv.out_channels
        k_t = conv.Filt_dim if st.is_sinc else conv.kernel_size
        l_conv = (l_out + 2 * (k_t // 2) - k_t) // conv.stride + 1
        flops += 2.0 * batch * l_conv * c_out * k_t * c_prev
        l_out, c_prev = -(-l_conv // st.pool), c_out
    nbytes = 4.0 * batch * samples + 4.0 * batch * l_out * c_prev
    was_training = model.training
    model.eval()
    res = {"workload": "SincNet + Conv1d front end of the phoneme module, forward only, B=%d x %.0f s (BASELINE configs[1])" % (batch, samples / 16000.0),
           "algorithmic_gflop": round(flops / 1e9, 3), "algorithmic_MB": round(nbytes / 1e6, 2)}
    old = os.environ.get("SLU_FROZEN_MATH")
    try:
        for mode, label in (("bf16x3", "default"), ("fp32", "exact_fp32"), ("auto", "f16x2_guarded")):
            os.environ["SLU_FROZEN_MATH"] = mode
            with torch.no_grad():
                if mode == "auto":
                    scope = models.frozen_math_scope(pm.range_guard() if pm.f16x2_allowed() else None)
                else:
                    scope = models.frozen_math_scope(None)
                with scope:
                    ms = _timed_graph(lambda: pm._run_stages(xd, 0, n_cnn), stream)
            res[label] = {"ms": round(ms, 4), "utterances_per_s": round(batch / (ms * 1e-3), 1),
                          "algorithmic_hbm_gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                          "algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                          "frac_of_fp32_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            if mode != "fp32":
                ns = 3 if mode == "bf16x3" else 2
                res[label]["mfma_issued_frac_of_16bit_peak"] = round(flops * MFMA_PRODUCTS[ns] / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
                res[label]["mfma_algorithmic_frac_of_16bit_peak"] = round(flops / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)
    finally:
        if old is None:
            os.environ.pop("SLU_FROZEN_MATH", None)
        else:
            os.environ["SLU_FROZEN_MATH"] = old
        model.train(was_training)
    # the oracle's front end on the host cores: the same three blocks (an encoder config without RNN layers)
    cfg = copy.copy(config)
    cfg.phone_rnn_num_hidden, cfg.word_rnn_num_hidden = [], []
    sd = {k: v.detach().cpu() for k, v in pm.state_dict().items()}
    n_thr = min(64, os.cpu_count() or 1)
    torch.set_num_threads(n_thr)
    with torch.no_grad():
        O.encoder_stages(sd, x, cfg, None, upto="phoneme_features")
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            O.encoder_stages(sd, x, cfg, None, upto="phoneme_features")
        cpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    res["cpu"] = {"ms": round(cpu_ms, 2), "utterances_per_s": round(batch / (cpu_ms * 1e-3), 1), "threads": n_thr,
                  "algorithmic_hbm_gbs": round(nbytes / (cpu_ms * 1e-3) / 1e9, 2),
                  "kind": "oracle (torch CPU, one conv per Sinc forward)"}
    return res


def large_batch_point(rank, samples, batch=2048, steps=3):
    """The same train step at B = 2048 per GPU (128 sequence tiles x 2 directions = 256 recurrence
    workgroups, one per CU): where the recurrence stops being bound by the 8-workgroup latency chain.
    Reported next to the headline number, never as `value`."""
    os.environ["SLU_LOOKAHEAD"] = "0"          # one big batch per step: nothing to look ahead to
    config, model, trainer, train_ds, work = setup("no_unfreezing", rank, batch, samples, 1)
    dev = next(model.parameters()).device
    batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
    model.train()
    run_steps(model, trainer, batches, 4)      # three eager steps, then the captured step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(model, trainer, batches, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    shutil.rmtree(work, ignore_errors=True)
    del model, trainer, batches
    torch.cuda.empty_cache()
    os.environ.pop("SLU_LOOKAHEAD", None)
    return {"batch_per_gpu": batch, "utterances_per_s": round(batch * steps / dt, 1),
            "ms_per_step": round(1e3 * dt / steps, 3)}


def note(msg):
    if os.environ.get("SLU_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()
_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Library code prints there too — the reference-style config messages
    (`read_config`), RCCL's version banner (C stdio, flushed at exit, i.e. AFTER the line) — and under torchrun all ranks
    share the pipe.  So: keep the real stdout aside for the JSON line and point file descriptor 1 at stderr for
    everything else, Python and C alike."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        try:
            saved = os.dup(1)
            os.dup2(2, 1)
            _JSON_FD = saved
        except OSError:                  # no usable stderr (or stdout): leave the descriptors alone, print as usual
            _JSON_FD = None


def emit_json(out):
    sys.stdout.flush()
    line = (json.dumps(out) + "\n").encode()
    if _JSON_FD is None:
        os.write(1, line)
    else:
        os.write(_JSON_FD, line)


# ------------------------------------------------------------------------------------------------------
# multi-GPU self-launch
# ------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start one rank per GPU (rank 0 owns stdout)."""
    n = args.gpus
    visible = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if visible < n:
        if not args.share_gpu:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible (--share-gpu runs all ranks on the "
                             "visible GPUs over gloo: a functional check of the multi-process path, not a measurement)"
                             % (n, visible))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        if visible < n:
            e["SLU_LOCAL_DEVICE"] = str(r % max(visible, 1))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in live:                 # a rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def _time_collective(call, stream, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        for _ in range(5):
            call()
        e0.record(stream)
        for _ in range(reps):
            call()
        e1.record(stream)
    e1.synchronize()
    return round(1e3 * e0.elapsed_time(e1) / reps, 2)


def scaling_model(model, trainer, width, steady_ms, batches=None):
    """What this ONE-GPU process can say about the data-parallel step on N GPUs: the two halves of the pipelined
    loop — the trainable suffix (the captured step graph on the training partition) and the frozen prefix (one super-batch
    graph on the look-ahead partition) — replayed ALONE and SIDE BY SIDE without dependencies (each stream between its own
    pair of HIP events: what the partitions cost each other), and the step's gradient all-reduce through the hand-written IPC
    plane with ONE rank (launch + three local passes + its flag protocol; no link traffic).  Under weak scaling every rank
    runs this same loop plus the collective on the training stream, so
        step(N) ~ step(1) + max(0, suffix + allreduce(N) - prefix)      (both side-by-side figures),
        efficiency(N) ~ step(1) / step(N).
    allreduce(8) is NOT measurable here: the line carries the one-rank call and the efficiency for an ASSUMED 8-rank call of
    that + 10 us (two cross-GPU flag hand-offs and two remote 16-byte round trips of ~2.5 us each), so
    that a SCALE record can be checked against a stated prediction."""
    from slu_hip import dp
    out = {}
    try:
        sg = next(iter(trainer._step_graphs.values()))
        main = trainer._train_stream
        # the WIDEST captured super-batch of any slot (a short run has captured its ramp sizes only)
        key, (graph, _x, _f), slot = max(((k, v, sl) for sl in trainer._slots for k, v in sl.graphs.items() if v is not None and not k[-1]),     # (k[-1]: whole-chip variant)
                                         key=lambda kv: kv[0][0])
        width = int(key[0])
        from slu_hip import ops as _ops
        if isinstance(_x, _ops.RowTable):
            # the graph reads its batches through the slot's row-pointer table: point ALL of its entries at live batches (the
            # table still holds whatever the slot's last super-batch — possibly a narrower one — wrote)
            xs = [batches[i % len(batches)][0] for i in range(width)]
            _ops.store_u64(slot.words, [x.data_ptr() for x in xs] + [0] * (slot.MAX_TABLE - width) + [16])
    except Exception as e:                                   # noqa: BLE001 - no captured pipeline (eager run): nothing to model
        return {"error": "no captured pipeline: %s" % str(e)[:100]}
    out["super_batch_width"] = width

    def run(n_prefix, n_suffix):
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if n_prefix:
            with torch.cuda.stream(slot.stream):
                ev[0].record(slot.stream)
                for _ in range(n_prefix):
                    graph.replay()
                ev[1].record(slot.stream)
        if n_suffix:
            with torch.cuda.stream(main):
                ev[2].record(main)
                for i in range(n_suffix):
                    sg.run(sg.inputs, 100000 + i)
                ev[3].record(main)
        torch.cuda.synchronize()
        return (ev[0].elapsed_time(ev[1]) / n_prefix / width if n_prefix else 0.0, ev[2].elapsed_time(ev[3]) / n_suffix if n_suffix else 0.0)

    run(2, 2 * width)
    out["prefix_ms_per_step_alone"] = round(run(24, 0)[0], 4)
    out["suffix_ms_per_step_alone"] = round(run(0, 24 * width)[1], 4)
    pb, sb = run(24, 24 * width)
    out["prefix_ms_per_step_side_by_side"], out["suffix_ms_per_step_side_by_side"] = round(pb, 4), round(sb, 4)
    b = trainer.bucket
    dev = next(model.parameters()).device
    try:
        comm = dp.IpcComm(0, 1, dev)
        flats = {k: torch.zeros_like(v) for k, v in b.flats.items()}
        out["allreduce_us_one_rank_ipc"] = _time_collective(lambda: comm.allreduce_flats(flats), main)
        out["allreduce_bytes"] = b.nbytes()
        comm.close()
    except Exception as e:                                   # noqa: BLE001
        out["allreduce_us_one_rank_ipc"] = None
        out["allreduce_error"] = str(e)[:120]
    if out.get("allreduce_us_one_rank_ipc"):
        ar8 = out["allreduce_us_one_rank_ipc"] + 10.0
        slack = 1e3 * (pb - sb)                              # what the suffix may grow by before it becomes the longer half
        step8 = steady_ms + max(0.0, 1e-3 * (ar8 - slack))
        out["allreduce_us_assumed_8_ranks"] = round(ar8, 2)
        out["slack_us_before_the_collective_shows"] = round(slack, 2)
        out["predicted_ms_per_step_8_gpus"] = round(step8, 4)
        out["predicted_weak_scaling_efficiency_8_gpus"] = round(steady_ms / step8, 4)
    return out


def dp_point(model, trainer, batches, steps, asr, fence):
    """Data parallel runs, EVERY rank (collectives inside): (1) the step's gradient all-reduce alone — the flat bucket(s),
    50 back-to-back calls on the training stream between two HIP events — through the communicator the loop used AND
    through the other data plane (hand-written IPC all-reduce / RCCL via the C ABI), so that one line carries both;
    (2) the timed loop once more with the collective stubbed out (each rank applies its local gradients: the replicas
    diverge, nothing is measured after this) — the difference to the headline's ms_per_step is what the collective costs
    INSIDE the loop."""
    from slu_hip import dp
    b = trainer.bucket
    stream = getattr(trainer, "_train_stream", None) or getattr(trainer, "_full_stream", None) or torch.cuda.current_stream()
    for f in b.flats.values():
        f.zero_()                                    # repeated in-place sums of real gradients would overflow
    used = getattr(b.comm, "kind", "torch.distributed (%s)" % torch.distributed.get_backend())
    out = {"data_plane": used, "allreduce_us_alone": _time_collective(b.allreduce_flats, stream),
           "allreduce_bytes": b.nbytes(), "collectives_per_step": 1 if b.comm is not None else len(b.flats)}
    alone = {used: out["allreduce_us_alone"]}
    dev = next(model.parameters()).device
    rank, world = dp.world()
    others = [("ipc", dp.IpcComm)] + ([] if dp._shared_device() else [("rccl", dp.DirectComm)])
    for kind, ctor in others:
        if kind == used:
            continue
        try:                                          # construction is collective: every rank walks this list in order
            other = ctor(rank, world, dev)
            alone[kind] = _time_collective(lambda: other.allreduce_flats(b.flats), stream)
            if kind == "ipc" and other.status() != 0:
                alone[kind] = "timed out"
            other.close()
        except Exception as e:                        # noqa: BLE001 - a side measurement never takes the headline down
            alone[kind] = "unavailable: %s" % str(e)[:120]
    out["allreduce_us_alone_by_data_plane"] = alone
    b.stub = True
    trainer._step_graphs.clear()
    try:
        run_steps(model, trainer, batches, steps, asr)           # re-captures the step without its collective
        fence()
        t0 = time.perf_counter()
        run_steps(model, trainer, batches, steps, asr)
        fence()
        dt = time.perf_counter() - t0
    finally:
        b.stub = False
        trainer._step_graphs.clear()
    tmax = torch.tensor([dt], dtype=torch.float64)
    if torch.distributed.get_backend() == "nccl":
        tmax = tmax.to(dev)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    out["ms_per_step_without_collective"] = round(1e3 * tmax.item() / steps, 4)
    return out


def rccl_info(world, trainer):
    """How the ranks talk: control plane (torch.distributed backend) and data plane (the gradient collective)."""
    if not torch.distributed.is_initialized():
        return None
    import torch.distributed as dist
    from slu_hip import lib
    backend = dist.get_backend()
    comm = trainer.bucket.comm if trainer.bucket is not None else None
    kind = getattr(comm, "kind", None)
    info = {"ranks": world,
            "control_plane": "torch.distributed backend %s%s" % (backend, " (= RCCL on ROCm)" if backend == "nccl" else ""),
            "data_plane": {"ipc": "slu_comm_allreduce_ipc: hand-written two-shot all-reduce over peer-mapped windows (xGMI)",
                           "rccl": "RCCL through the C ABI (slu_comm_allreduce_group) on the training stream",
                           None: "torch.distributed's collective on the flat buckets (%s)" % backend}[kind],
            "backend": backend, "algo": os.environ.get("NCCL_ALGO", "default"), "proto": os.environ.get("NCCL_PROTO", "default")}
    if getattr(comm, "selftest_us", None) is not None:
        info["ipc_selftest_us_per_call"] = round(comm.selftest_us, 2)
    if getattr(comm, "race_us", None) is not None:
        info["start_up_race_us"] = comm.race_us       # both planes timed at start-up (1.21 MB): the faster one carries the gradients
    try:
        v = lib.load().slu_comm_version()
        info["version"] = "RCCL %d.%d.%d" % (v // 10000, v // 100 % 100, v % 100) if v else "no RCCL mapped"
    except Exception as e:
        info["version"] = "unknown (%s)" % (e,)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="no_unfreezing", choices=sorted(CFG))
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: BASELINE configs[4] arithmetic (SLU_DTYPE=bf16) - reported as a separate line, never the headline")
    ap.add_argument("--hidden", type=int, default=0,
                    help="GRU hidden size of every layer (default: the cfg's 128); any other value is a labelled "
                         "synthetic variant (SURVEY 8.0-A: H = 512 extra point), never the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the extra large-batch point (B=2048/GPU, forward-dominant kernels throughput-bound)")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the per-kernel roofline measurement")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 PMC passes of the dominant kernel")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the extra measurements attached to the default line (exact fp32, host inputs, other workloads)")
    ap.add_argument("--sub", action="store_true", help=argparse.SUPPRESS)       # a side run started by another bench.py
    ap.add_argument("--share-gpu", action="store_true",
                    help="with --gpus N > visible GPUs: run the N ranks on the visible GPUs over gloo (functional check)")
    args = ap.parse_args()

    if args.dtype == "bf16":
        os.environ["SLU_DTYPE"] = "bf16"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
        return

    claim_stdout()
    from slu_hip import dp, lib
    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    lib.require_gfx950()
    samples = int(args.seconds * FS)
    asr = args.workload == "asr_pretrain"
    dev = torch.device("cuda", local)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.sub:
        args.no_side_runs = True
    parity = None
    if rank == 0:
        note("parity on the golden batch")
        parity = parity_check(dev)
    note("setup")
    # >= one look-ahead super-batch of DISTINCT synthetic batches (32 x 64 x 3 s = 393 MB resident): no super-batch holds a
    # batch twice
    n_distinct = int(os.environ.get("SLU_BENCH_DISTINCT", "32" if samples * args.batch <= 64 * 48000 else "8"))
    config, model, trainer, train_ds, work = setup(args.workload, rank, args.batch, samples, n_distinct, args.hidden)
    batches = [tuple(t.to(dev) for t in b) for b in train_ds.loader]       # inputs resident in HBM
    model.train()

    # One-off initialisation, not part of the W warm-up steps the contract asks for (which follow): every
    # super-batch shape a K-step and a W-step run produce (incl. the tail K mod width) is captured as a
    # hipGraph on its second appearance in its look-ahead slot, the optimisation step after three eager steps.
    note("initialisation")
    import training as _training
    depth = trainer.lookahead_depth(True, asr)[0]
    width = _training._lookahead_width(depth, args.batch) if depth else 1
    for _ in range(3):
        run_steps(model, trainer, batches, args.steps, asr)
        if args.warmup:
            run_steps(model, trainer, batches, args.warmup, asr)
    note("warmup")
    if args.warmup:
        run_steps(model, trainer, batches, args.warmup, asr)
    fence()
    note("timed region")
    ev0, ev_first = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    sums = run_steps(model, trainer, batches, args.steps, asr, first_done=ev_first)
    fence()
    elapsed = time.perf_counter() - t0
    fill_ms = ev0.elapsed_time(ev_first)
    note("timed region done: %.3f s" % elapsed)
    tmax = torch.tensor([elapsed], dtype=torch.float64)          # control plane: a host tensor over gloo
    if world > 1:
        if torch.distributed.get_backend() == "nccl":
            tmax = tmax.to(dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = tmax.item()
    loss_mean = (sums[0] / (args.steps * args.batch)).item()
    graphs = trainer.graph_stats()
    dp_costs = None
    if trainer.data_parallel:
        note("data parallel: the collective alone, the loop without it")
        try:
            dp_costs = dp_point(model, trainer, batches, args.steps, asr, fence)
        except Exception as e:                                   # a side measurement never takes the headline down
            dp_costs = {"error": str(e)[:200]}

    comm_info = rccl_info(world, trainer) if trainer.data_parallel else None
    graphs_dp = trainer.graph_stats() if trainer.data_parallel else {}
    if world > 1:
        trainer.close()                      # collective (the IPC windows are unmapped behind a barrier): every rank, now

    # steady state of the same loop (long run), reported beside `value` when K is short: the first
    # super-batch of a run has to be computed before its first step can start (pipeline fill)
    steady = None
    if args.steps < 256 and world == 1 and os.environ.get("SLU_BENCH_NO_STEADY", "0") != "1":      # (the env: timeline traces of the K-step region)
        n_long = 512
        run_steps(model, trainer, batches, n_long, asr)
        run_steps(model, trainer, batches, n_long, asr)
        fence()
        t1 = time.perf_counter()
        run_steps(model, trainer, batches, n_long, asr)
        fence()
        dt = time.perf_counter() - t1
        steady = {"steps": n_long, "ms_per_step": round(1e3 * dt / n_long, 4),
                  "utterances_per_s": round(args.batch * n_long / dt, 2)}

    payload = trainer.bucket.nbytes() if trainer.bucket is not None else 0
    if rank == 0:
        out = {
            "metric": "utterances/sec (train step, 3 s @16 kHz, B=64)",
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype_label(), "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[args.workload] + (
                           " -- SYNTHETIC VARIANT: GRU hidden size %d in every layer (reference cfgs use 128)" % args.hidden
                           if args.hidden else ""),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "samples_per_utterance": samples, "parallelism": "dp%d" % world,
                       "allreduce_bytes_per_step": payload, "mean_loss": round(loss_mean, 5),
                       "encoder_lookahead_batches": width if depth else 0},
            "pipeline_fill_ms": round(fill_ms, 3),
            "graphs_captured": graphs,
            "parity": parity,
        }
        if steady:
            out["steady_state"] = steady
            # (also inside `config`, which every consumer of the line keeps)
            out["config"]["steady_state_utterances_per_s"] = steady["utterances_per_s"]
            out["config"]["steady_state_ms_per_step"] = steady["ms_per_step"]
            out["config"]["steady_state_steps"] = steady["steps"]
        if trainer.data_parallel:
            out["rccl"] = comm_info
            if dp_costs:
                # per step: where the all-reduce sits, what it costs alone and what the loop costs without it
                out["rccl"].update(dp_costs)
            out["rccl"]["collective"] = graphs.get("collective")
            if "SLU_LOCAL_DEVICE" in os.environ:
                out["config"]["parallelism"] += " (ranks share the visible GPU(s): functional check only)"
        if world == 1 and not args.no_kernel_table:
            note("kernel table")
            table = kernel_table(model, trainer, args.batch, samples, width, asr)
            top = table[0]
            is_mfma = top["bound"] == "mfma"
            alg_per_launch = round(sum(s["algorithmic_MB"] for s in top["shapes"]) * 1e6 / len(top["shapes"]))
            # all bytes the frozen-prefix kernels move per super-batch against SURVEY 8(d)'s minimum (0.914 MB / utterance)
            prefix_bytes = sum(sh["algorithmic_MB"] * 1e6 for k in table for sh in k["shapes"] if "look-ahead" in sh["shape"])
            prefix_ms = sum(sh["us"] * 1e-3 for k in table for sh in k["shapes"] if "look-ahead" in sh["shape"])
            prefix_gflop = sum(sh["gflop"] for k in table for sh in k["shapes"] if "look-ahead" in sh["shape"])
            n_utt = args.batch * width
            out["roofline"] = {
                # the dominant kernel of the timed steps (largest share of a step's GPU time), measured by THIS process:
                # ALGORITHMIC work (SURVEY 8(d) flops / bytes of its launch shapes) / the average launch duration between two
                # HIP events on the CU-masked stream it runs on
                "bound": top["bound"],
                "achieved": top["algorithmic_tflops"] if is_mfma else top["hbm_tbs"] * 1e3,
                "peak": top["mfma_peak"] if is_mfma else PEAK_HBM_TBS * 1e3,
                "unit": "TFLOP/s" if is_mfma else "GB/s",
                "frac": top["frac"],
                "frac_hbm": top["hbm_frac"], "frac_mfma_algorithmic": top["mfma_frac"],
                "mfma_issued_frac": top["mfma_issued_frac"],           # utilisation of the unit by the 16-bit products issued
                "frac_of_fp32_mfma_peak": top["frac_of_fp32_mfma_peak"],
                "traffic": None, "traffic_over_algorithmic": None,      # filled below from live rocprofv3 PMC passes
                "traffic_source": PMC_SOURCE, "in_loop_source": INLOOP_SOURCE,
                "algorithmic_bytes_per_launch": alg_per_launch,
                "algorithmic_gflop_per_launch": round(top["algorithmic_gflop"] / top["launches_per_cycle"], 3),
                "kernel": top["kernel"], "launches": top["launches_per_cycle"], "avg_launch_ms": round(top["avg_us"] / 1e3, 5),
                "prefix_traffic_over_8d": round(prefix_bytes / (0.914e6 * n_utt), 2) if width > 1 else None,
                "prefix_bytes_per_super_batch": round(prefix_bytes) if width > 1 else None,
                "prefix_ms_per_super_batch_isolated": round(prefix_ms, 3) if width > 1 else None,
                # the frozen stages as a whole: their algorithmic (fp32-equivalent) flops per super-batch / the sum of their
                # isolated kernel times, against the whole chip's fp32 MFMA peak (they run on the look-ahead partition only)
                "prefix_fp32_equiv_tflops_isolated": round(prefix_gflop / prefix_ms, 1) if width > 1 and prefix_ms else None,
                "prefix_frac_of_fp32_mfma_peak": round(prefix_gflop / prefix_ms / PEAK_FP32_MFMA_TFLOPS, 3) if width > 1 and prefix_ms else None,
                "kernels": table[:6],
                "note": "per kernel: sums over its distinct launch shapes in one look-ahead cycle (%d steps) of the ALGORITHMIC fp32 "
                        "flops and HBM bytes / sum of the average durations; `frac` = max(algorithmic flops / peak of the MFMA unit "
                        "the kernel issues to, algorithmic bytes / HBM peak); mfma_issued_frac counts the 16-bit products actually "
                        "issued (6 per fp32 product on bf16x3, 3 on f16x2).  Each shape is launched 20x back to back (one hipGraph) on "
                        "the CU-masked stream it runs on during the timed steps (frozen stages: %d sequences on CUs [%d,%d); trainable "
                        "stages: %d sequences on CUs [0,%d)) between two HIP events on that stream.  Peaks are whole-chip: fp32 MFMA "
                        "157.3, dense bf16 / fp16 MFMA 2500 TFLOP/s, HBM 8 TB/s."
                        % (width, args.batch * width, _cu_split(), _n_cus(), args.batch, _cu_split())}
            r_ = out["roofline"]
            # SURVEY 8(d)'s MINIMUM bytes of the dominant kernel's share (layer input + pooled output only: the fp32 gx it
            # reads is avoidable in principle): 2 x 4 x 256 x (T + T/2) bytes per sequence and layer
            if top["kernel"].startswith("gru_bf") and width > 1:
                min_bytes = sum(4.0 * 256 * (T_ + -(-T_ // 2)) for T_ in (300, 150, 75, 38)) * n_utt
                r_["frac_8d"] = round(min_bytes / (top["avg_us"] * 1e-6 * top["launches_per_cycle"]) / (PEAK_HBM_TBS * 1e12), 4)
                if not args.no_pmc:
                    note("PMC passes of the dominant kernel (rocprofv3)")
                    pmc = pmc_dominant_kernel(n_utt, models_nsplit())
                    r_["pmc"] = pmc
                    if "error" not in pmc:
                        r_["traffic"] = pmc["hbm_bytes_per_launch"]
                        r_["traffic_over_algorithmic"] = round(pmc["hbm_bytes_per_launch"] / max(1, alg_per_launch), 3)
                        # the larger MEASURED limiter names the bound: the kernel's own dependent issue chain when the SIMDs
                        # spend most of a step on it while neither roofline is near
                        r_["issue_frac"] = pmc["issue_frac"]
                        if pmc["issue_frac"] > max(r_["frac_hbm"], r_["frac_mfma_algorithmic"], r_["mfma_issued_frac"]):
                            r_["bound_measured"] = "issue"
                        else:
                            r_["bound_measured"] = r_["bound"]
        # The side measurements run on rank 0 at N = 1 only: under data parallelism a Trainer built by
        # one rank alone would issue gradient all-reduces the other ranks never join.
        if world == 1 and args.workload == "no_unfreezing" and depth and not args.sub and not args.no_kernel_table:
            note("scaling model (suffix / prefix alone, one-rank collective)")
            try:
                ss = steady["ms_per_step"] if steady else 1e3 * elapsed / args.steps
                out["scaling_model"] = scaling_model(model, trainer, width, ss, batches)
                out["config"]["predicted_weak_scaling_efficiency_8_gpus"] = out["scaling_model"].get("predicted_weak_scaling_efficiency_8_gpus")
            except Exception as e:                           # noqa: BLE001 - a side measurement never takes the headline down
                out["scaling_model"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_large_batch and args.workload == "no_unfreezing" and not args.hidden:
            note("large-batch point")
            out["large_batch_point"] = large_batch_point(rank, samples)
        default_line = (world == 1 and args.workload == "no_unfreezing" and not args.hidden and args.dtype == "f32"
                        and not args.no_side_runs)
        if default_line:
            note("host-input point")
            out["host_inputs"] = host_inputs_point(model, trainer, batches, max(args.steps, 256), asr)
            note("feature parity of a super-batch, per arithmetic")
            try:
                out["parity"]["features_vs_oracle"] = feature_parity(model, config, args.batch, samples)
            except Exception as e:                           # a side measurement never takes the headline down
                out["parity"]["features_vs_oracle"] = {"error": str(e)[:200]}
            note("front end forward (configs[1])")
            try:
                out.setdefault("other_workloads", {})["frontend_fwd"] = frontend_fwd_point(model, config, args.batch, samples)
            except Exception as e:
                out.setdefault("other_workloads", {})["frontend_fwd"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline and not asr:
            note("cpu baseline")
            out["cpu_baseline"] = (cpu_baseline_seq2seq(config, args.batch, samples) if args.workload == "seq2seq"
                                   else cpu_baseline(config, args.batch, samples))
        if default_line:
            # fresh processes (their own graphs / environment); this process' model is released first
            del model, trainer, batches
            torch.cuda.empty_cache()
            note("side runs")
            common = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch), "--seconds", str(args.seconds)]
            # the other arithmetics of the frozen stages, each at the driver's K and in the 512-step steady state (`value` /
            # `steady_state` of the sub-line): exact fp32 MFMA; f16x2 under its range guard (22-bit operands: NARROWER than
            # the reference's fp32, opt-in, never the headline)
            out["exact_fp32"] = side_run(common, {"SLU_FROZEN_MATH": "fp32"})
            note("side run: frozen stages on guarded f16x2")
            out["frozen_f16x2"] = side_run(common, {"SLU_FROZEN_MATH": "auto"})
            short = ["--steps", "40", "--warmup", "10", "--batch", str(args.batch), "--seconds", str(args.seconds)]
            out.setdefault("other_workloads", {}).update({w: side_run(short + ["--workload", w]) for w in ("unfreeze_all", "asr_pretrain", "seq2seq")})
        emit_json(out)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
