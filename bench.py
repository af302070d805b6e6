#!/usr/bin/env python3
"""Benchmark of the hot path: SLU training steps (utterances/s) on synthetic 3 s @16 kHz waveforms,
B = 64 per GPU — BASELINE.json `metric`, workload = configs[3] (no_unfreezing: frozen pre-trained
SincNet/conv/biGRU encoder in train mode + intent GRU trained; forward, loss, backward, gradient
all-reduce, Adam) — through models.Model / training.Trainer on the HIP kernels.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload no_unfreezing|unfreeze_all]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  `value` = utterances all ranks processed / max-over-ranks wall time of the
K timed steps (inputs resident in HBM, barrier + synchronize on both sides).  `roofline` is for the
dominant kernel (the persistent GRU recurrence, fp32 MFMA bound): algorithmic flops of its launches
in the timed region / their HIP-event durations.  `cpu_baseline` times the CPU oracle (torch-CPU
restatement of the reference path) on a bounded sample of the same workload on this host's cores.
"""
import argparse
import json
import os
import shutil
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


def setup(workload, rank, batch, samples, n_batches):
    """read_config on the package's synthetic cfg, synthetic pre-training checkpoint, Model, Trainer."""
    import data
    import models
    import training

    work = tempfile.mkdtemp(prefix="slu_bench_")
    os.makedirs(os.path.join(work, "experiments"))
    name = "no_unfreezing_synthetic.cfg" if workload == "no_unfreezing" else "unfreeze_all_layers_synthetic.cfg"
    shutil.copy(os.path.join(PKG, "experiments", name), os.path.join(work, "experiments", name))
    cwd = os.getcwd()
    os.chdir(work)
    try:
        config = data.read_config(os.path.join("experiments", name))
        config.folder = os.path.join(work, config.folder)
        config.slu_path = "synthetic:%dx%dx%d" % (n_batches, batch, samples)
        config.seed = 1234 + rank                      # per-rank synthetic data
        train_ds, _, _ = data.get_SLU_datasets(config)
        torch.manual_seed(4321)                        # synthetic "pre-trained" encoder (no .pth published)
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


def run_steps(model, trainer, batches, n):
    """n optimisation steps through Trainer's own step loop (forward, loss, backward, gradient
    all-reduce, Adam; with the frozen-encoder look-ahead pipeline when it applies)."""
    dev = next(model.parameters()).device
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    loader = [batches[i % len(batches)] for i in range(n)]
    for vals, _ in trainer._iterate(loader, True, False):
        if torch.is_tensor(vals):
            sums.add_(vals)
        else:
            sums += torch.stack([v.detach().double() for v in vals])
    return sums


def cpu_baseline(config, batch, samples, budget_s=20.0):
    """The CPU oracle (torch-CPU restatement of the reference path, ATen GRU like the reference) on
    the same workload: train-mode forward + backward of the trainable part + Adam, all host cores."""
    from oracle import slu_oracle as O
    # torch's default intra-op pool (what the reference would run with on this host), capped: with
    # one thread per SMT sibling (256 on the MI355X host) ATen's small per-step GRU matmuls thrash.
    cores = max(1, min(torch.get_num_threads(), 64))
    torch.set_num_threads(cores)
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
    step(False)                                                # warm-up
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (n < 3 and first < budget_s / 4) or (time.perf_counter() - t0 + first < budget_s * 0.6 and n < 50):
        step(False)
        n += 1
    dedup = n * batch / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    step(True)                                                 # one step with the 80 in-loop convolutions
    faithful = batch / (time.perf_counter() - t0)
    return {"value": round(dedup, 2), "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": "%d train steps of B=%d x %d s on %d torch-CPU threads (oracle, ATen GRU, one conv per "
                      "Sinc forward); reference-faithful variant with the 80 redundant in-loop convolutions "
                      "(models.py:98-108): %.2f utterances/s" % (n, batch, samples // FS, cores, faithful)}


def large_batch_point(rank, samples, batch=2048, steps=3):
    """The same train step at B = 2048 per GPU (128 sequence tiles x 2 directions = 256 recurrence
    workgroups, one per CU): where the recurrence stops being bound by the 8-workgroup latency chain.
    Reported next to the headline number, never as `value`."""
    from slu_hip import ops
    os.environ["SLU_LOOKAHEAD"] = "0"          # one big batch per step: nothing to look ahead to
    config, model, trainer, train_ds, work = setup("no_unfreezing", rank, batch, samples, 1)
    dev = next(model.parameters()).device
    batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]
    model.train()
    run_steps(model, trainer, batches, 1)
    torch.cuda.synchronize()
    ops.profile_start()
    t0 = time.perf_counter()
    run_steps(model, trainer, batches, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ops.profile_stop()
    kflops = sum(p[3] for p in prof)
    kms = sum(p[1].elapsed_time(p[2]) for p in prof)
    ach = kflops / (kms * 1e-3) / 1e12
    shutil.rmtree(work, ignore_errors=True)
    del model, trainer, batches
    torch.cuda.empty_cache()
    os.environ.pop("SLU_LOOKAHEAD", None)
    return {"batch_per_gpu": batch, "utterances_per_s": round(batch * steps / dt, 1),
            "ms_per_step": round(1e3 * dt / steps, 3),
            "gru_seq_fwd_kernel_tflops": round(ach, 2), "gru_seq_fwd_kernel_frac_of_fp32_mfma_peak":
            round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}


def note(msg):
    if os.environ.get("SLU_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="no_unfreezing", choices=["no_unfreezing", "unfreeze_all"])
    ap.add_argument("--batch", type=int, default=BATCH, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-large-batch", action="store_true",
                    help="skip the extra large-batch point (B=2048/GPU, forward-dominant kernels throughput-bound)")
    args = ap.parse_args()

    from slu_hip import dp, lib, ops
    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local)
    lib.require_gfx950()
    samples = int(args.seconds * FS)
    note("setup")
    config, model, trainer, train_ds, work = setup(args.workload, rank, args.batch, samples, 4)
    dev = torch.device("cuda", local)
    batches = [(x.to(dev), y.to(dev)) for x, y in train_ds.loader]       # inputs resident in HBM
    model.train()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # one-off initialisation (the look-ahead slots capture a super-batch shape as a hipGraph on its second
    # appearance, the step graph after three eager steps: a few dozen steps per shape); not part of the W
    # warm-up steps the contract asks for, which follow
    note("initialisation")
    import training as _training
    width = _training._lookahead_width(trainer.lookahead_depth(True, False)[0] or 1, args.batch)
    run_steps(model, trainer, batches, max(0, max(48, 8 * width) - args.warmup))
    note("warmup")
    run_steps(model, trainer, batches, args.warmup)
    fence()
    note("timed region")
    t0 = time.perf_counter()
    sums = run_steps(model, trainer, batches, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.3f s" % elapsed)
    # Dominant-kernel timing: the same steps once more with the kernels launched eagerly (hipGraph
    # replays cannot carry per-kernel events) and a HIP event pair around every recurrence launch,
    # recorded on the stream the kernel is launched on.
    os.environ["SLU_GRAPHS"] = "0"
    ops.profile_start()
    run_steps(model, trainer, batches, min(args.steps, 32))
    fence()
    prof = ops.profile_stop()
    os.environ.pop("SLU_GRAPHS", None)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    elapsed = tmax.item()
    loss_mean = (sums[0] / args.steps).item()

    # dominant kernel: the persistent GRU recurrence (5 launches per step)
    kflops = sum(p[3] for p in prof)
    kms = sum(p[1].elapsed_time(p[2]) for p in prof)
    achieved = kflops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    payload = trainer.bucket.nbytes() if trainer.bucket is not None else 0

    if rank == 0:
        look = trainer.lookahead_depth(True, False)[0]
        if look:
            import training
            look = training._lookahead_width(look, args.batch)
        out = {
            "metric": "utterances/sec (train step, 3 s @16 kHz, B=64)",
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": "utterances/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("experiments/no_unfreezing.cfg SLU train step (frozen SincNet/conv/biGRU "
                                    "encoder in train mode + intent biGRU trained): fwd + 3-slot CE + bwd + "
                                    "grad all-reduce + Adam" if args.workload == "no_unfreezing" else
                                    "SLU train step with every encoder layer unfrozen (unfreeze_all_layers "
                                    "end state): fwd + CE + full bwd + grad all-reduce + Adam"),
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "samples_per_utterance": samples, "parallelism": "dp%d" % world,
                       "allreduce_bytes_per_step": payload, "mean_loss": round(loss_mean, 5),
                       "encoder_lookahead_batches": look},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 5), "traffic": None,
                         "kernel": "gru_seq_fwd4_kernel<128> / gru_seq_fwd_kernel<128>",
                         "launches": len(prof), "avg_launch_ms": round(kms / max(len(prof), 1), 4),
                         "note": "fp32 MFMA flops of h(BxH)*W_hh^T(Hx3H) per recurrence step, both directions, "
                                 "summed over the launches of an eager re-run of the timed steps / their HIP-event "
                                 "durations; a launch covers %d sequences (look-ahead super-batch of the frozen "
                                 "layers) or %d (trainable intent layer): %d or %d 4-sequence workgroups "
                                 "(v_mfma_f32_4x4x1) on 256 CUs" %
                                 (args.batch * max(1, look), args.batch,
                                  2 * -(-args.batch * max(1, look) // 4), 2 * -(-args.batch // 4))},
        }
        # The two side measurements run on rank 0 at N = 1 only: under data parallelism a Trainer built by
        # one rank alone would issue gradient all-reduces the other ranks never join.
        if world == 1 and not args.no_large_batch and args.workload == "no_unfreezing":
            note("large-batch point")
            out["large_batch_point"] = large_batch_point(rank, samples)
        note("cpu baseline")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(config, args.batch, samples)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
