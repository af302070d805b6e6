#!/usr/bin/env python3
"""Where does a small-batch recurrence step spend its time?  Runs the 4-sequence forward kernel of the PROBE build
(tools/build_probe.sh; SLU_HIP_LIB) with parts switched off (SLU_GRU_DBG bit mask: 1 no gx prefetch loads, 2 no output
stores, 4 no reserve stores, 8 gates without transcendentals, 16 no MFMAs, 32 no LDS exchange and no barrier, 64 no LDS
exchange but the barrier) and prints microseconds per step."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "SLU_HIP_LIB" not in os.environ:
    lib = os.path.join(ROOT, "end-to-end-slu_amd", "lib_alt", "libslu_hip_probe.so")
    for mask in [int(m) for m in os.environ.get('PROBE_MASKS', '0,7,16,31,39,63,71,95').split(',')]:
        env = dict(os.environ, SLU_HIP_LIB=lib, SLU_GRU_DBG=str(mask))
        out = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, capture_output=True, text=True)
        print("dbg=%2d  %s" % (mask, out.stdout.strip().replace("\n", " | ")), (out.stderr[-300:] if out.returncode else ""))
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "end-to-end-slu_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from slu_hip import ops  # noqa: E402
from bench_kernels import timeit  # noqa: E402

H, D = 128, 2
if os.environ.get("PROBE_KERNEL") == "bf":
    # the split-precision recurrence of the frozen layers at the super-batch size (bits: 1 no gx loads, 2 no output
    # stores, 8 gates without transcendentals, 16 no MFMAs, 32 one rounding instead of the 3-way split, 64 no LDS stores)
    wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
    bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
    for B, T in ((1280, 150), (64, 150)):
        gx = torch.randn(T, B, 2 * 3 * H, device="cuda")
        f, _ = timeit(lambda: ops.gru_seq_fwd_bf16(gx, wf, wr, bf, br, T, B, H, D, 3), reps=3)
        print("bf16x3 B=%d T=%d: %.3f us/step" % (B, T, f / T))
    sys.exit(0)
wf, wr = torch.randn(3 * H, H, device="cuda") * 0.08, torch.randn(3 * H, H, device="cuda") * 0.08
bf, br = torch.randn(3 * H, device="cuda"), torch.randn(3 * H, device="cuda")
for B, T in ((64, 300), (32, 1000)):
    gx = torch.randn(T, B, 2 * 3 * H, device="cuda")
    f, _ = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, False), reps=3)
    fr, _ = timeit(lambda: ops.gru_seq_fwd(gx, wf, wr, bf, br, T, B, H, D, True), reps=3)
    print("B=%d T=%d: fwd %.3f us/step, fwd+reserve %.3f us/step" % (B, T, f / T, fr / T))
